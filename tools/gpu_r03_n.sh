#!/bin/bash
# r03n: after ug::WaveWords -- full GPU suite, the kernel tables again, random pixfmt search, bench line
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log
python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; grep -c . $OUT/kernels_table.txt
python tools/bench_pixfmt_all.py --json $OUT/pixfmt_all_8k.json > $OUT/pixfmt_all_8k.txt 2>&1; tail -1 $OUT/pixfmt_all_8k.txt | cut -c1-300
timeout 600 python tools/find_pixfmt_mismatch.py 2>&1 | tail -2 > $OUT/find_pixfmt.txt; cat $OUT/find_pixfmt.txt
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench_line.json
