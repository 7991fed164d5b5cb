#!/bin/bash
# r03o: end-of-round refresh after the whole-line store work -- full GPU suite, tables, bytes written per row, bench line + kernel trace
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; tail -2 $OUT/pytest.log
python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; grep -c . $OUT/kernels_table.txt
python tools/bench_pixfmt_all.py --json $OUT/pixfmt_all_8k.json > $OUT/pixfmt_all_8k.txt 2>&1; tail -1 $OUT/pixfmt_all_8k.txt | cut -c1-200
bash tools/pmc_write_by_row.sh tools/ab/libZ_final.so > /dev/null 2>&1
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-120 $OUT/bench_line.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-e2e > $OUT/trace.log 2>&1 )
python tools/pmc_summary.py $OUT/trace/bench_results.db > $OUT/kernel_trace.txt 2>&1; head -3 $OUT/kernel_trace.txt | cut -c1-120; rm -rf $OUT/trace
timeout 200 python tools/find_pixfmt_mismatch.py 2>&1 | tail -1
