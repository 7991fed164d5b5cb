#!/bin/bash
# Round 4, end-of-round evidence in one gpurun call: the GPU suite, the bench line + the rocprofv3 kernel trace of the same command, the JPEG
# encoder's rates, the per-kernel table, random searches.  Lands in gpurun_out/r04n/; tools/copy_evidence_r04.sh gpurun_out/r04n copies what is kept.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04n; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_line.json
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-e2e > $OUT/trace.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/trace -name "*.db") > $OUT/kernel_trace.txt 2>&1; head -4 $OUT/kernel_trace.txt | cut -c1-160; tail -1 $OUT/trace.log | cut -c1-400 > $OUT/trace_bench_line.json
rm -rf $OUT/trace
python bench.py --workload 4k-uyvy-jpeg420 --no-e2e > $OUT/bench_4k-uyvy-jpeg420.json 2>> $OUT/bench.err
for wl in 8k-v210 1080p-rgb-dxt1; do python bench.py --workload $wl --no-e2e > $OUT/bench_$wl.json 2>> $OUT/bench.err; done
{ timeout 100 python tools/bench_jpeg_batch.py; timeout 100 python tools/bench_jpeg_batch.py --n 16 --only batch; timeout 100 python tools/bench_jpeg_batch.py --sub 422; timeout 100 python tools/bench_jpeg_batch.py --sub 444;
  timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 7680x4320 --n 4; timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 1920x1080 --n 16; } 2>&1 | grep "per call" > $OUT/jpeg_batch_all.txt; cat $OUT/jpeg_batch_all.txt
timeout 900 python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; grep -c . $OUT/kernels_table.txt
timeout 600 python tools/find_encode_mismatch.py 1500 2>&1 | tail -2 > $OUT/find_encode.txt; cat $OUT/find_encode.txt
timeout 600 python tools/find_dxt_mismatch.py 1500 2>&1 | tail -2 > $OUT/find_dxt.txt; cat $OUT/find_dxt.txt
timeout 300 python tools/find_module_mismatch.py 2>&1 | tail -2 > $OUT/find_module.txt; cat $OUT/find_module.txt
ls $OUT
