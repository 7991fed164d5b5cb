cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04p; mkdir -p $OUT
FILTER="copyBuffer\|roll_cuda\|elementwise\|fillBuffer\|CatArray\|at::native"
( cd /tmp && export TMPDIR=/tmp
  for mode in default two; do
    if [ $mode = two ]; then export UG_JPEG_LOOKBACK=0; fi
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/js_$mode -o t -- python $ROOT/tools/bench_jpeg_batch.py --only single --calls 20 > $OUT/js_$mode.log 2>&1
    python $ROOT/tools/pmc_summary.py $(find $OUT/js_$mode -name "*.db") 2>&1 | grep -v "$FILTER" | grep -A1 "^kernel" | head -8 | cut -c1-150
    grep "per call" $OUT/js_$mode.log | tail -1
    rm -rf $OUT/js_$mode
  done )
UG_JPEG_PROF=1 timeout 120 python tools/bench_jpeg_batch.py --sub 444 --only batch --calls 40 2>&1 | grep "UG_JPEG_PROF"
UG_JPEG_PROF=1 timeout 120 python tools/bench_jpeg_batch.py --only single --calls 40 2>&1 | grep "UG_JPEG_PROF"
