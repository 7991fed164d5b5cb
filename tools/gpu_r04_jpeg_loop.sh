#!/bin/bash
# Round 4: the quick loop the JPEG encoder work ran on (sessions B..L): its tests, the rate of every placement / fusion variant, phase clock, SQ counters.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04jpeg; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_jpeg_rtp_compat.py -q -x 2>&1 | grep -v lavc_vid_conv | tail -4 > $OUT/pytest_jpeg.log; tail -4 $OUT/pytest_jpeg.log
for v in "" "UG_JPEG_LOOKBACK=1" "UG_JPEG_FUSED=0"; do echo "== $v"; env $v timeout 120 python tools/bench_jpeg_batch.py --only batch 2>&1 | grep "frames per call"; done > $OUT/jpeg_batch.txt; cat $OUT/jpeg_batch.txt
timeout 120 python tools/bench_jpeg_batch.py 2>&1 | grep "per call" > $OUT/jpeg_batch_both.txt; cat $OUT/jpeg_batch_both.txt
timeout 120 python tools/bench_jpeg_batch.py --sub 422 --only batch 2>&1 | grep "frames per call" > $OUT/jpeg_batch_422.txt; cat $OUT/jpeg_batch_422.txt
timeout 120 python tools/bench_jpeg_batch.py --n 16 --only batch 2>&1 | grep "frames per call" > $OUT/jpeg_batch_n16.txt; cat $OUT/jpeg_batch_n16.txt
UG_JPEG_PROF=1 timeout 120 python tools/bench_jpeg_batch.py --only batch --calls 40 2>&1 | grep "UG_JPEG_PROF" > $OUT/jpeg_prof_fused.txt; cat $OUT/jpeg_prof_fused.txt
( cd /tmp && export TMPDIR=/tmp
  CMD="python $ROOT/tools/bench_jpeg_batch.py --only batch --calls 40"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/jp4 -o p -- $CMD > $OUT/jp4.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/jp4 -name "*.db") 2>&1 | grep -v "copyBuffer\|roll_cuda\|elementwise\|fillBuffer\|CatArray\|at::native" > $OUT/jpeg_batch_pmc.txt
grep -A1 "^kernel" $OUT/jpeg_batch_pmc.txt | head -6 | cut -c1-150; grep "^pmc" $OUT/jpeg_batch_pmc.txt | cut -c40-150
rm -rf $OUT/jp4
