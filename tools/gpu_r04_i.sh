#!/bin/bash
# Round 4, session G: JPEG coder after the cheaper append + the look-back that waits only for what it needs.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04i; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_jpeg_rtp_compat.py -q -x 2>&1 | grep -v lavc_vid_conv | tail -4 > $OUT/pytest_jpeg.log; tail -4 $OUT/pytest_jpeg.log
timeout 120 python tools/bench_jpeg_batch.py --only batch > $OUT/jpeg_batch.txt 2>&1; grep "frames per call" $OUT/jpeg_batch.txt
UG_JPEG_FUSED=0 timeout 120 python tools/bench_jpeg_batch.py --only batch > $OUT/jpeg_batch_unfused.txt 2>&1; grep "frames per call" $OUT/jpeg_batch_unfused.txt
timeout 120 python tools/bench_jpeg_batch.py --sub 422 --only batch > $OUT/jpeg_batch_422.txt 2>&1; grep "frames per call" $OUT/jpeg_batch_422.txt
timeout 120 python tools/bench_jpeg_batch.py --n 16 --only batch > $OUT/jpeg_batch_n16.txt 2>&1; grep "frames per call" $OUT/jpeg_batch_n16.txt
UG_JPEG_PROF=1 timeout 120 python tools/bench_jpeg_batch.py --only batch --calls 40 2>&1 | grep "UG_JPEG_PROF" > $OUT/jpeg_prof_fused.txt; cat $OUT/jpeg_prof_fused.txt
( cd /tmp && export TMPDIR=/tmp
  CMD="python $ROOT/tools/bench_jpeg_batch.py --only batch --calls 40"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/jp4 -o p -- $CMD > $OUT/jp4.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/jp4 -name "*.db") 2>&1 | grep -v "copyBuffer\|roll_cuda\|elementwise\|fillBuffer\|CatArray\|at::native" > $OUT/jpeg_batch_pmc.txt
grep -A1 "^kernel" $OUT/jpeg_batch_pmc.txt | head -4 | cut -c1-150; grep "^pmc" $OUT/jpeg_batch_pmc.txt | cut -c60-150
rm -rf $OUT/jp4
