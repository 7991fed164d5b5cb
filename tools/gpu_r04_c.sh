#!/bin/bash
# Round 4, session C..: quick loop on the JPEG coder -- its tests, its rate, its kernel times.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_jpeg_rtp_compat.py -q -x 2>&1 | grep -v lavc_vid_conv | tail -6 > $OUT/pytest_jpeg.log; tail -6 $OUT/pytest_jpeg.log
timeout 120 python tools/bench_jpeg_batch.py > $OUT/jpeg_batch.txt 2>&1; grep "frames per call" $OUT/jpeg_batch.txt
UG_JPEG_FUSED=0 timeout 120 python tools/bench_jpeg_batch.py --only batch > $OUT/jpeg_batch_unfused.txt 2>&1; grep "frames per call" $OUT/jpeg_batch_unfused.txt
UG_JPEG_TICKET=1 timeout 120 python tools/bench_jpeg_batch.py --only batch > $OUT/jpeg_batch_ticket.txt 2>&1; grep "frames per call" $OUT/jpeg_batch_ticket.txt
timeout 120 python tools/bench_jpeg_batch.py --sub 422 --only batch > $OUT/jpeg_batch_422.txt 2>&1; grep "frames per call" $OUT/jpeg_batch_422.txt
timeout 120 python tools/bench_jpeg_batch.py --n 16 --only batch > $OUT/jpeg_batch_n16.txt 2>&1; grep "frames per call" $OUT/jpeg_batch_n16.txt
( cd /tmp && export TMPDIR=/tmp
  CMD="python $ROOT/tools/bench_jpeg_batch.py --only batch --calls 40"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt -o t -- $CMD > $OUT/jt.log 2>&1
  UG_JPEG_FUSED=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt2 -o t -- $CMD > $OUT/jt2.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/jp4 -o p -- $CMD > $OUT/jp4.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_ANY -d $OUT/jp5 -o p -- $CMD > $OUT/jp5.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/jt $OUT/jt2 $OUT/jp4 $OUT/jp5 -name "*.db") 2>&1 | grep -v "copyBuffer\|roll_cuda\|elementwise\|fillBuffer\|CatArray\|at::native" > $OUT/jpeg_batch_pmc.txt
grep -A1 "^kernel" $OUT/jpeg_batch_pmc.txt | head -12 | cut -c1-150; grep "^pmc" $OUT/jpeg_batch_pmc.txt | cut -c60-150
rm -rf $OUT/jt $OUT/jt2 $OUT/jp4 $OUT/jp5
