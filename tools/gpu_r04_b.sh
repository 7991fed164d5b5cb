#!/bin/bash
# Round 4, session B: the placing / fused JPEG coder -- tests first, then its rate and kernel trace.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_jpeg_rtp_compat.py tests/test_gpu_jpeg_decode.py -q -x 2>&1 | grep -v lavc_vid_conv | tail -25 > $OUT/pytest_jpeg.log; tail -25 $OUT/pytest_jpeg.log
timeout 600 python -m pytest tests/test_module_harness.py tests/test_reference_unit_tests.py -k "jpeg or gpujpeg" -q -x 2>&1 | grep -v lavc_vid_conv | tail -8 > $OUT/pytest_mod.log; tail -8 $OUT/pytest_mod.log
timeout 120 python tools/bench_jpeg_batch.py > $OUT/jpeg_batch.txt 2>&1; cat $OUT/jpeg_batch.txt
UG_JPEG_FUSED=0 timeout 120 python tools/bench_jpeg_batch.py > $OUT/jpeg_batch_unfused.txt 2>&1; cat $OUT/jpeg_batch_unfused.txt
timeout 120 python tools/bench_jpeg_batch.py --sub 422 > $OUT/jpeg_batch_422.txt 2>&1; cat $OUT/jpeg_batch_422.txt
timeout 120 python tools/bench_jpeg_batch.py --n 16 > $OUT/jpeg_batch_n16.txt 2>&1; cat $OUT/jpeg_batch_n16.txt
( cd /tmp && export TMPDIR=/tmp
  CMD="python $ROOT/tools/bench_jpeg_batch.py --only batch --calls 40"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt -o t -- $CMD > $OUT/jt.log 2>&1
  UG_JPEG_FUSED=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt2 -o t -- $CMD > $OUT/jt2.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/jp4 -o p -- $CMD > $OUT/jp4.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/jp5 -o p -- $CMD > $OUT/jp5.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/jp1 -o p -- $CMD > $OUT/jp1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/jp3 -o p -- $CMD > $OUT/jp3.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/jt $OUT/jt2 $OUT/jp4 $OUT/jp5 $OUT/jp1 $OUT/jp3 -name "*.db") 2>&1 | grep -v "copyBuffer\|roll_cuda\|elementwise\|fillBuffer\|CatArray\|at::native" > $OUT/jpeg_batch_pmc.txt
grep -A1 "^kernel" $OUT/jpeg_batch_pmc.txt | head -24 | cut -c1-150
rm -rf $OUT/jt $OUT/jt2 $OUT/jp1 $OUT/jp3 $OUT/jp4 $OUT/jp5
