#!/bin/bash
# Round 4, session A: the new tests (self-launching bench, NUMA in the product, decoders vs oracle at scale, batch regressions),
# the JPEG batch path's evidence BEFORE the coder rework (kernel trace of the batch form alone + traffic counters per kernel), the rotation sweep.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04a; mkdir -p $OUT
python -m pytest tests/test_gpu_dxt_decode.py tests/test_gpu_jpeg.py tests/test_numa.py "tests/test_module_harness.py::test_workers_run_on_the_gpu_s_numa_node" \
   "tests/test_gpu_bench_contract.py::test_gpus_flag_without_a_launcher_starts_its_own_ranks" tests/test_module_harness.py -k "numa or batched or launcher or dxt_decode or jpeg" \
   -q -x 2>&1 | grep -v lavc_vid_conv | tail -15 > $OUT/pytest.log; tail -4 $OUT/pytest.log
python tools/bench_jpeg_batch.py > $OUT/jpeg_batch_before.txt 2>&1; cat $OUT/jpeg_batch_before.txt
( cd /tmp && export TMPDIR=/tmp
  CMD="python $ROOT/tools/bench_jpeg_batch.py --only batch --calls 40"
  rocprofv3 --kernel-trace --stats -d $OUT/jt -o t -- $CMD > $OUT/jt.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/jp1 -o p -- $CMD > $OUT/jp1.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/jp2 -o p -- $CMD > $OUT/jp2.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/jp3 -o p -- $CMD > $OUT/jp3.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/jp4 -o p -- $CMD > $OUT/jp4.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/jp5 -o p -- $CMD > $OUT/jp5.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/jt $OUT/jp1 $OUT/jp2 $OUT/jp3 $OUT/jp4 $OUT/jp5 -name "*.db") 2>&1 | grep -v "copyBuffer\|roll_cuda\|elementwise\|fillBuffer" > $OUT/jpeg_batch_before_pmc.txt
grep -c . $OUT/jpeg_batch_before_pmc.txt; head -12 $OUT/jpeg_batch_before_pmc.txt | cut -c1-150
rm -rf $OUT/jt $OUT/jp1 $OUT/jp2 $OUT/jp3 $OUT/jp4 $OUT/jp5
timeout 900 python tools/rotation_sweep.py > $OUT/rotation_sweep.txt 2> $OUT/rotation_sweep.err; cat $OUT/rotation_sweep.txt; tail -2 $OUT/rotation_sweep.err
