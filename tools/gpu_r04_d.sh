#!/bin/bash
# Round 4, session D: phase clock of the fused JPEG kernel; the headline kernel's LDS conversion tables A/B.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04d; mkdir -p $OUT
UG_JPEG_PROF=1 timeout 120 python tools/bench_jpeg_batch.py --only batch --calls 40 > $OUT/jpeg_prof_fused.txt 2>&1; grep "UG_JPEG_PROF\|frames per call" $OUT/jpeg_prof_fused.txt
UG_JPEG_PROF=1 UG_JPEG_FUSED=0 timeout 120 python tools/bench_jpeg_batch.py --only batch --calls 40 > $OUT/jpeg_prof_unfused.txt 2>&1; grep "UG_JPEG_PROF\|frames per call" $OUT/jpeg_prof_unfused.txt
UG_MI355X_LIB=$ROOT/ultragrid_amd/libug_mi355x_ldsconv.so timeout 600 python -m pytest tests/test_gpu_dxt.py -q -x -k "uyvy or UYVY or full or golden or glsl" 2>&1 | tail -3 > $OUT/pytest_ldsconv.log; tail -3 $OUT/pytest_ldsconv.log
ROUNDS=3 STEPS=100 timeout 900 bash tools/ab_bench.sh ultragrid_amd/libug_mi355x.so ultragrid_amd/libug_mi355x_ldsconv.so > $OUT/ldsconv_ab.txt 2>&1; cat $OUT/ldsconv_ab.txt
( cd /tmp && export TMPDIR=/tmp
  for v in "" _ldsconv; do
    CMD="python $ROOT/bench.py --steps 2 --warmup 1 --launches-per-step 16 --no-cpu-baseline --no-e2e"
    UG_MI355X_LIB=$ROOT/ultragrid_amd/libug_mi355x$v.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $OUT/pm$v -o p -- $CMD > $OUT/pm$v.log 2>&1
  done )
for v in "" _ldsconv; do echo "== build libug_mi355x$v.so"; python tools/pmc_summary.py $(find $OUT/pm$v -name "*.db") 2>&1 | grep "dxt_encode_kernel" | grep -v "^kernel" | cut -c1-160; done > $OUT/ldsconv_pmc.txt; cat $OUT/ldsconv_pmc.txt
rm -rf $OUT/pm $OUT/pm_ldsconv
