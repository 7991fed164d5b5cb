#!/bin/bash
# Collect rocprofv3 counters for the headline bench in separate passes (no trace domains besides
# --kernel-trace; PMC slot limits: SQ 8 / TCC 4 / GRBM 2, MI355X_MICROARCH.md).  Run on the GPU box:
#   bash tools/pmc_collect.sh <tag>      -> gpurun_out/pmc_<tag>/*.db
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --launches-per-step 16 --no-cpu-baseline --no-e2e --no-configs --no-parity-check"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT -o sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT -o tcc -- $CMD > $OUT/tcc.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/*.db > $OUT/summary.txt 2>&1
tail -3 $OUT/*.log | grep -iE "error|fail|invalid" | head
cat $OUT/summary.txt
