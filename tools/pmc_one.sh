#!/bin/bash
# rocprofv3 counter passes for one DXT configuration: bash tools/pmc_one.sh <tag> IN OUT W H FRAMES
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/one_kernel.py $* 20"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT -o sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $OUT -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT -o tcc -- $CMD > $OUT/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d $OUT -o tcp -- $CMD > $OUT/tcp.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/*.db > $OUT/summary.txt 2>&1
grep -h "Mpx/s" $OUT/sq1.log | tail -1
grep -E "dxt_encode|calls=" $OUT/summary.txt | grep -v "^==" | awk '{print $(NF-2), $(NF-1), $NF}' | sort -u | head -60
