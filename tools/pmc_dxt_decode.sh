#!/bin/bash
# rocprofv3 counter passes over the DXT decoders (4K, 8 frames per launch): bash tools/pmc_dxt_decode.sh <tag> [configs...].  GPU box.
# Counters in their own runs (--kernel-trace + --pmc only).  Output: gpurun_out/pmc_dxtdec_<tag>/summary.txt
set -u
TAG=${1:-r03}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_dxtdec_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "DXT5 RGBA 8" "DXT5 UYVY 8" "DXT1 RGBA 8" "DXT1 UYVY 8"; fi
for cfg in "$@"; do
    n=$(echo $cfg | tr ' ' '_')
    CMD="python $ROOT/tools/one_decode.py $cfg 20"
    rocprofv3 --kernel-trace --stats -d $OUT -o tr_$n -- $CMD > $OUT/tr_$n.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT -o sq1_$n -- $CMD > $OUT/sq1_$n.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE -d $OUT -o sq2_$n -- $CMD > $OUT/sq2_$n.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVES_EQ_64 SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_FLAT -d $OUT -o sq3_$n -- $CMD > $OUT/sq3_$n.log 2>&1
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT -o tcc_$n -- $CMD > $OUT/tcc_$n.log 2>&1
    grep -h "us/frame" $OUT/tr_$n.log | tail -1
done
python $ROOT/tools/pmc_summary.py $OUT/*.db > $OUT/summary.txt 2>&1
grep -iE "error|invalid|not found|fail" $OUT/*.log | head -5
grep -E "decode_kernel|calls=" $OUT/summary.txt | cut -c1-220 | sed 's/.*HIP_vect//' | head -120
rm -f $OUT/*.db
