#!/bin/bash
# rocprofv3 kernel trace + counter passes over the JPEG decoder (4K 4:2:2 restart 4 only); GPU box.  Output: gpurun_out/pmc_jpeg_dec/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_jpeg_dec
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_jpeg_decode.py --configs 1 --concurrent 1 --seconds 0.05"
rocprofv3 --kernel-trace --stats -d $OUT -o tr -- python $ROOT/tools/bench_jpeg_decode.py --configs 1 --concurrent 1 > $OUT/tr.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT -o sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SENDMSG -d $OUT -o sq3 -- $CMD > $OUT/sq3.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/tr_results.db $OUT/sq*.db 2>&1 > $OUT/summary.txt
grep -iE "error|invalid|not found|fail" $OUT/*.log | head -5
grep -A1 "^kernel" $OUT/summary.txt | grep -v "^--" | head -24 | cut -c1-100
grep "huff_decode" $OUT/summary.txt | grep pmc | cut -c1-20,75-150
rm -f $OUT/*.db
