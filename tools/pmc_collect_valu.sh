cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_v7; mkdir -p $O
CMD="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O -o a -- $CMD > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_FLOPS_FP32 SQ_THREAD_CYCLES_VALU -d $O -o b -- $CMD > $O/b.log 2>&1
python $R/tools/pmc_summary.py $O/*.db
tail -2 $O/a.log | grep -i -E "error|invalid" 
