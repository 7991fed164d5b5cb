cd ${GRAFT_REPO_ROOT:-.}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02e; mkdir -p $OUT
python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; tail -2 $OUT/kernels_table.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace_kernels -o k -- python $ROOT/tools/bench_kernels.py > $OUT/trace_kernels.log 2>&1)
python tools/pmc_summary.py $OUT/trace_kernels/k_results.db > $OUT/all_kernels_trace.txt 2>&1
python tools/bench_pixfmt_all.py --json $OUT/pixfmt_all_8k.json > $OUT/pixfmt_all_8k.txt 2>&1; tail -1 $OUT/pixfmt_all_8k.txt | cut -c1-400
python tools/bench_decode.py --json $OUT/decode.json > $OUT/decode.txt 2>&1; cat $OUT/decode.txt | grep -- "->"
rm -rf $OUT/trace_kernels
