mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -q --maxfail=15 2>&1 | grep -v "lavc_vid_conv" | tail -25 > gpurun_out/r02b/pytest.log
tail -5 gpurun_out/r02b/pytest.log
python bench.py > gpurun_out/r02b/bench_line.json 2> gpurun_out/r02b/bench.err; tail -c 3000 gpurun_out/r02b/bench_line.json; tail -5 gpurun_out/r02b/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02b/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/r02b/trace.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/r02b/trace | head; 
bash tools/pmc_collect.sh r02b > gpurun_out/r02b/pmc.log 2>&1; tail -30 gpurun_out/r02b/pmc.log
