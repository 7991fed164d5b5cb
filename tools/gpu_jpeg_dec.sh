#!/bin/bash
# JPEG decoder: parity tests, timing, kernel trace.  usage (via gpurun): bash tools/gpu_jpeg_dec.sh <tag>
tag=${1:-dec}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_jpeg_decode.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -3 $out/pytest.log
timeout 300 python tools/bench_jpeg_decode.py --json $out/jpeg_decode.json 2>&1 | tee $out/bench.log
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace -o t -- python $R/tools/bench_jpeg_decode.py --concurrent 1 > $R/$out/trace.log 2>&1)
python tools/pmc_summary.py $out/trace/t_results.db > $out/kernel_trace.txt 2>&1; head -14 $out/kernel_trace.txt | cut -c1-170
rm -rf $out/trace
timeout 900 python -m pytest tests/test_module_harness.py -m gpu -x -q -k "jpeg" > $out/pytest_module.log 2>&1; tail -3 $out/pytest_module.log
