#!/bin/bash
# r03i: fast index stages of the DXT encoders -- parity first, then interleaved A/B against the -DUG_DXT_NO_FAST_INDEX build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03i
timeout 900 python -m pytest tests/test_gpu_dxt.py -x -q 2>&1 | tail -5 | tee gpurun_out/r03i/tests.txt
timeout 900 python tools/find_dxt_mismatch.py 3000 2>&1 | tail -12 | tee gpurun_out/r03i/find_dxt.txt
for wl in "" 1080p-rgb-dxt1 8k-v210; do
  echo "== workload ${wl:-4k-uyvy-dxt5 (default)}" | tee -a gpurun_out/r03i/ab.txt
  WORKLOAD=$wl ROUNDS=3 STEPS=100 bash tools/ab_bench.sh tools/ab/libA_nofast.so tools/ab/libB_fastindex.so 2>&1 | tee -a gpurun_out/r03i/ab.txt
done
