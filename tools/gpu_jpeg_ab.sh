#!/bin/bash
# JPEG entropy coder A/B on the GPU box: correctness first (the JPEG GPU tests), then rocprofv3 kernel traces of the whole encoder
# with the block-parallel coder (default) and the wave-per-segment coder (UG_JPEG_WAVE_KERNEL=1), restart intervals 2 / 4 / 8.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/jpeg_ab
python -m pytest tests/test_gpu_jpeg.py -q -x 2>&1 | tail -8
python -m pytest tests/test_module_harness.py -q -k "jpeg" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for ri in ${RIS:-4}; do
for mode in 0 1; do
  UG_JPEG_WAVE_KERNEL=$mode rocprofv3 --kernel-trace --stats -d /tmp/pj_$mode -o t -- python $GRAFT_REPO_ROOT/tools/jpeg_profile.py 3840 2160 $ri > /tmp/pj_$mode.log 2>&1
  echo "== restart $ri wave_kernel=$mode  $(tail -1 /tmp/pj_$mode.log | grep -E '^[0-9]+$')"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pj_$mode -name "*.db" | head -1) 2>&1 | grep -v "^==" | paste - - | sed 's/(anonymous namespace):://; s/unsigned //g' | awk '{printf "%-44.44s %s %s %s\n", $2, $(NF-3), $(NF-2), $(NF-1)}'
  rm -rf /tmp/pj_$mode
done
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/jpeg_ab/ab.txt
