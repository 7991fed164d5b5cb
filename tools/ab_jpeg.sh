#!/bin/bash
# A/B of the JPEG encoder kernels under rocprofv3 --kernel-trace: tools/ab_jpeg.sh libA.so libB.so ...  (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
for r in 1 2; do
for lib in "$@"; do
  n=$(basename $lib .so)
  UG_MI355X_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --stats -d /tmp/pj_$n -o t -- python $GRAFT_REPO_ROOT/tools/jpeg_profile.py 3840 2160 4 > /tmp/pj_$n.log 2>&1
  echo "== $n"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pj_$n -name "*.db" | head -1) 2>&1 | grep -v "^==" | paste - - | sed 's/(anonymous namespace):://; s/unsigned //g' | awk '{printf "%-40.40s %s %s %s\n", $2, $(NF-3), $(NF-2), $(NF-1)}'
  rm -rf /tmp/pj_$n
done
done
