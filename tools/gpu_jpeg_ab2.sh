#!/bin/bash
# A/B of library variants on the whole JPEG encoder: tools/gpu_jpeg_ab2.sh lib1.so lib2.so ...   (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp
for r in 1 2; do
for lib in "$@"; do
  n=$(basename $lib .so)
  UG_MI355X_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --stats -d /tmp/pj_$n -o t -- python $GRAFT_REPO_ROOT/tools/jpeg_profile.py 3840 2160 ${RI:-4} > /tmp/pj_$n.log 2>&1
  echo "== $n $(tail -1 /tmp/pj_$n.log)"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pj_$n -name "*.db" | head -1) 2>&1 | grep -v "^==" | paste - - | sed 's/(anonymous namespace):://; s/unsigned //g' | awk '{printf "%-44.44s %s %s %s\n", $2, $(NF-3), $(NF-2), $(NF-1)}' | head -2
  rm -rf /tmp/pj_$n
done
done
