#!/bin/bash
# Bytes written to HBM (WRITE_SIZE) per dispatch of every kernel tools/bench_kernels.py launches, grouped by kernel AND grid size (= per table row),
# for several builds of the library: tools/pmc_write_by_row.sh libA.so libB.so ...  -> gpurun_out/write_by_row/<lib>.txt + the tool's own timing table.  GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/write_by_row; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so); rm -rf /tmp/wr_$n; mkdir -p /tmp/wr_$n
  UG_MI355X_LIB=$ROOT/$lib rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/wr_$n -o w -- python $ROOT/tools/bench_kernels.py > $OUT/${n}_table.txt 2>&1
  python - /tmp/wr_$n/w_results.db > $OUT/$n.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
g = [x for x in ("grid_size", "grid_size_x", "grid_x") if x in cols]
gy = [x for x in ("grid_size_y", "grid_y") if x in cols]
gz = [x for x in ("grid_size_z", "grid_z") if x in cols]
key = ", ".join(g[:1] + gy[:1] + gz[:1]) or "0"
print("# columns:", cols)
for row in c.execute(f"select kernel_name, {key}, count(*), avg(value) from counters_collection where counter_name = 'WRITE_SIZE' group by kernel_name, {key} order by kernel_name, {key}"):
    k = row[0]; grid = row[1:-2]; n, avg = row[-2], row[-1]
    print(f"{avg * 1024 / 1e6:10.2f} MB written  n={n:5d}  grid={grid}  {k[:140]}")
PY
done
