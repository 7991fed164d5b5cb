#!/bin/bash
# Meaning of SQ_ACTIVE_INST_VALU2: counters on the occupancy microbenchmark's pure-fast / pure-slow / mixed kernels.  GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_mb; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -d $O -o mb -- $R/tools/occupancy_microbench > $O/mb.log 2>&1
python - <<PY
import sqlite3
c = sqlite3.connect("$O/mb_results.db")
v = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')") if 'counters_collection' in r[0]][0]
cols = [r[1] for r in c.execute(f"pragma table_info({v})")]
kn = 'kernel_name' if 'kernel_name' in cols else [x for x in cols if 'kernel' in x and 'name' in x][0]
rows = list(c.execute(f"select {kn}, dispatch_id, counter_name, value, grid_size from {v} order by dispatch_id"))
import collections
d = collections.OrderedDict()
for k, disp, name, val, g in rows:
    d.setdefault((disp, k[:60], g), {})[name] = val
for (disp, k, g), vals in d.items():
    iv = vals.get('SQ_INSTS_VALU', 0)
    if iv < 1e6: continue
    print(disp, k[-22:], "grid", g, "INSTS %.3e ACTIVE %.3e VALU2 %.3e  VALU2/INSTS %.3f ACTIVE/INSTS %.3f" % (iv, vals.get('SQ_ACTIVE_INST_VALU', 0), vals.get('SQ_ACTIVE_INST_VALU2', 0), vals.get('SQ_ACTIVE_INST_VALU2', 0) / iv, vals.get('SQ_ACTIVE_INST_VALU', 0) / iv))
PY
