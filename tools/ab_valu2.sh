#!/bin/bash
# A/B with the co-issue counter: tools/ab_valu2.sh libA.so libB.so ...  -> ms per launch, VALU instr/wave, co-issued (VALU2) instr/wave
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  n=$(basename $lib .so); O=/tmp/v2_$n; rm -rf $O
  UG_MI355X_LIB=$R/$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVES -d $O -o a -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-e2e > /dev/null 2>&1
  python - <<PY
import sqlite3
c = sqlite3.connect("$O/a_results.db")
v = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')") if 'counters_collection' in r[0]][0]
d = dict((r[0], r[1]) for r in c.execute(f"select counter_name, avg(value) from {v} group by counter_name"))
t = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')") if 'top_kernels' in r[0]]
dur = list(c.execute(f"select * from {t[0]}"))[0] if t else None
w = d['SQ_WAVES']
print("$n", "instr/wave %.0f  co-issued/wave %.0f  slots/wave %.0f" % (d['SQ_INSTS_VALU'] / w, d['SQ_ACTIVE_INST_VALU2'] / w, (d['SQ_INSTS_VALU'] - d['SQ_ACTIVE_INST_VALU2']) / w), dur[:6] if dur else "")
PY
done
