#!/bin/bash
# HBM read requests and written bytes per launch of EVERY kernel that tools/bench_kernels.py launches (two --pmc passes); GPU box.
#   bash tools/pmc_all_kernels.sh -> gpurun_out/pmc_all_kernels/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_all_kernels
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk; mkdir -p /tmp/pk
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d /tmp/pk -o rd -- python $ROOT/tools/bench_kernels.py > /tmp/pk/rd.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pk -o wr -- python $ROOT/tools/bench_kernels.py > /tmp/pk/wr.log 2>&1
python - <<'PY' > $OUT/summary.txt
import sqlite3, collections
rows = collections.defaultdict(dict)
for db in ("/tmp/pk/rd_results.db", "/tmp/pk/wr_results.db"):
    c = sqlite3.connect(db)
    for k, counter, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        rows[k][counter] = (n, avg)
for k, v in sorted(rows.items()):
    rd = v.get("TCC_EA0_RDREQ_sum", (0, 0))[1]; rd32 = v.get("TCC_EA0_RDREQ_32B_sum", (0, 0))[1]; wr = v.get("WRITE_SIZE", (0, 0))[1]
    # 32-byte requests are counted in RDREQ too: bytes = 128 * (RDREQ - RDREQ_32B) + 32 * RDREQ_32B  (requests are 32 B or 128 B on this path)
    rbytes = 128 * (rd - rd32) + 32 * rd32
    print(f"{rbytes / 1e6:10.2f} MB read {wr * 1024 / 1e6:10.2f} MB written  rd32={rd32:10.0f}  {k[:150]}")
PY
grep -iE "error|invalid" /tmp/pk/*.log | head -3
wc -l $OUT/summary.txt
