#!/bin/bash
# What a receiver's frame is made of: rocprofv3 kernel + memory-copy trace of the reference's decompress framework around our modules (oracle/_ref/ug_dec_harness),
# 4K, 200 frames per row.   tools/receiver_trace.sh > gpurun_out/<name>.txt   (GPU box)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
python - <<'PY'
import numpy as np
from ultragrid_amd import synth
np.concatenate([synth.s2_video("UYVY", 3840, 2160, salt=i) for i in range(2)]).tofile("/tmp/4k_uyvy.raw")
PY
H=oracle/_ref/ug_harness; D=$(pwd)/oracle/_ref/ug_dec_harness
$H "jpeg:q=75:restart=4" UYVY 3840 2160 /tmp/4k_uyvy.raw /tmp/4k.jpg 1 host 1 1 > /dev/null
$H "dxt:DXT5" UYVY 3840 2160 /tmp/4k_uyvy.raw /tmp/4k.dxt5 1 host 1 1 > /dev/null
for row in "JPEG UYVY /tmp/4k.jpg" "JPEG RGBA /tmp/4k.jpg" "DXT5 UYVY /tmp/4k.dxt5"; do
  set -- $row
  out=/tmp/rt_$1_$2; rm -rf $out
  echo "== decompress $1 -> $2 3840x2160, 200 frames"
  (cd /tmp && UG_DEC_REPEAT=200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out -- $D $1 $2 3840 2160 $3 /tmp/o.raw 2>&1 | grep THROUGHPUT)
  python - $out <<'PY'
import glob, sqlite3, sys
for path in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(path)
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"  kernel {name[:96]:96s} calls={calls:5d} avg_us={avg / 1e3 if avg > 1e4 else avg:9.2f}")
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    for n in names:
        if "memory_cop" in n.lower() and "rocpd_" not in n.lower():
            cols = [r[1] for r in c.execute(f"pragma table_info('{n}')")]
            print("  [", n, cols, "]")
            try:
                if "duration" in cols:
                    key = "name" if "name" in cols else cols[0]
                    for row in c.execute(f"select {key}, count(*), avg(duration), sum(size) * 1.0 / count(*) from {n} group by {key}" if "size" in cols else f"select {key}, count(*), avg(duration), 0 from {n} group by {key}"):
                        print(f"  copy {str(row[0])[:60]:60s} calls={row[1]:5d} avg_us={row[2] / 1e3:9.2f} avg_bytes={row[3]:.0f}")
            except sqlite3.Error as e:
                print("   (", e, ")")
PY
done
