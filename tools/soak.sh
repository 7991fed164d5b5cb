#!/bin/bash
# Soak: thousands of 4K frames through the modules (reference framework harness), throughput + peak RSS.  GPU box.
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import numpy as np
from ultragrid_amd import synth
fr = [synth.s2_video("UYVY", 3840, 2160, salt=i) for i in range(2)]
np.concatenate([fr[i % 2] for i in range(8)]).tofile("/tmp/4k.raw")
PY
for cfg in ${CFGS:-"dxt:DXT5:workers=1" "dxt:DXT5" "jpeg:q=75:restart=4:workers=1" "jpeg:q=75:restart=4"}; do
  oracle/_ref/ug_harness $cfg UYVY 3840 2160 /tmp/4k.raw /tmp/o.bin 1 host 8 ${REPEAT:-400} > /tmp/h.log 2>&1 &
  pid=$!
  peak=0
  while kill -0 $pid 2>/dev/null; do
    r=$(awk '/VmRSS/{print $2}' /proc/$pid/status 2>/dev/null); [ -n "$r" ] && [ "$r" -gt "$peak" ] && peak=$r
    sleep 0.2
  done
  wait $pid; echo "== $cfg rc=$? peak_rss_kb=$peak"; grep -E "THROUGHPUT|OK|fail|error" /tmp/h.log | cut -c1-160
done
