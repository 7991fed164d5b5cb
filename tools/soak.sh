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
# round 6: a frame size that is not a multiple of 4 on the sending side, and the receivers (band pipeline with its second thread, the transcoder's worker rotation)
python - <<'PY'
import numpy as np
from ultragrid_amd import synth
np.concatenate([synth.s2_video("UYVY", 1366, 766, salt=i) for i in range(4)]).tofile("/tmp/wxga.raw")
PY
oracle/_ref/ug_harness dxt:DXT5 UYVY 1366 766 /tmp/wxga.raw /tmp/o.bin 1 host 4 ${REPEAT:-400} 2>&1 | grep -E "THROUGHPUT|fail|error" | sed "s/^/== dxt:DXT5 1366x766 /" | cut -c1-160
oracle/_ref/ug_harness dxt:DXT5 UYVY 3840 2160 /tmp/4k.raw /tmp/4k.dxt5 1 host 1 1 > /dev/null
oracle/_ref/ug_harness jpeg:q=75:restart=4 UYVY 3840 2160 /tmp/4k.raw /tmp/4k.jpg 1 host 1 1 > /dev/null
for job in "DXT5 UYVY /tmp/4k.dxt5 7680 -" "DXT5 RGBA /tmp/4k.dxt5 15424 mi355x-bands=4" "JPEG UYVY /tmp/4k.jpg 7680 -" "JPEG DXT5 /tmp/4k.jpg 3840 mi355x-device=0:0"; do
  set -- $job
  UG_PARAM=$([ "$5" = - ] && echo "" || echo $5) UG_DEC_REPEAT=${DEC_REPEAT:-4000} oracle/_ref/ug_dec_harness $1 $2 3840 2160 $3 /tmp/o.raw $4 > /tmp/d.log 2>&1 &
  pid=$!; peak=0
  while kill -0 $pid 2>/dev/null; do r=$(awk '/VmRSS/{print $2}' /proc/$pid/status 2>/dev/null); [ -n "$r" ] && [ "$r" -gt "$peak" ] && peak=$r; sleep 0.2; done
  wait $pid; echo "== decompress $1 -> $2 pitch=$4 ${5} rc=$? peak_rss_kb=$peak"; grep -E "THROUGHPUT|DELAY|differs|fail|error" /tmp/d.log | cut -c1-160
done
