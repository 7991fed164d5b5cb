#!/bin/bash
# Receiver-side throughput through the reference's own video_decompress.c (host frame in, host frame out, pageable buffers as a display
# hands them over), 4K and 8K, display pitch = line size and = line size + 64, beside a copy-only twin of the same byte mix where the
# harness has one (COPYONLY lines).  VERDICT r5 "What's weak" #3.
#   tools/receiver_fps.sh > gpurun_out/<name>.txt     (GPU box; needs oracle/_ref/ug_harness and ug_dec_harness)
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import numpy as np
from ultragrid_amd import synth
for name, w, h in (("4k", 3840, 2160), ("8k", 7680, 4320)):
    one = synth.s2_video("UYVY", w, 240, salt=1)
    np.tile(one.reshape(240, -1), (h // 240, 1)).tofile(f"/tmp/{name}_uyvy.raw")
PY
H=oracle/_ref/ug_harness
D=oracle/_ref/ug_dec_harness
for sz in "4k 3840 2160 150" "8k 7680 4320 50"; do
  set -- $sz; n=$1; w=$2; h=$3; rep=$4
  $H "jpeg:q=75:restart=4" UYVY $w $h /tmp/${n}_uyvy.raw /tmp/$n.jpg 1 host 1 1 > /dev/null
  $H "dxt:DXT5" UYVY $w $h /tmp/${n}_uyvy.raw /tmp/$n.dxt5 1 host 1 1 > /dev/null
  $H "dxt:DXT1" UYVY $w $h /tmp/${n}_uyvy.raw /tmp/$n.dxt1 1 host 1 1 > /dev/null
  for in in DXT5 DXT1 JPEG; do
    case $in in DXT5) f=/tmp/$n.dxt5;; DXT1) f=/tmp/$n.dxt1;; JPEG) f=/tmp/$n.jpg;; esac
    for out in RGBA UYVY; do
      [ $out = RGBA ] && ls=$(( w * 4 )) || ls=$(( w * 2 ))
      for pitch in $ls $(( ls + 64 )); do
        echo "== $in -> $out ${w}x${h} pitch=$pitch (line $ls) $UG_RECV_NOTE"
        UG_DEC_REPEAT=$rep UG_DEC_ROUNDS=3 UG_DEC_COPY_TWIN=1 $D $in $out $w $h $f /tmp/o.raw $pitch 2>&1 | grep -E "COPYONLY|status [^2]|failed"
      done
    done
  done
  echo "== JPEG -> DXT5 ${w}x${h}"; UG_DEC_REPEAT=$rep UG_DEC_ROUNDS=3 UG_DEC_COPY_TWIN=1 $D JPEG DXT5 $w $h /tmp/$n.jpg /tmp/o.raw $w 2>&1 | grep -E "COPYONLY|DELAY|status [^2]|failed"
done
