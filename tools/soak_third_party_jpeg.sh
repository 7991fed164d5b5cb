#!/bin/bash
# A receiver decoding another sender's JPEG for a while: 1080p 4:2:2 libjpeg streams without restart intervals and with eight per frame (FFmpeg's slices), 4000 frames each
# through the reference's decompress framework (oracle/_ref/ug_dec_harness); fps, peak resident memory.   tools/soak_third_party_jpeg.sh   (GPU box)
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import io, numpy as np
from PIL import Image
yy, xx = np.mgrid[0:1080, 0:1920]
rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
rgb = (rgb + np.random.default_rng(1).normal(0, 6, rgb.shape)).clip(0, 255).astype(np.uint8)
for name, kw in (("nr", {}), ("s8", {"restart_marker_rows": 9})):
    b = io.BytesIO(); Image.fromarray(rgb).save(b, "JPEG", quality=80, subsampling=1, **kw); open(f"/tmp/{name}.jpg", "wb").write(b.getvalue())
PY
for name in nr s8; do
  for out in UYVY RGBA; do
    UG_DEC_REPEAT=${DEC_REPEAT:-4000} oracle/_ref/ug_dec_harness JPEG $out 1920 1080 /tmp/$name.jpg /tmp/o.raw > /tmp/d.log 2>&1 &
    pid=$!; peak=0
    while kill -0 $pid 2>/dev/null; do r=$(awk '/VmRSS/{print $2}' /proc/$pid/status 2>/dev/null); [ -n "$r" ] && [ "$r" -gt "$peak" ] && peak=$r; sleep 0.2; done
    wait $pid; echo "== decompress third-party 1080p JPEG ($name) -> $out rc=$? peak_rss_kb=$peak"; grep -E "THROUGHPUT|differs|fail|error" /tmp/d.log | cut -c1-160
  done
done
