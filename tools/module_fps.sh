#!/bin/bash
# Module-level throughput through the reference's compress framework (host frames in, host frames out, PCIe included):
#   tools/module_fps.sh   (GPU box; needs oracle/_ref/ug_harness)
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import numpy as np
from ultragrid_amd import synth
for name, fmt, w, h, n in (("4k_uyvy", "UYVY", 3840, 2160, 8), ("8k_v210", "v210", 7680, 4320, 4), ("1080_rgb", "RGB", 1920, 1080, 16)):
    fr = [synth.s2_video(fmt, w, h, salt=i) if fmt != "RGB" else synth.s1_random(fmt, w, h, salt=i) for i in range(min(n, 2))]
    np.concatenate([fr[i % len(fr)] for i in range(n)]).tofile(f"/tmp/{name}.raw")
PY
H=oracle/_ref/ug_harness
for cfg in "dxt:DXT5:workers=1" "dxt:DXT5" "dxt:DXT5:workers=4" "jpeg:q=75:restart=4:workers=1" "jpeg:q=75:restart=4" "jpeg:q=75:restart=4:workers=3"; do
  echo "== $cfg  4K UYVY"; $H $cfg UYVY 3840 2160 /tmp/4k_uyvy.raw /tmp/o.bin 1 host 8 40 | grep THROUGHPUT
done
for cfg in "dxt:DXT5:workers=1" "dxt:DXT5"; do
  echo "== $cfg  8K v210"; $H $cfg v210 7680 4320 /tmp/8k_v210.raw /tmp/o.bin 1 host 4 25 | grep THROUGHPUT
  echo "== $cfg  1080p RGB -> DXT5"; $H $cfg RGB 1920 1080 /tmp/1080_rgb.raw /tmp/o.bin 1 host 16 60 | grep THROUGHPUT
done
# receiver side: the reference's decompress framework around our modules, host frame in, host frame out
D=oracle/_ref/ug_dec_harness
$H "jpeg:q=75:restart=4" UYVY 3840 2160 /tmp/4k_uyvy.raw /tmp/4k.jpg 1 host 1 1 > /dev/null
$H "dxt:DXT5" UYVY 3840 2160 /tmp/4k_uyvy.raw /tmp/4k.dxt5 1 host 1 1 > /dev/null
for out in UYVY RGBA DXT1; do echo "== decompress JPEG -> $out 4K"; UG_DEC_REPEAT=300 $D JPEG $out 3840 2160 /tmp/4k.jpg /tmp/o.raw | grep THROUGHPUT; done
for out in UYVY RGBA; do echo "== decompress DXT5 -> $out 4K"; UG_DEC_REPEAT=300 $D DXT5 $out 3840 2160 /tmp/4k.dxt5 /tmp/o.raw | grep THROUGHPUT; done
