#!/bin/bash
# One frame in flight through the reference's compress framework + the product's DXT module: compress_end - compress_start per frame (host frame in, host
# frame out, PCIe included), bands=1 against bands=<k>.  GPU box; needs oracle/_ref/ug_harness.   tools/module_latency.sh [pace_us]
cd ${GRAFT_REPO_ROOT:-.}
PACE=${1:-4000}
python - <<'PY'
import numpy as np
from ultragrid_amd import synth
for name, fmt, w, h in (("4k_uyvy", "UYVY", 3840, 2160), ("8k_v210", "v210", 7680, 4320), ("8k_uyvy", "UYVY", 7680, 4320)):
    fr = [synth.s2_video(fmt, w, h, salt=i) for i in range(2)]
    np.concatenate(fr).tofile(f"/tmp/lat_{name}.raw")
PY
H=oracle/_ref/ug_harness
for pinned in 1 0; do
for spec in "8k_v210 v210 7680 4320" "8k_uyvy UYVY 7680 4320" "4k_uyvy UYVY 3840 2160"; do
  set -- $spec
  for k in 1 2 4 8 16; do
    line=$( ( [ $pinned = 1 ] && export UG_HARNESS_PINNED=1; UG_HARNESS_PACE_US=$PACE $H dxt:DXT5:workers=1:bands=$k $2 $3 $4 /tmp/lat_$1.raw /tmp/o.bin 1 host 2 30 ) | grep LATENCY)
    echo "$1 $( [ $pinned = 1 ] && echo pinned || echo pageable ) source, bands=$k: $line"
  done
done
done
