#!/usr/bin/env python3
"""CPU: random search for a valid JPEG stream on which the decode oracle (oracle/jpeg_decode_oracle.c) and libjpeg-turbo (Pillow) disagree,
wherever libjpeg hands out untouched samples: every plane of 4:4:4 streams, the luma plane of 4:2:2 / 4:2:0 streams, greyscale.  Qualities
1..100, optimised tables, restart intervals, sizes from 1x1, smooth to noise content.  usage: python tools/find_oracle_vs_libjpeg.py [n]"""
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
bad = 0
for seed in range(n):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 260)), int(rng.integers(1, 160))
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / (3 + 40 * rng.random())) * np.cos(yy / (3 + 30 * rng.random())), 128 + 90 * np.cos(xx / 33.0 + yy / (5 + 20 * rng.random())),
                     128 + 80 * np.sin(yy / (2 + 9 * rng.random()))], -1)
    img = (base + rng.normal(0, [0.0, 2.0, 10.0, 60.0, 200.0][int(rng.integers(5))], base.shape)).clip(0, 255).astype(np.uint8)
    grey = rng.random() < 0.1
    kw = dict(quality=int(rng.integers(1, 101)), optimize=bool(rng.integers(2)))
    if not grey:
        kw["subsampling"] = int(rng.integers(3))
    r = int(rng.integers(4))
    if r == 1:
        kw["restart_marker_blocks"] = int(rng.integers(1, 9))
    elif r == 2:
        kw["restart_marker_rows"] = int(rng.integers(1, 3))
    b = io.BytesIO()
    try:
        Image.fromarray(img[..., 1] if grey else img).save(b, "JPEG", **kw)
    except OSError:
        continue        # libjpeg refuses the combination (e.g. a restart interval the picture has no room for)
    data = b.getvalue()
    _, crop, _ = po.jpeg_decode_planes(data)
    ref = Image.open(io.BytesIO(data))
    if grey:
        ok = np.array_equal(crop[0], np.asarray(ref))
    else:
        ref.draft("YCbCr", None)
        ref = np.asarray(ref)
        ok = np.array_equal(crop[0], ref[..., 0])
        if kw["subsampling"] == 0:
            ok = ok and np.array_equal(crop[1], ref[..., 1]) and np.array_equal(crop[2], ref[..., 2])
    if not ok:
        print("MISMATCH seed", seed, w, h, kw, "grey" if grey else "", flush=True)
        bad += 1
        if bad >= 5:
            break
print("streams", seed + 1, "mismatches", bad)
