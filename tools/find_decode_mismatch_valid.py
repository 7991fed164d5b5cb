#!/usr/bin/env python3
"""Random search over VALID streams (libjpeg via Pillow: qualities 1..100, all three samplings, optimised tables, restart intervals in blocks
and rows, picture sizes from 1x1 up, smooth to pure-noise content) for one on which the GPU decoder and the oracle disagree.  GPU box.
usage: python tools/find_decode_mismatch_valid.py [n]"""
import io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 24     # (Pillow's encoder buffer: optimised tables on the larger pictures need more than the default)
from oracle import pyoracle as po
from ultragrid_amd import codec as hip, lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dec = hip.JpegDecoder()
found = refused = 0
for seed in range(n):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    if seed % 4 == 3:      # round 6: pictures large enough for the parallel decode of scans without restart intervals (from 4 KiB of scan data)
        w, h = int(rng.integers(100, 900)), int(rng.integers(80, 600))
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / (3 + 40 * rng.random())) * np.cos(yy / (3 + 30 * rng.random())), 128 + 90 * np.cos(xx / 33.0 + yy / (5 + 20 * rng.random())),
                     128 + 80 * np.sin(yy / (2 + 9 * rng.random()))], -1)
    img = (base + rng.normal(0, [0.0, 2.0, 10.0, 60.0, 200.0][int(rng.integers(5))], base.shape)).clip(0, 255).astype(np.uint8)
    kw = dict(quality=int(rng.integers(1, 101)), subsampling=int(rng.integers(3)), optimize=bool(rng.integers(2)))
    r = int(rng.integers(4))
    if r == 1:
        kw["restart_marker_blocks"] = int(rng.integers(1, 9))
    elif r == 2:
        kw["restart_marker_rows"] = int(rng.integers(1, 3))
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    data = b.getvalue()
    _, crop, _ = po.jpeg_decode_planes(data)
    try:
        got = dec.planes(data)
    except L.UgHipError as e:
        refused += 1
        print("refused", seed, w, h, kw, e, flush=True)
        continue
    for c in range(3):
        g = got[c].cpu().numpy()
        if g.shape != crop[c].shape or not np.array_equal(g, crop[c]):
            print("MISMATCH seed", seed, w, h, kw, "comp", c, flush=True)
            found += 1
            break
    if found >= 3:
        break
print("streams", seed + 1, "mismatches", found, "refused", refused)
