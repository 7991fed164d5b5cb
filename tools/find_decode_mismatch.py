#!/usr/bin/env python3
"""Search for a damaged stream on which the GPU decoder and the oracle disagree; dump it.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from oracle import pyoracle as po
from ultragrid_amd import codec as hip, lib as L
from test_oracle_jpeg_decode import picture
from test_gpu_jpeg_decode import _damage, _own_stream

w, h = 208, 88
rgb = picture(w, h, seed=11, noise=3.0)
data = _own_stream(hip, po.convert_frame("RGB", "UYVY", rgb, w, h), L.PF_UYVY, w, h, 85, 3, 422)
# round 6: every third draw damages a stream WITHOUT restart intervals (one segment: the self-synchronising parallel decode, or the sequential walk where the data ends early)
w2, h2 = 400, 240
data_nori = _own_stream(hip, po.convert_frame("RGB", "UYVY", picture(w2, h2, seed=12, noise=5.0), w2, h2), L.PF_UYVY, w2, h2, 90, 0, 420)
data_few = _own_stream(hip, po.convert_frame("RGB", "UYVY", picture(w2, h2, seed=13, noise=5.0), w2, h2), L.PF_UYVY, w2, h2, 90, 100, 420)   # ... and one of four long segments (FFmpeg's slices)
dec = hip.JpegDecoder()
found = 0
KINDS = ["bytes", "cut", "cut_raw", "drop_rst", "extra_rst", "marker"]
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12000):
    rng = np.random.default_rng(seed)
    bad = _damage((data_nori if (seed // 6) % 3 == 2 else (data_few if (seed // 6) % 3 == 1 else data)), rng, KINDS[seed % len(KINDS)])
    if seed % 7 == 0:                      # and two kinds of damage at once
        bad = _damage(bad, rng, "bytes")
    try:
        info, crop, _ = po.jpeg_decode_planes(bad)
    except Exception:
        continue
    got = dec.planes(bad)
    for c in range(3):
        g = got[c].cpu().numpy()
        if not np.array_equal(g, crop[c]):
            diff = np.argwhere(g != crop[c])
            print("seed", seed, KINDS[seed % len(KINDS)], "comp", c, "ndiff", len(diff), "first", diff[0], "last", diff[-1], flush=True)
            os.makedirs("gpurun_out/mismatch", exist_ok=True)
            open(f"gpurun_out/mismatch/bad_{seed}.jpg", "wb").write(bad)
            open("gpurun_out/mismatch/good.jpg", "wb").write(data)
            found += 1
            break
    if found >= 3:
        break
print("found", found)
