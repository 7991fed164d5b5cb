#!/usr/bin/env python3
"""Random search for a picture on which the product's JPEG stream and libjpeg-turbo's (float DCT, tests/libjpeg_float.py) differ in their
entropy-coded bytes: packed RGB -> R,G,B 4:4:4 at any size; UYVY -> 4:2:2 / 4:2:0 (libjpeg gets the planes of the reference's converters
through jpeg_write_raw_data) at sizes whose block grid fills whole MCUs (blocks wholly outside the picture are padding that the two fill
differently, tests/test_gpu_jpeg.py); quality 1 ... 100, restart intervals 1 ... 64 and none... every content kind.  GPU box.
usage: python tools/find_libjpeg_mismatch.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import libjpeg_float as ljf
from oracle import pyoracle as po
from ultragrid_amd import codec as hip, lib as L


def content(rng, n):
    kind = int(rng.integers(5))
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:
        return np.clip(128 + 30 * rng.standard_normal(n), 0, 255).astype(np.uint8)
    if kind == 2:
        return ((np.arange(n) // int(rng.integers(1, 300))) % 256).astype(np.uint8)
    if kind == 3:
        return np.full(n, int(rng.integers(256)), np.uint8)
    return (rng.integers(0, 2, n) * int(rng.integers(1, 256))).astype(np.uint8)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    selftest = len(sys.argv) > 2 and sys.argv[2] == "selftest"   # libjpeg gets another quality on odd seeds: every one of those must be reported
    lj = ljf.load()
    assert lj is not None, "no libjpeg-turbo with the IJG v8 API here"
    bad = 0
    for seed in range(n):
        rng = np.random.default_rng(seed)
        sub = [444, 422, 420][int(rng.integers(3))]
        q = int(rng.choice([int(rng.integers(1, 101)), 75, 90, 100]))
        ri = int(rng.choice([1, 2, 4, 8, 16, 32, 64]))
        if sub == 444:
            w, h = int(rng.integers(1, 500)), int(rng.integers(1, 120))
            img = content(rng, 3 * w * h).reshape(h, w, 3)
            e = hip.JpegEncoder(w, h, q, ri, subsampling=444)
            got = e.encode(torch.from_numpy(np.ascontiguousarray(img).ravel()).cuda(), L.PF_RGB)
            want = ljf.compress(lj, img, q + (1 if selftest and seed % 2 and q < 100 else 0), restart=ri)
        else:
            vs = 2 if sub == 420 else 1
            bw, bh = 2 * int(rng.integers(1, 40)), vs * int(rng.integers(1, 16 // vs + 1))   # blocks: whole MCUs
            w = 8 * bw - 2 * int(rng.integers(0, 4))                                          # the last block column / row may straddle the edge
            h = 8 * bh - int(rng.integers(0, 8))
            uyvy = content(rng, 2 * w * h)
            if sub == 420:
                y, u, v = po.uyvy_to_i420(uyvy, w, h)
            else:
                a = uyvy.reshape(h, 2 * w)
                y, u, v = a[:, 1::2], a[:, 0::4], a[:, 2::4]
            e = hip.JpegEncoder(w, h, q, ri, subsampling=sub)
            got = e.encode(torch.from_numpy(uyvy).cuda(), L.PF_UYVY)
            want = ljf.compress_planes(lj, y, u, v, w, h, sub, q, restart=ri)
        e.close()
        if ljf.scan_bytes(got) != ljf.scan_bytes(want):
            bad += 1
            print(f"seed {seed}: sub {sub} {w}x{h} q {q} restart {ri}: {len(ljf.scan_bytes(got))} against {len(ljf.scan_bytes(want))} bytes")
            if bad > 10 and not selftest:
                break
    print(f"pictures {n} mismatches {bad}")


if __name__ == "__main__":
    main()
