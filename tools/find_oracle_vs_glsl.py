#!/usr/bin/env python3
"""CPU (build container: needs /root/reference and oracle/_ref/glsl_ref): random search for a frame on which the DXT oracle differs from the
reference's OWN GLSL encoders executed on Mesa llvmpipe -- RGB / RGBA / UYVY input, DXT5-YCoCg / DXT1 / DXT1_YUV, sizes that are multiples of
4, content chosen to sit on the rounding ties and the clamps: noise, flat, extremes, two-valued blocks, ramps, low contrast.
usage: python tools/find_oracle_vs_glsl.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import pyoracle as po

PIN = {"RGB": po.IN_RGB, "RGBA": po.IN_RGBA, "UYVY": po.IN_UYVY}
BPP = {"RGB": 3, "RGBA": 4, "UYVY": 2}


def main():
    assert po.have_glsl_ref(), "oracle/_ref/glsl_ref or /root/reference missing"
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    bad = blocks = 0
    for seed in range(n):
        rng = np.random.default_rng(seed)
        fmt = ["RGB", "RGBA", "UYVY"][int(rng.integers(3))]
        mode = ["dxt5", "dxt1", "dxt1yuv"][int(rng.integers(3))] if fmt == "UYVY" else ["dxt5", "dxt1"][int(rng.integers(2))]
        w, h = 4 * int(rng.integers(1, 64)), 4 * int(rng.integers(1, 32))
        nbytes = BPP[fmt] * w * h
        kind = int(rng.integers(7))
        if kind == 0:
            src = rng.integers(0, 256, nbytes)
        elif kind == 1:
            src = np.full(nbytes, int(rng.integers(256)))
        elif kind == 2:
            src = rng.choice([0, 255, 16, 235, 240, 128, 1, 254], nbytes)
        elif kind == 3:
            a, b = int(rng.integers(256)), int(rng.integers(256))
            src = rng.choice([a, b], nbytes)                                       # two-valued: ranges of 1..255, half-way ties in the palette
        elif kind == 4:
            src = (np.arange(nbytes) // int(rng.integers(1, 50))) % 256
        elif kind == 5:
            src = np.clip(128 + int(rng.integers(1, 6)) * rng.standard_normal(nbytes), 0, 255)
        else:
            base = int(rng.integers(0, 250))
            src = base + rng.integers(0, int(rng.integers(2, 6)), nbytes)        # ranges of a few LSBs: the insets and the scale steps
        src = np.asarray(src, dtype=np.uint8)
        pin = po.IN_UYVY_RAW if mode == "dxt1yuv" else PIN[fmt]
        want = po.ref_glsl_dxt_encode(mode, fmt.lower(), src, w, h)
        got = po.dxt_encode(pin, po.OUT_DXT5YCOCG if mode == "dxt5" else po.OUT_DXT1, src, w, h, ties="even")
        blocks += w * h // 16
        if not np.array_equal(got, want):
            bs = 16 if mode == "dxt5" else 8
            d = np.flatnonzero((got.reshape(-1, bs) != want.reshape(-1, bs)).any(axis=1))
            print("MISMATCH seed", seed, fmt, mode, w, h, "kind", kind, "blocks differing", d.size, "of", w * h // 16, flush=True)
            bad += 1
            if bad >= 5:
                break
    print("frames", seed + 1, "blocks", blocks, "mismatching frames", bad)


if __name__ == "__main__":
    main()
