#!/usr/bin/env python3
"""Generate tests/golden/dxt_glsl_ref.npz: outputs of the reference's OWN GLSL DXT encoders
(dxt_compress/compress_dxt5ycocg_fp.glsl, compress_dxt1_fp.glsl, yuv422_to_yuv444.glsl, compress_vp.glsl) executed by Mesa llvmpipe
through oracle/_ref/glsl_ref (oracle/glsl_ref.c: headless DRI swrast context, the GL call sequence of dxt_compress/dxt_encoder.c).

Run in the build container (needs /root/reference and Mesa's swrast_dri.so):
    make -C oracle ref && python tests/golden/make_glsl_golden.py

These vectors PIN oracle/dxt_oracle.c to the reference implementation itself: tests/test_oracle_dxt.py requires the restatement --
with the two choices GLSL leaves to the implementation set the way Mesa makes them (round() ties to even, dot(vec3) summed from the
last component; the oracle's default mode, "ties even") -- to reproduce every block bit for bit, on any machine.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402
from ultragrid_amd import synth  # noqa: E402

W, H = 96, 48
CASES = [(kind, fmt) for kind in ("S1", "S2", "S3", "S4") for fmt in ("RGB", "RGBA", "UYVY") if not (kind == "S3" and fmt == "RGBA")]


def main():
    po.build()
    assert po.have_glsl_ref(), "needs oracle/_ref/glsl_ref (make -C oracle ref) and /root/reference"
    g = {"size": np.array([W, H])}
    for kind, fmt in CASES:
        src = synth.frame(kind, fmt, W, H)
        g[f"in_{kind}_{fmt}"] = src
        modes = ("dxt5", "dxt1", "dxt1yuv") if fmt == "UYVY" else ("dxt5", "dxt1")
        for mode in modes:
            g[f"out_{kind}_{fmt}_{mode}"] = po.ref_glsl_dxt_encode(mode, fmt.lower(), src, W, H)
    # one bigger uniform-random frame per input format (more exact-tie blocks), regenerated from the seed by the test
    for fmt in ("RGB", "UYVY"):
        w, h = 512, 128
        src = synth.s1_random(fmt, w, h, salt=77)
        g[f"big_{fmt}_crc"] = np.array([int(np.sum(src.astype(np.uint64) * (np.arange(src.size, dtype=np.uint64) % 251 + 1)))], np.uint64)
        for mode in ("dxt5", "dxt1"):
            g[f"big_{fmt}_{mode}"] = po.ref_glsl_dxt_encode(mode, fmt.lower(), src, w, h)
    np.savez_compressed(os.path.join(HERE, "dxt_glsl_ref.npz"), **g)
    print("wrote dxt_glsl_ref.npz:", len(g), "arrays")


if __name__ == "__main__":
    main()
