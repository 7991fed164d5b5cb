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
# Sizes that are not multiples of 4 ("edge_*" arrays).  Width: the reference as it is (the shaders' own GL_CLAMP_TO_EDGE fetches).  GL_RGB
# lines are handed over at the stride GL reads them with (pyoracle.ref_glsl_dxt_encode(gl_row_stride=True)).  Height: the reference run on
# the picture padded to a multiple of 4 lines by repeating its last line -- see oracle/dxt_oracle.c (encode_rows) for why.
EDGE_SIZES = [(10, 8), (6, 4), (9, 8), (11, 12), (7, 4), (5, 4), (2, 4), (1, 4), (3, 8), (1366, 8), (1998, 8), (722, 4),
              (10, 6), (14, 9), (8, 7), (22, 5), (6, 1), (720, 6), (1, 1), (2, 2), (13, 3)]
EDGE_FMTS = {"RGB": 3, "RGBA": 4, "UYVY": 2}


def edge_cases():
    for w, h in EDGE_SIZES:
        for fmt in EDGE_FMTS:
            if fmt == "UYVY" and w % 2:
                continue
            yield w, h, fmt, (("dxt5", "dxt1", "dxt1yuv") if fmt == "UYVY" else ("dxt5", "dxt1"))


def edge_input(w, h, fmt):
    """seeded; half the cases noise (every block distinct), half smooth + noise (blocks like video)"""
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{w}x{h}{fmt}".encode()))
    n = w * EDGE_FMTS[fmt]
    if (w + h) % 2:
        return rng.integers(0, 256, (h, n), dtype=np.uint8)
    ramp = (np.arange(n)[None, :] * 3 + np.arange(h)[:, None] * 7) % 200 + 20
    return (ramp + rng.integers(-12, 13, (h, n))).astype(np.uint8)


def pad_lines(src, h):
    a = src.reshape(h, -1)
    return np.ascontiguousarray(np.concatenate([a, np.repeat(a[-1:], (h + 3) // 4 * 4 - h, axis=0)], axis=0))


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
    g["edge_sizes"] = np.array(EDGE_SIZES)
    for w, h, fmt, modes in edge_cases():
        src = edge_input(w, h, fmt)
        g[f"edge_in_{w}x{h}_{fmt}"] = src
        for mode in modes:
            g[f"edge_out_{w}x{h}_{fmt}_{mode}"] = po.ref_glsl_dxt_encode(mode, fmt.lower(), pad_lines(src, h), w, (h + 3) // 4 * 4, gl_row_stride=True)
    # The reference AS IT IS where this implementation deliberately does something else (tests/test_oracle_dxt.py::test_reference_*_slip):
    #   a height that is not a multiple of 4 (vertical resampling + an unrendered last block row), packed RGB lines with 3 w % 4 != 0
    #   (GL reads them at a 4-byte-aligned stride: a skewed picture)
    for w, h in ((16, 10), (8, 6), (8, 486)):
        g[f"slip_h_{w}x{h}_RGBA_dxt5"] = po.ref_glsl_dxt_encode("dxt5", "rgba", edge_input(w, h, "RGBA"), w, h)
    for w, h in ((10, 8), (6, 8)):
        g[f"slip_rgb_{w}x{h}_dxt1"] = po.ref_glsl_dxt_encode("dxt1", "rgb", edge_input(w, h, "RGB"), w, h)
    np.savez_compressed(os.path.join(HERE, "dxt_glsl_ref.npz"), **g)
    print("wrote dxt_glsl_ref.npz:", len(g), "arrays")


if __name__ == "__main__":
    main()
