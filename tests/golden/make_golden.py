#!/usr/bin/env python3
"""Generate tests/golden/*.npz.

Run in the build container (needs /root/reference to build oracle/_ref):
    python tests/golden/make_golden.py

  pixfmt_ref.npz  -- inputs + outputs of the reference's OWN compiled C
                     (oracle/_ref/libugref.so: get_decoder_from_to() line loop, uyvy_to_i420,
                     v210_to_p010le, get_color_coeffs).  These pin oracle/pixfmt_oracle.c and
                     the HIP pixfmt kernels wherever /root/reference is absent (GPU box).
  dxt_oracle.npz  -- outputs of oracle/dxt_oracle.c ("parity unpinned": the reference has no
                     CPU DXT encoder and no DXT known-answer test); regression anchor only.
  jpeg_oracle.npz -- outputs of oracle/jpeg_oracle.c (same status).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402
from ultragrid_amd import synth  # noqa: E402

PAIRS = [("v210", "UYVY"), ("YUYV", "UYVY"), ("UYVY", "YUYV"), ("UYVY", "RGB"), ("UYVY", "RGBA"), ("RGB", "UYVY"),
         ("BGR", "UYVY"), ("RGBA", "UYVY"), ("RG48", "UYVY"), ("v210", "RGB"), ("v210", "RG48"), ("RGBA", "RGB"),
         ("RGB", "RGBA"), ("RGBA", "RGBA"), ("RGB", "RGB"), ("BGR", "RGB"), ("UYVY", "v210")]
SIZES = [(48, 4), (50, 3), (96, 2)]
SHIFTS = [(0, 8, 16), (16, 8, 0)]


def main():
    po.build()
    assert po.have_ref(), "needs /root/reference (oracle/_ref)"
    g = {}
    for d in (0, 8, 10, 12, 16):
        g[f"coeffs_d{d}"] = np.array(po.ref_color_coeffs(d), np.int32)
    for pi, (i, o) in enumerate(PAIRS):
        for (w, h) in SIZES:
            src = synth.s1_random(i, w, h, salt=1000 * pi + w)
            g[f"in_{i}_{o}_{w}x{h}"] = src
            for sh in SHIFTS:
                # RGBA->RGB: the SSSE3 build has a line-tail bug (see oracle/Makefile); golden = portable path
                scalar = (i, o) == ("RGBA", "RGB")
                g[f"out_{i}_{o}_{w}x{h}_{sh[0]}_{sh[1]}_{sh[2]}"] = po.ref_convert_frame(i, o, src, w, h, sh, scalar=scalar)
    for (w, h) in [(1, 2), (2, 1), (16, 1), (16, 16), (127, 255), (64, 6)]:
        src = synth.s1_random("UYVY", w, h, salt=w * 1000 + h)
        y, u, v = po.uyvy_to_i420(src, w, h, use_ref=True)
        g[f"i420_in_{w}x{h}"] = src
        g[f"i420_y_{w}x{h}"], g[f"i420_u_{w}x{h}"], g[f"i420_v_{w}x{h}"] = y, u, v
    for (w, h) in [(48, 2), (96, 4)]:
        src = synth.s1_random("v210", w, h, salt=7)
        y, uv = po.v210_to_p010le(src, w, h, use_ref=True)
        g[f"p010_in_{w}x{h}"] = src
        g[f"p010_y_{w}x{h}"], g[f"p010_uv_{w}x{h}"] = y, uv
    np.savez_compressed(os.path.join(HERE, "pixfmt_ref.npz"), **g)

    d = {}
    w, h = 48, 16
    fmts = {"RGB": po.IN_RGB, "RGBA": po.IN_RGBA, "UYVY": po.IN_UYVY, "v210": po.IN_V210}
    for kind in ("S1", "S2", "S4"):
        for name, fid in fmts.items():
            src = synth.frame(kind, name, w, h)
            d[f"in_{kind}_{name}"] = src
            for oname, oid in (("dxt1", po.OUT_DXT1), ("dxt5ycocg", po.OUT_DXT5YCOCG)):
                d[f"out_{kind}_{name}_{oname}"] = po.dxt_encode(fid, oid, src, w, h)
                d[f"outm_{kind}_{name}_{oname}"] = po.dxt_encode(fid, oid, src, w, -h)
    np.savez_compressed(os.path.join(HERE, "dxt_oracle.npz"), **d)

    j = {}
    w, h = 40, 24
    src = synth.s2_video("UYVY", w, h)
    y, u, v = po.uyvy_to_i420(src, w, h)
    j["in_uyvy"] = src
    for q in (50, 75, 90):
        for comp, plane in ((0, y), (1, u)):
            div = po.jpeg_divisors(po.jpeg_qtable(q, comp))
            out, coef = po.jpeg_fdct_quant_plane(plane, div, want_coef=True)
            j[f"q{q}_c{comp}_out"] = out
            j[f"q{q}_c{comp}_coef"] = coef
            j[f"q{q}_c{comp}_qtable"] = po.jpeg_qtable(q, comp)
    np.savez_compressed(os.path.join(HERE, "jpeg_oracle.npz"), **j)
    for f in ("pixfmt_ref.npz", "dxt_oracle.npz", "jpeg_oracle.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
