#!/usr/bin/env python3
"""Generate tests/golden/lavc_ref.npz and pixfmt_ext_ref.npz from the reference's OWN functions (compiled from /root/reference:
oracle/_ref/libugref_lavc.so, libugref.so; `make -C oracle ref`).  Run where /root/reference exists:  python tests/golden/make_lavc_golden.py
The GPU tests compare the HIP kernels with these vectors even on a box where oracle/_ref is absent."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C

import test_lavc_conv as T
import test_gpu_pixfmt_ext as X
from oracle import pyoracle as po

SIZES = [(48, 8), (50, 6)]
out = {}
r = T.ref()
for uv, av in T.TO_AV:
    for (w, h) in SIZES:
        if (uv, av) in T.FORWARDED_TO and w % 48:
            continue
        ls = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
        src = np.random.default_rng(w + h).integers(0, 256, ls * h + 64).astype(np.uint8)
        planes = T.ref_uv_to_av(uv, av, src, w, h)
        key = f"to|{uv}|{av}|{w}x{h}"
        out[key + "|in"] = src
        for k, p in enumerate(planes):
            out[key + f"|p{k}"] = p
for av, uv in T.FROM_AV:
    for i, ((w, h), cs, rng_) in enumerate([((48, 8), 1, 1), ((50, 6), 5, 2)]):
        if av.startswith("yuvj"):
            rng_ = 2
        frp = T.make_frame(r, av, w, h, 40 + i, cs, rng_)
        pitch = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
        want = T.ref_av_to_uv(frp, av, uv, w, h, pitch, (0, 8, 16))
        key = f"from|{av}|{uv}|{w}x{h}|{cs}|{rng_}"
        for k, p in enumerate(T.plane_arrays(r, frp.contents, h)):
            out[key + f"|p{k}"] = p.copy()
        out[key + "|out"] = want
        r.av_frame_free(C.byref(frp))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lavc_ref.npz"), **out)
print("lavc_ref.npz:", len(out), "arrays")

out = {}
for i, o in X.PAIRS + [("DVS10", "UYVY")]:
    for (w, h) in [(48, 4), (50, 3)]:
        if (i, o) == ("DVS10", "UYVY") and w % 48:
            continue
        rr = po.ref()
        rr.get_codec_from_name.argtypes = [C.c_char_p]
        sls = rr.vc_get_linesize(w, rr.get_codec_from_name(i.encode()))
        src = X.aligned(sls * h + 64, rng=np.random.default_rng(w))
        want, sls, dls = X.ref_frame(po, i, o, src, w, h, (0, 8, 16))
        out[f"{i}|{o}|{w}x{h}|in"] = src.copy()
        out[f"{i}|{o}|{w}x{h}|out"] = want.copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pixfmt_ext_ref.npz"), **out)
print("pixfmt_ext_ref.npz:", len(out), "arrays")
