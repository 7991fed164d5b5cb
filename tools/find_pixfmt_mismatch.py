#!/usr/bin/env python3
"""Random search over decoders[] (every pair the compiled reference answers get_decoder_from_to() for) for a frame geometry on which the GPU
conversion differs from the reference's own line converter run on this box's CPU (oracle/_ref/libugref.so): widths around multiples of the
vector units (and anything else), 1-5 lines, source and destination pitches with 0-3 extra 16-byte words or odd paddings, both shift orders,
a pre-filled destination (bytes the reference leaves alone must stay).  GPU box.  usage: python tools/find_pixfmt_mismatch.py [n]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import pyoracle as po
from ultragrid_amd import lib as L

FILL = 0x5A
DEC = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)
NAMES = ["RGBA", "UYVY", "YUYV", "VUYA", "R10k", "R12L", "v210", "DVS10", "RGB", "BGR", "RG48", "Y216", "Y416"]


def aligned(n, align=64):
    buf = np.zeros(n + 2 * align, np.uint8)
    off = (-buf.ctypes.data) % align
    return buf[off: off + n]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    r = po.ref(scalar=True)   # the portable build: the SSE one has the RGBA->RGB tail slip
    r.get_codec_from_name.argtypes = [C.c_char_p]
    lib = L.load()
    pairs = []
    for i in NAMES:
        for o in NAMES:
            ci, co = r.get_codec_from_name(i.encode()), r.get_codec_from_name(o.encode())
            if i != o and r.get_decoder_from_to(ci, co) and lib.ug_hip_pixfmt_supported(L.PF_NAMES[i], L.PF_NAMES[o]) == 1:
                pairs.append((i, o, ci, co))
    print(len(pairs), "pairs", flush=True)
    bad = 0
    for seed in range(n):
        rng = np.random.default_rng(seed)
        i, o, ci, co = pairs[seed % len(pairs)]
        base = int(rng.choice([0, 32, 48, 64, 96, 128, 192, 256, 384, 480, 640]))
        w = max(1, base + int(rng.integers(-3, 40)))
        h = int(rng.integers(1, 6))
        sh = [(0, 8, 16), (16, 8, 0), (8, 16, 24), (24, 16, 8)][int(rng.integers(4))]
        sls, dls, dsz = r.vc_get_linesize(w, ci), r.vc_get_linesize(w, co), r.vc_get_size(w, co)
        gran = 16 if rng.random() < 0.7 else [2, 4, 8][int(rng.integers(3))]
        sp = sls + gran * int(rng.integers(0, 4)) if rng.random() < 0.6 else sls
        dp = dls + gran * int(rng.integers(0, 4)) if rng.random() < 0.6 else dls
        if i in ("RG48", "Y216", "Y416") and sp % 2: sp += 1
        if o in ("RG48", "Y216", "Y416") and dp % 2: dp += 1
        if i in ("v210", "DVS10") and sp % 4: sp += 4 - sp % 4
        if (o not in ("RGB", "R12L", "R10k", "BGR") or (o == "R10k" and i not in ("R12L", "Y416"))) and dp % 4: dp += 4 - dp % 4   # word stores, as in the reference
        src = aligned(sp * h + 128)
        src[:] = rng.integers(0, 256, src.size)
        want = aligned(dp * h + 128)
        want[:] = FILL
        fn = DEC(r.get_decoder_from_to(ci, co))
        for y in range(h):
            fn(want.ctypes.data + y * dp, src.ctypes.data + y * sp, dsz, *sh)
        dsrc = torch.from_numpy(src.copy()).cuda()
        ddst = torch.full((dp * h + 128,), FILL, dtype=torch.uint8, device="cuda")
        rc = lib.ug_hip_pixfmt_convert(L.PF_NAMES[i], L.PF_NAMES[o], dsrc.data_ptr(), ddst.data_ptr(), w, h, sp, dp, *sh, None)
        if rc != 0:
            print("REFUSED", seed, i, o, w, h, sp, dp, sh, L.last_error(), flush=True)
            bad += 1
        else:
            torch.cuda.synchronize()
            got = ddst.cpu().numpy()
            # the reference may write up to the whole last group past dst_len into the next line / the padding: compare what lies inside the lines
            rows_g = np.stack([got[y * dp: y * dp + dsz] for y in range(h)])
            rows_w = np.stack([want[y * dp: y * dp + dsz] for y in range(h)])
            if not np.array_equal(rows_g, rows_w):
                d = np.argwhere(rows_g != rows_w)
                print("MISMATCH", seed, i, o, "w", w, "h", h, "pitches", sp, dp, sh, "first (line, byte)", d[0].tolist(), "count", len(d), flush=True)
                bad += 1
        if bad >= 6:
            break
    print("frames", seed + 1, "problems", bad)


if __name__ == "__main__":
    main()
