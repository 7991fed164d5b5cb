#!/usr/bin/env python3
"""Random search for a DXT encode / decode call on which the GPU differs from the oracle: input formats x outputs, ANY width and height (round 6:
half the draws are not multiples of 4 -- 4:2:2 formats keep an even width; v210 widths that are not a multiple of 12 included), padded pitches, bottom-up sources, both tie rules, content from
flat over gradients to noise and extreme values; every encoded frame is also decoded on both sides.  GPU box.
usage: python tools/find_dxt_mismatch.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import pyoracle as po
from ultragrid_amd import codec as hip, lib as L

COMBOS = [("RGB", po.IN_RGB, L.PF_RGB, 3), ("RGBA", po.IN_RGBA, L.PF_RGBA, 4), ("UYVY", po.IN_UYVY, L.PF_UYVY, 2), ("v210", po.IN_V210, L.PF_V210, 0)]
OUTS = [(po.OUT_DXT1, L.DXT1), (po.OUT_DXT5YCOCG, L.DXT5_YCOCG)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    bad = 0
    for seed in range(n):
        rng = np.random.default_rng(seed)
        name, pin, pf, bpp = COMBOS[int(rng.integers(len(COMBOS)))]
        pout, oid = OUTS[int(rng.integers(2))]
        w, h = 4 * int(rng.integers(1, 90)), 4 * int(rng.integers(1, 12))
        if rng.random() < 0.5:                                                # any size (dxt_util.h:59-67): the EDGE instantiations
            w, h = max(1, w - int(rng.integers(0, 4))), max(1, h - int(rng.integers(0, 4)))
            if name in ("UYVY", "v210"):
                w += w & 1
        ties = ["even", "away"][int(rng.integers(2))]
        line = (w + 47) // 48 * 128 if name == "v210" else bpp * w
        line = line if name != "UYVY" else 2 * w
        pitch = line + 16 * int(rng.integers(0, 3)) + (0 if name != "RGB" or not ((w & 3) or (h & 3)) else int(rng.integers(0, 4)))   # (3 w bytes per line: any pitch, when the EDGE form runs)
        kind = int(rng.integers(5))
        buf = np.zeros(pitch * h + 64, np.uint8)
        if kind == 0:
            buf[:] = rng.integers(0, 256, buf.size)                          # noise
        elif kind == 1:
            buf[:] = int(rng.integers(256))                                   # flat
        elif kind == 2:
            buf[:] = rng.choice([0, 255, 16, 235, 128], buf.size)            # extremes
        elif kind == 3:
            buf[:] = (np.arange(buf.size) // int(rng.integers(1, 40))) % 256  # ramps
        else:
            buf[:] = np.clip(128 + 20 * rng.standard_normal(buf.size), 0, 255)   # low contrast
        if name == "v210":
            buf.view(np.uint32)[:] &= 0x3FFFFFFF
        mirror = rng.random() < 0.3
        hh = -h if mirror else h
        want = po.dxt_encode(pin, pout, buf[: pitch * h], w, hh, pitch=pitch, ties=ties)
        got = hip.dxt_encode(pf, oid, torch.from_numpy(buf).cuda(), w, hh, pitch=pitch, ties=L.TIES_EVEN if ties == "even" else L.TIES_AWAY).cpu().numpy()
        if not np.array_equal(got, want):
            print("ENCODE MISMATCH", seed, name, "DXT1" if oid == L.DXT1 else "DXT5", w, hh, pitch, ties, kind, "bytes differing", int((got != want).sum()), flush=True)
            bad += 1
        if seed % 3 == 0 and po.dxt_size(pout, w, h) % 16 == 0:   # the batched entry point (frames start on 16-byte boundaries on both sides): 2-3 frames a stride apart, each must equal its own single-frame result
            frames = int(rng.integers(2, 4))
            stride = (pitch * h + 15) // 16 * 16 + 16 * int(rng.integers(0, 5))   # frames start on 16-byte boundaries (the API asks for it)
            big = np.zeros(stride * frames + 64, np.uint8)
            singles = []
            for f in range(frames):
                fr = np.roll(buf[: pitch * h], 97 * f)
                if name == "v210":
                    fr = fr.copy(); fr.view(np.uint32)[:] &= 0x3FFFFFFF
                big[f * stride: f * stride + pitch * h] = fr
                singles.append(po.dxt_encode(pin, pout, fr, w, hh, pitch=pitch, ties=ties))
            gb = hip.dxt_encode_batch(pf, oid, torch.from_numpy(big).cuda(), w, hh, frames, stride, pitch=pitch,
                                      ties=L.TIES_EVEN if ties == "even" else L.TIES_AWAY).cpu().numpy()
            if not np.array_equal(gb, np.concatenate(singles)):
                print("BATCH MISMATCH", seed, name, w, hh, pitch, frames, stride, flush=True)
                bad += 1
        for out in ("RGBA", "RGB", "UYVY"):
            if out == "UYVY" and w % 2:
                continue
            dw = po.dxt_decode(pout, out, want, w, h, ties=ties)
            dg = hip.dxt_decode(oid, L.PF_NAMES[out], torch.from_numpy(want).cuda(), w, h, ties=L.TIES_EVEN if ties == "even" else L.TIES_AWAY).cpu().numpy()
            if not np.array_equal(dg, dw):
                print("DECODE MISMATCH", seed, "DXT1" if oid == L.DXT1 else "DXT5", out, w, h, ties, flush=True)
                bad += 1
        if bad >= 6:
            break
    print("calls", seed + 1, "problems", bad)


if __name__ == "__main__":
    main()
