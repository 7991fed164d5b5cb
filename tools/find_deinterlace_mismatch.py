#!/usr/bin/env python3
"""Random search for a frame on which ug_hip_deinterlace_blend[_batch] differs from the oracle's vc_deinterlace (which is pinned to the compiled
reference): line sizes from 16 bytes (below: refused, see csrc/deinterlace.hip) to 8K RGBA (multiples of 16, of 4 only, odd ones: the byte kernel and the column that reaches into the next
line), heights 0 ... 2 400 (no step, one active wave, every split of the steps over 16 waves, several rounds), 1 ... 5 frames per launch with
gaps between them, content that sits on the thresholds of the segment summaries (two-valued, nearly flat, ramps) besides noise; the bytes
between and behind the frames must stay as they were.  GPU box.   usage: python tools/find_deinterlace_mismatch.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import pyoracle as po
from ultragrid_amd import lib as L


def content(rng, n):
    kind = int(rng.integers(6))
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:
        return (rng.integers(0, 2, n, dtype=np.uint8) * 255).astype(np.uint8)
    if kind == 2:
        return np.clip(int(rng.integers(256)) + rng.integers(-1, 2, n), 0, 255).astype(np.uint8)
    if kind == 3:
        return ((np.arange(n) // int(rng.integers(1, 5000))) % 256).astype(np.uint8)
    if kind == 4:
        return rng.choice(np.array([0, 1, 127, 128, 254, 255], np.uint8), n)
    return np.full(n, int(rng.integers(256)), np.uint8)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    l = L.load()
    bad = 0
    for seed in range(n):
        rng = np.random.default_rng(seed)
        shape = int(rng.integers(4))
        if shape == 0:
            ls = 16 * int(rng.integers(1, 1921))            # whole 16-byte columns, up to 8K RGBA
        elif shape == 1:
            ls = 4 * int(rng.integers(4, 2000))
        else:
            ls = int(rng.integers(16, 6000))
        lines = int(rng.choice([int(rng.integers(0, 40)), int(rng.integers(40, 1200)), int(rng.integers(1080, 1100)), int(rng.integers(1200, 2400))]))
        if ls * lines > 24_000_000:
            lines = 24_000_000 // ls
        frames = int(rng.integers(1, 6)) if ls * lines < 4_000_000 else 1
        gap = int(rng.choice([0, 0, 16, 20, 4096])) if ls % 4 == 0 else int(rng.integers(0, 50))
        stride = ls * lines + gap
        if ls % 4 == 0 and stride % 4:
            stride += 4 - stride % 4
        host = content(rng, frames * stride + 64)
        dev = torch.from_numpy(host.copy()).cuda()
        if frames == 1 and rng.random() < 0.5:
            rc = l.ug_hip_deinterlace_blend(dev.data_ptr(), ls, lines, None)
        else:
            rc = l.ug_hip_deinterlace_blend_batch(dev.data_ptr(), ls, lines, frames, stride, None)
        got = dev.cpu().numpy()
        want = host.copy()
        for f in range(frames):
            # the oracle (like the reference) reads the 16-byte column that hangs over the last line's end: the bytes behind the frame are in `host`
            seg = host[f * stride: f * stride + ls * lines + 64].copy()
            want[f * stride: f * stride + ls * lines] = po.deinterlace_blend(seg, ls, lines)[: ls * lines] if lines else seg[:0]
        if rc != 0 or not np.array_equal(got, want):
            bad += 1
            where = np.flatnonzero(got != want)
            print(f"seed {seed}: rc {rc} linesize {ls} lines {lines} frames {frames} stride {stride}: {where.size} bytes differ, first at {where[:3]}")
            if bad > 10:
                break
    print(f"frames {n} problems {bad}")


if __name__ == "__main__":
    main()
