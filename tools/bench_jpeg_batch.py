#!/usr/bin/env python3
"""ug_hip_jpeg_encoder_encode (one frame per call) against ug_hip_jpeg_encoder_encode_batch (n frames per call), 3840x2160, device-resident
input, per-frame time incl. the synchronisation each call ends with.  usage: python tools/bench_jpeg_batch.py [--sub 420] [--n 8]"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ultragrid_amd import lib as L, synth

ap = argparse.ArgumentParser()
ap.add_argument("--sub", type=int, default=420, help="420 / 422: UYVY input; 444: RGB input (R, G, B components, gpujpeg.cpp:303-305)")
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--seconds", type=float, default=1.0)
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--q", type=int, default=75)
ap.add_argument("--ri", type=int, default=4, help="restart interval (MCUs)")
ap.add_argument("--only", choices=["both", "batch", "single"], default="both", help="profile runs: one call form only, so that per-kernel averages are not a blend")
ap.add_argument("--calls", type=int, default=0, help="exactly this many timed calls per leg instead of --seconds (counter passes)")
a = ap.parse_args()
l = L.load()
w, h = (int(x) for x in a.size.split("x"))
rgb = a.sub == 444
fmt_in, pf_in, line = ("RGB", L.PF_RGB, 3 * w) if rgb else ("UYVY", L.PF_UYVY, 2 * w)
base = torch.from_numpy(synth.s2_video("UYVY", w, h) if not rgb else synth.frame("S2", "RGB", w, h)).cuda()
sets = 4
src = torch.stack([torch.stack([torch.roll(base, line * 37 * (f + a.n * s)) for f in range(a.n)]) for s in range(sets)])   # (sets, n, bytes): 4 x n distinct frames
enc = C.c_void_p()
assert l.ug_hip_jpeg_encoder_create_sub(w, h, a.q, a.ri, a.sub, C.byref(enc)) == 0
cap = l.ug_hip_jpeg_encoder_max_size(enc)
stride = (min(cap, w * h * 3 + 4096) + 15) // 16 * 16
out = torch.empty((a.n, stride), dtype=torch.uint8, device="cuda")
lens = (C.c_size_t * a.n)()
st = torch.cuda.current_stream().cuda_stream
one = C.c_size_t(0)


def single():
    k = single.k = getattr(single, "k", 0) + 1
    for f in range(a.n):
        assert l.ug_hip_jpeg_encoder_encode(enc, pf_in, src[k % sets, f].data_ptr(), 0, out[f].data_ptr(), stride, C.byref(one), st) == 0


def batch():
    k = batch.k = getattr(batch, "k", 0) + 1
    assert l.ug_hip_jpeg_encoder_encode_batch(enc, pf_in, a.n, src[k % sets].data_ptr(), 0, src.shape[2], out.data_ptr(), stride, stride, lens, st) == 0, L.last_error()
    assert all(lens[f] <= stride for f in range(a.n)), "a stream did not fit its slice"


legs = (("one frame per call", single), (f"{a.n} frames per call", batch), ("one frame per call", single), (f"{a.n} frames per call", batch))
if a.only == "batch":
    legs = legs[1::2]
elif a.only == "single":
    legs = legs[0::2]
for name, fn in legs:
    for _ in range(2 if a.calls else 5):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while (n < a.calls) if a.calls else (time.perf_counter() - t0 < a.seconds):
        fn()
        n += 1
    dt = time.perf_counter() - t0
    print(f"jpeg encode {w}x{h} {fmt_in} 4:{str(a.sub)[1:2]}:{str(a.sub)[2:]} q{a.q} restart {a.ri}, {name}: {dt / (n * a.n) * 1e6:.1f} us per frame ({n * a.n / dt:.0f} fps), stream {lens[0] or one.value} B")

l.ug_hip_jpeg_encoder_destroy(enc)
