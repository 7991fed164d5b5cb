#!/usr/bin/env python3
"""Rotation sweep: do the table rows that sit above ~0.79 of 8 TB/s owe part of their rate to the 256 MB Infinity Cache?
Each row is timed with its sources AND its destinations rotating over R bytes PER SIDE, R in {0.15, 0.3, 0.6, 1.2, 2.4, 4.8} GB (round 3's tables used 0.6 GB in
total = 0.3 GB per side).  If the cache replaced at random, a working set of R + R bytes would still hit in ~256 MB / 2R of its accesses; a rate that is flat in R
says the tables' figure is the HBM figure.  HIP events on the launch stream, >= 150 ms of launches per point, points interleaved A B C D A B C D.
usage (GPU box): python tools/rotation_sweep.py > gpurun_out/rotation_sweep.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ultragrid_amd import lib as L

l = L.load()
st = torch.cuda.current_stream().cuda_stream
SIZES = [0.15e9, 0.3e9, 0.6e9, 1.2e9, 2.4e9, 4.8e9]


def timeit(fn, min_ms=150.0):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1) / 3, 1e-3)
    n = max(10, int(min_ms / one))
    for _ in range(n // 3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def row(name, in_bytes, out_bytes, call):
    """call(src_ptr, dst_ptr); in_bytes / out_bytes per launch"""
    res = []
    for rep in range(2):          # interleaved: every size twice, best kept; buffers of one point at a time
        for R in SIZES:
            sets = max(2, int(R // min(in_bytes, out_bytes)) + 1)
            src = torch.randint(0, 256, (sets, in_bytes), dtype=torch.uint8, device="cuda")
            dst = torch.empty((sets, out_bytes), dtype=torch.uint8, device="cuda")
            k = [0]

            def fn():
                j = k[0] % sets
                k[0] += 1
                call(src[j].data_ptr(), dst[j].data_ptr())
            ms = timeit(fn)
            res.append((R, ms))
            del src, dst
            torch.cuda.empty_cache()
    best = {R: min(ms for r, ms in res if r == R) for R in SIZES}
    line = f"{name:34s}"
    for R in SIZES:
        gbs = (in_bytes + out_bytes) / (best[R] * 1e-3) / 1e9
        line += f"  {R / 1e9:.2f} GB/side: {best[R]:7.4f} ms {gbs:7.1f} GB/s {gbs / 8000:5.3f}"
    print(line, flush=True)


w, h, nb = 3840, 2160, 8
for (i, o) in (("UYVY", "v210"), ("UYVY", "RGB"), ("UYVY", "RGBA"), ("v210", "UYVY")):
    sls, dls = l.ug_hip_linesize(L.PF_NAMES[i], w), l.ug_hip_linesize(L.PF_NAMES[o], w)

    def call(s, d, i=i, o=o, sls=sls, dls=dls):
        rc = l.ug_hip_pixfmt_convert_batch(L.PF_NAMES[i], L.PF_NAMES[o], s, d, w, h, 0, 0, 0, 8, 16, nb, sls * h, dls * h, st)
        assert rc == 0, L.last_error()
    row(f"pixfmt {i}->{o} 4K (batch of 8)", nb * sls * h, nb * dls * h, call)

w8, h8 = 7680, 4320
sls, dls = l.ug_hip_linesize(L.PF_UYVY, w8), l.ug_hip_linesize(L.PF_RGB, w8)
row("pixfmt UYVY->RGB 8K", sls * h8, dls * h8, lambda s, d: l.ug_hip_pixfmt_convert(L.PF_UYVY, L.PF_RGB, s, d, w8, h8, 0, 0, 0, 8, 16, st))
# DXT1 -> RGBA, 8 frames of 4K per launch (one tall picture: the decoder has no batch entry point and needs none)
row("dxt_decode DXT1->RGBA 4K x8", nb * w * h // 2, nb * w * h * 4,
    lambda s, d: l.ug_hip_dxt_decode(L.DXT1, L.PF_RGBA, s, d, w, h * nb, 0, 0, 8, 16, st))
row("dxt_decode DXT5->RGBA 4K x8", nb * w * h, nb * w * h * 4,
    lambda s, d: l.ug_hip_dxt_decode(L.DXT5_YCOCG, L.PF_RGBA, s, d, w, h * nb, 0, 0, 8, 16, st))
