#!/usr/bin/env python3
"""Launch one DXT encode configuration repeatedly (for rocprofv3 --pmc / --kernel-trace): one_kernel.py IN OUT W H FRAMES [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultragrid_amd import codec, lib as L
from tools.bench_kernels import frames, timeit

fmt, out, w, h, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 50
L.load()
src = frames(fmt, w, h, n)
oid = {"DXT1": L.DXT1, "DXT5": L.DXT5_YCOCG}[out]
pf = L.PF_NAMES[fmt]
dst = torch.empty(codec.dxt_size(oid, w, h) * n, dtype=torch.uint8, device="cuda")


def run():
    codec.dxt_encode_batch(pf, oid, src, w, h, n, src.shape[1], dst=dst)


ms = timeit(run, iters=iters)
print(f"{fmt}->{out} {w}x{h}x{n}: {ms:.4f} ms/launch, {w * h * n / ms / 1e3:.1f} Mpx/s")
