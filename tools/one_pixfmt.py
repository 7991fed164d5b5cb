#!/usr/bin/env python3
"""One pixel-format pair repeatedly, sources and destinations rotating over >= 600 MB (for A/B runs and rocprofv3):
one_pixfmt.py IN OUT W H [frames per launch, default 1] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ultragrid_amd import lib as L

i, o, w, h = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
nb = int(sys.argv[5]) if len(sys.argv) > 5 else 1
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 0
l = L.load()
pi, po = L.PF_NAMES[i], L.PF_NAMES[o]
sls, dls = l.ug_hip_linesize(pi, w), l.ug_hip_linesize(po, w)
per = (sls + dls) * h * nb
nbuf = max(2, int(600e6 // per) + 1)
src = torch.randint(0, 256, (nbuf, nb * sls * h + 64), dtype=torch.uint8, device="cuda")
dst = torch.empty((nbuf, nb * dls * h + 64), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def run(k):
    if nb == 1:
        rc = l.ug_hip_pixfmt_convert(pi, po, src[k % nbuf].data_ptr(), dst[k % nbuf].data_ptr(), w, h, 0, 0, 0, 8, 16, st)
    else:
        rc = l.ug_hip_pixfmt_convert_batch(pi, po, src[k % nbuf].data_ptr(), dst[k % nbuf].data_ptr(), w, h, 0, 0, 0, 8, 16, nb, sls * h, dls * h, st)
    assert rc == 0, L.last_error()


for k in range(5):
    run(k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(10):
    run(k)
e1.record()
torch.cuda.synchronize()
one = max(e0.elapsed_time(e1) / 10, 1e-3)
n = iters or max(30, int(150.0 / one))
for k in range(n // 3):
    run(k)
torch.cuda.synchronize()
e0.record()
for k in range(n):
    run(k)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"{i}->{o} {w}x{h} x{nb}: {ms / nb * 1e3:.2f} us/frame, {per / (ms * 1e-3) / 1e9:.1f} GB/s, frac {per / (ms * 1e-3) / 8e12:.3f}")
