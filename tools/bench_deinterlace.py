#!/usr/bin/env python3
"""ug_hip_deinterlace_blend / _batch on the GPU: one frame per launch and eight (csrc/deinterlace.hip).  In place; algorithmic
bytes = 2 x the frame (every line read once, written once)."""
import os, sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ultragrid_amd import lib as L

l = L.load()
for (w, h, bpp, name) in ((1920, 1080, 2, "1080i UYVY"), (1920, 1080, 3, "1080i RGB"), (1920, 1080, 4, "1080i RGBA"), (720, 576, 2, "576i UYVY"), (3840, 2160, 2, "2160-line UYVY")):
    ls = w * bpp
    n = 8
    # 40 sets of 8 frames: the rotation keeps the Infinity Cache from answering (tools/rotation_sweep.py)
    sets = max(2, int(2.4e9 // (n * ls * h)))
    buf = torch.randint(0, 256, (sets, n, ls * h), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        l.ug_hip_deinterlace_blend_batch(buf[0].data_ptr(), ls, h, n, ls * h, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    calls = 2 * sets
    e0.record()
    for i in range(calls):
        l.ug_hip_deinterlace_blend_batch(buf[i % sets].data_ptr(), ls, h, n, ls * h, st)
    e1.record(); torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / calls / n * 1e3
    e0.record()
    for i in range(calls):
        l.ug_hip_deinterlace_blend(buf[i % sets][i % n].data_ptr(), ls, h, st)
    e1.record(); torch.cuda.synchronize()
    t1 = e0.elapsed_time(e1) / calls * 1e3
    print(f"deinterlace {name}: {t1:.1f} us one frame per launch ({2 * ls * h / t1 / 1e3:.0f} GB/s), {tb:.2f} us per frame at 8 per launch ({2 * ls * h / tb / 1e3:.0f} GB/s = {2 * ls * h / tb / 1e3 / 8000:.3f} of 8 TB/s)")
