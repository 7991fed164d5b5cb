#!/usr/bin/env python3
"""What the EDGE instantiations cost (round 6): the DXT encoder / decoders on a frame that is a few pixels short of a multiple of 4 against the same
frame at the multiple, 16 frames per launch (encoder) / one launch per frame (decoders), rotating buffers.  GPU box.  usage: python tools/bench_edge.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ultragrid_amd import codec as hip, lib as L


def enc(pf, bpp_line, w, h, frames=16, iters=200):
    ls = bpp_line(w)
    fb = (ls * h + 15) // 16 * 16
    src = torch.randint(0, 256, (4, frames * fb), dtype=torch.uint8, device="cuda")
    if pf == L.PF_V210:
        src = (src.view(torch.int32) & 0x3FFFFFFF).view(torch.uint8)
    per = (hip.dxt_size(L.DXT5_YCOCG, w, h) + 15) // 16 * 16
    dst = torch.empty((4, frames * per), dtype=torch.uint8, device="cuda")
    l = L.load()
    st = torch.cuda.current_stream().cuda_stream
    for b in range(4):
        assert l.ug_hip_dxt_encode_batch(pf, L.DXT5_YCOCG, src[b].data_ptr(), dst[b].data_ptr(), w, h, ls, frames, fb, per, st) == 0, L.last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        l.ug_hip_dxt_encode_batch(pf, L.DXT5_YCOCG, src[i % 4].data_ptr(), dst[i % 4].data_ptr(), w, h, ls, frames, fb, per, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / frames * 1e3   # us per frame


def dec(out, w, h, iters=300):
    n = hip.dxt_size(L.DXT5_YCOCG, w, h)
    src = torch.randint(0, 256, (8, (n + 15) // 16 * 16), dtype=torch.uint8, device="cuda")
    ls = hip.linesize(out, w)
    dst = torch.empty((8, (ls * h + 79) // 16 * 16), dtype=torch.uint8, device="cuda")
    l = L.load()
    st = torch.cuda.current_stream().cuda_stream
    for b in range(8):
        assert l.ug_hip_dxt_decode(L.DXT5_YCOCG, out, src[b].data_ptr(), dst[b].data_ptr(), w, h, 0, 0, 8, 16, st) == 0, L.last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        l.ug_hip_dxt_decode(L.DXT5_YCOCG, out, src[i % 8].data_ptr(), dst[i % 8].data_ptr(), w, h, 0, 0, 8, 16, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    print("DXT5-YCoCg encoder, 16 frames per launch, us per frame: multiple of 4 | 2 columns and 2 lines short (EDGE) | ratio")
    for name, pf, line in (("UYVY", L.PF_UYVY, lambda w: 2 * w), ("RGB", L.PF_RGB, lambda w: 3 * w), ("RGBA", L.PF_RGBA, lambda w: 4 * w), ("v210", L.PF_V210, lambda w: (w + 47) // 48 * 128)):
        for w, h in ((3840, 2160), (1920, 1080), (1368, 768)):
            a, b = enc(pf, line, w, h), enc(pf, line, w - 2, h - 2)
            print(f"  {name:5s} {w}x{h}: {a:7.2f} | {w - 2}x{h - 2}: {b:7.2f} | {b / a:.3f}")
    print("DXT5-YCoCg decoders, one frame per launch, us per frame")
    for name, out in (("RGBA", L.PF_RGBA), ("UYVY", L.PF_UYVY), ("RGB", L.PF_RGB)):
        for w, h in ((3840, 2160), (1368, 768)):
            a, b, c = dec(out, w, h), dec(out, w - 2, h - 2), dec(out, w, h - 2)
            print(f"  ->{name:5s} {w}x{h}: {a:7.2f} | {w - 2}x{h - 2}: {b:7.2f} | {b / a:.3f} | {w}x{h - 2} (lines keep their alignment): {c:7.2f} | {c / a:.3f}")


if __name__ == "__main__":
    main()
