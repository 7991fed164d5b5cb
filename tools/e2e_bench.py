#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the hot path as the UltraGrid module drives it: pinned host frame -> H2D -> fused
encode kernel -> D2H of the compressed frame, DEPTH frames in flight on DEPTH streams.  This is NOT bench.py's `value` (that
one is HBM-resident); it is the number DESIGN.md section 5 quotes beside it.
usage: python tools/e2e_bench.py [--workload 8k-v210|4k-uyvy|1080p-rgb-dxt1] [--depth 3] [--frames 300]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ultragrid_amd import codec, lib, synth

WL = {"8k-v210": ("v210", lib.PF_V210, lib.DXT5_YCOCG, 7680, 4320), "4k-uyvy": ("UYVY", lib.PF_UYVY, lib.DXT5_YCOCG, 3840, 2160),
      "1080p-rgb-dxt1": ("RGB", lib.PF_RGB, lib.DXT1, 1920, 1080)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="8k-v210", choices=sorted(WL))
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--frames", type=int, default=300)
    a = ap.parse_args()
    fmt, pf, oid, w, h = WL[a.workload]
    lib.load()
    one = synth.s2_video(fmt if fmt != "RGB" else "RGB", w, 48)
    ls = one.size // 48
    host_src = torch.from_numpy(np.tile(one.reshape(48, ls), (h // 48, 1)).ravel().copy()).pin_memory()
    out_len = codec.dxt_size(oid, w, h)
    slots = []
    for _ in range(a.depth):
        slots.append(dict(st=torch.cuda.Stream(), dev_in=torch.empty(host_src.numel(), dtype=torch.uint8, device="cuda"),
                          dev_out=torch.empty(out_len, dtype=torch.uint8, device="cuda"),
                          host_out=torch.empty(out_len, dtype=torch.uint8).pin_memory(), busy=False))

    def submit(s):
        with torch.cuda.stream(s["st"]):
            s["dev_in"].copy_(host_src, non_blocking=True)
            codec.dxt_encode(pf, oid, s["dev_in"], w, h, dst=s["dev_out"])
            s["host_out"].copy_(s["dev_out"], non_blocking=True)
        s["busy"] = True

    for s in slots:          # warm-up
        submit(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.frames):
        s = slots[i % a.depth]
        if s["busy"]:
            s["st"].synchronize()
        submit(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fps = a.frames / dt
    gbs = (host_src.numel() + out_len) * fps / 1e9
    print(f"{a.workload}: {fps:.1f} fps end-to-end ({w}x{h} {fmt}, depth {a.depth}), {w * h * fps / 1e6:.0f} Mpixel/s, "
          f"PCIe traffic {gbs:.1f} GB/s ({host_src.numel() / 1e6:.1f} MB in + {out_len / 1e6:.1f} MB out per frame)")


if __name__ == "__main__":
    main()
