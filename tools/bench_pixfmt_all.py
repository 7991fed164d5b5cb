#!/usr/bin/env python3
"""Every pair of decoders[] (src/pixfmt_conv.c:3041-3103) at 7680x4320 on the GPU: ms per frame, algorithmic GB/s (input + output
line bytes) and the fraction of 8 TB/s.  Frames rotate over enough buffers to exceed the 256 MB Infinity Cache.
Usage (GPU box): python tools/bench_pixfmt_all.py [--json out.json] [--w 7680 --h 4320]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ultragrid_amd import lib as L

CODECS = ["RGBA", "UYVY", "YUYV", "RGB", "BGR", "v210", "RG48", "R10k", "R12L", "Y216", "Y416", "VUYA", "DVS10"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--w", type=int, default=7680)
    ap.add_argument("--h", type=int, default=4320)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    l = L.load()
    w, h = a.w, a.h
    rows = []
    for i in CODECS:
        for o in CODECS:
            if a.only and f"{i}->{o}" not in a.only.split(","):
                continue
            pi, po = L.PF_NAMES[i], L.PF_NAMES[o]
            if i == o or not l.ug_hip_pixfmt_supported(pi, po):
                continue
            sls, dls = l.ug_hip_linesize(pi, w), l.ug_hip_linesize(po, w)
            per = (sls + dls) * h
            nbuf = max(2, int(2.4e9 // per) + 1)   # >= 2.4 GB of rotating buffers: the flat part of profiles/r04_rotation_sweep.txt (600 MB still hit in the Infinity Cache)
            src = torch.randint(0, 256, (nbuf, sls * h + 64), dtype=torch.uint8, device="cuda")
            dst = torch.empty((nbuf, dls * h + 64), dtype=torch.uint8, device="cuda")

            def run(k):
                rc = l.ug_hip_pixfmt_convert(pi, po, src[k % nbuf].data_ptr(), dst[k % nbuf].data_ptr(), w, h, 0, 0, 0, 8, 16, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, (i, o, L.last_error())
            for k in range(3):
                run(k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(5):
                run(k)
            e1.record()
            torch.cuda.synchronize()
            one = max(e0.elapsed_time(e1) / 5, 1e-3)
            for k in range(int(40.0 / one)):      # warm-up: ~40 ms
                run(k)
            n = max(20, int(120.0 / one))         # timed: >= 120 ms of launches (20 launches of 30 us would time the clock ramp)
            torch.cuda.synchronize()
            e0.record()
            for k in range(n):
                run(k)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            gbs = per / (ms * 1e-3) / 1e9
            rows.append({"pair": f"{i}->{o}", "ms": round(ms, 4), "bytes_per_px": round((sls + dls) / w, 3), "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 3)})
            print(f"{i:>5}->{o:<5} {ms:8.4f} ms  {gbs:8.1f} GB/s  {gbs / 8000:.3f}", flush=True)
            del src, dst
    rows.sort(key=lambda r: r["frac_of_8TBps"])
    print("slowest:", [(r["pair"], r["frac_of_8TBps"]) for r in rows[:12]])
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump({"width": w, "height": h, "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
