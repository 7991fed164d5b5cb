#!/usr/bin/env python3
"""Time the reference's OWN CPU line converters (oracle/_ref/libugref.so, compiled from /root/reference with its -O3 -msse4.1)
on this box's host cores, one thread, one 4K frame per conversion -- the CPU side of SURVEY.md 8(d) for the pixfmt rows
("kind": "reference").  usage: python tools/cpu_reference_bench.py [--json out.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import pyoracle as po
from ultragrid_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    assert po.have_ref(), "oracle/_ref/libugref.so missing"
    w, h = 3840, 2160
    rows = []
    for i, o in [("v210", "UYVY"), ("UYVY", "RGB"), ("UYVY", "RGBA"), ("RGB", "UYVY"), ("RGBA", "UYVY"), ("v210", "RGB"), ("RGBA", "RGB"), ("RGB", "RGBA"),
                 ("UYVY", "YUYV"), ("UYVY", "v210")]:
        src = synth.s1_random(i, w, h, salt=1)
        po.ref_convert_frame(i, o, src, w, h)          # warm
        n, t0 = 0, time.perf_counter()
        while n < 3 or time.perf_counter() - t0 < 1.0:
            po.ref_convert_frame(i, o, src, w, h)
            n += 1
        ms = (time.perf_counter() - t0) / n * 1e3
        rows.append({"conversion": f"{i}->{o}", "size": f"{w}x{h}", "ms_per_frame_1core": round(ms, 3), "Mpix_s_1core": round(w * h / ms / 1e3, 1)})
        print(f"{i:>5s} -> {o:<5s} {ms:8.3f} ms/frame/core  {w * h / ms / 1e3:8.1f} Mpx/s", flush=True)
    src = synth.s1_random("UYVY", w, h, salt=2)
    n, t0 = 0, time.perf_counter()
    while n < 3 or time.perf_counter() - t0 < 1.0:
        po.uyvy_to_i420(src, w, h, use_ref=True)
        n += 1
    ms = (time.perf_counter() - t0) / n * 1e3
    rows.append({"conversion": "uyvy_to_i420", "size": f"{w}x{h}", "ms_per_frame_1core": round(ms, 3), "Mpix_s_1core": round(w * h / ms / 1e3, 1)})
    print(f"uyvy_to_i420   {ms:8.3f} ms/frame/core", flush=True)
    if args.json:
        json.dump({"cores_used": 1, "host_cpus": os.cpu_count(), "rows": rows}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
