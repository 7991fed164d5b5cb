#!/usr/bin/env python3
"""Time the reference's OWN CPU line converters (oracle/_ref/libugref.so, compiled from /root/reference with its -O3 -msse4.1)
on this box's host cores, one 4K frame per conversion -- the CPU side of SURVEY.md 8(d) for the pixfmt rows ("kind": "reference"):
  (i)  one thread, the per-line decoder_t loop of tools/convert.cpp:43-48;
  (ii) all host cores by even row bands with the reference's OWN parallel_pix_conv() (src/utils/parallel_conv.c:64-85, compiled into
       libugref.so): the thread count with the best time out of 1, 2, 4, ... , all visible CPUs is reported with its count.
usage: python tools/cpu_reference_bench.py [--json out.json]"""
import ctypes as C
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
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for i, o in [("v210", "UYVY"), ("UYVY", "RGB"), ("UYVY", "RGBA"), ("RGB", "UYVY"), ("RGBA", "UYVY"), ("v210", "RGB"), ("RGBA", "RGB"), ("RGB", "RGBA"),
                 ("UYVY", "YUYV"), ("UYVY", "v210")]:
        src = synth.s1_random(i, w, h, salt=1)
        po.ref_convert_frame(i, o, src, w, h)          # warm
        n, t0 = 0, time.perf_counter()
        while n < 3 or time.perf_counter() - t0 < 1.0:
            po.ref_convert_frame(i, o, src, w, h)
            n += 1
        ms = (time.perf_counter() - t0) / n * 1e3
        row = {"conversion": f"{i}->{o}", "size": f"{w}x{h}", "ms_per_frame_1core": round(ms, 3), "Mpix_s_1core": round(w * h / ms / 1e3, 1)}
        # (ii) the reference's own row-band threading
        r = po.ref()
        r.parallel_pix_conv.restype = None
        r.parallel_pix_conv.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        ci, co = po.REF_CODEC[i], po.REF_CODEC[o]
        fn = r.get_decoder_from_to(ci, co)
        sls, dls = r.vc_get_linesize(w, ci), r.vc_get_linesize(w, co)
        pad = np.concatenate([src, np.zeros(64, np.uint8)])
        dst = np.zeros(dls * h + 64, np.uint8)
        best = (None, 1e9)
        t = 1
        cands = []
        while t < ncpu:
            cands.append(t)
            t *= 2
        cands.append(ncpu)
        for t in cands:
            r.parallel_pix_conv(h, dst.ctypes.data, dls, pad.ctypes.data, sls, fn, t)   # warm / spawn
            n, t0 = 0, time.perf_counter()
            while n < 5 or time.perf_counter() - t0 < 0.4:
                r.parallel_pix_conv(h, dst.ctypes.data, dls, pad.ctypes.data, sls, fn, t)
                n += 1
            mt = (time.perf_counter() - t0) / n * 1e3
            if t == 1:
                ms = mt   # the one-core figure: the same C line loop on one thread (the Python per-line loop above costs more than the conversion)
                row["ms_per_frame_1core"], row["Mpix_s_1core"] = round(ms, 3), round(w * h / ms / 1e3, 1)
            if mt < best[1]:
                best = (t, mt)
        row.update({"ms_per_frame_best_threads": round(best[1], 3), "threads_best": best[0], "Mpix_s_best_threads": round(w * h / best[1] / 1e3, 1)})
        rows.append(row)
        print(f"{i:>5s} -> {o:<5s} {ms:8.3f} ms/frame on 1 core ({w * h / ms / 1e3:8.1f} Mpx/s); {best[1]:7.3f} ms on {best[0]} threads of {ncpu} CPUs "
              f"({w * h / best[1] / 1e3:9.1f} Mpx/s, parallel_pix_conv row bands)", flush=True)
    src = synth.s1_random("UYVY", w, h, salt=2)
    n, t0 = 0, time.perf_counter()
    while n < 3 or time.perf_counter() - t0 < 1.0:
        po.uyvy_to_i420(src, w, h, use_ref=True)
        n += 1
    ms = (time.perf_counter() - t0) / n * 1e3
    rows.append({"conversion": "uyvy_to_i420", "size": f"{w}x{h}", "ms_per_frame_1core": round(ms, 3), "Mpix_s_1core": round(w * h / ms / 1e3, 1)})
    print(f"uyvy_to_i420   {ms:8.3f} ms/frame/core", flush=True)
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        json.dump({"host_cpus_visible": ncpu, "host_cpus": os.cpu_count(), "threading": "reference parallel_pix_conv (even row bands), best of 1,2,4,..,all CPUs", "rows": rows},
                  open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
