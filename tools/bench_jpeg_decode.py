#!/usr/bin/env python3
"""The JPEG decoder on a 4K frame made by the repository's own encoder: time per call (host header parse + restart-marker scan, upload, kernels,
no download) for several restart intervals / samplings.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
Usage (GPU box): python tools/bench_jpeg_decode.py [--json out.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ultragrid_amd import codec, lib as L, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--concurrent", type=int, default=4)
    ap.add_argument("--configs", type=int, default=6, help="only the first N stream configurations")
    ap.add_argument("--seconds", type=float, default=0.5)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    a = ap.parse_args()
    w, h = a.width, a.height
    src = torch.from_numpy(synth.s2_video("UYVY", w, h)).cuda()
    rgb = torch.from_numpy(synth.s1_random("RGB", w, h)).cuda()
    rows = []
    for sub, ri, out in ((422, 4, "UYVY"), (422, 1, "UYVY"), (422, 16, "UYVY"), (420, 4, "UYVY"), (422, 4, "RGBA"), (444, 4, "RGB"))[:a.configs]:
        enc = codec.JpegEncoder(w, h, 75, ri, subsampling=sub)
        data = enc.encode(rgb if sub == 444 else src, L.PF_RGB if sub == 444 else L.PF_UYVY)
        enc.close()
        dec = codec.JpegDecoder()
        l = L.load()
        dst = torch.empty(codec.linesize(L.PF_NAMES[out], w) * h, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream

        def run():
            rc = l.ug_hip_jpeg_decoder_decode(dec._h, data, len(data), L.PF_NAMES[out], dst.data_ptr(), 0, 0, 8, 16, st)
            assert rc == 0, L.last_error()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while n < 20 or time.perf_counter() - t0 < a.seconds:
            run()
            n += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        t1 = time.perf_counter()
        for _ in range(20):
            codec.jpeg_read_info(data)
        parse_ms = (time.perf_counter() - t1) / 20 * 1e3
        rows.append({"stream": f"{w}x{h} {sub} q75 restart {ri} ({len(data)} B)", "out": out, "ms_per_frame": round(ms, 4), "fps": round(1e3 / ms, 1), "header_parse_ms": round(parse_ms, 4)})
        print(f"{sub} ri={ri:<2d} -> {out:<4s}: {ms * 1e3:8.1f} us per frame ({1e3 / ms:7.1f} fps), {len(data)} B; header-only parse {parse_ms * 1e3:6.1f} us", flush=True)
        dec.close()
        if a.concurrent > 1 and (sub, ri, out) == (422, 4, "UYVY"):  # frames of a stream decoded side by side: one decoder + HIP stream each
            decs = [codec.JpegDecoder() for _ in range(a.concurrent)]
            sts = [torch.cuda.Stream() for _ in range(a.concurrent)]
            dsts = [torch.empty_like(dst) for _ in range(a.concurrent)]

            def run_k(k):
                rc = l.ug_hip_jpeg_decoder_decode(decs[k]._h, data, len(data), L.PF_NAMES[out], dsts[k].data_ptr(), 0, 0, 8, 16, sts[k].cuda_stream)
                assert rc == 0, L.last_error()
            for i in range(4 * a.concurrent):
                run_k(i % a.concurrent)
            torch.cuda.synchronize()
            n, t0 = 0, time.perf_counter()
            while n < 40 or time.perf_counter() - t0 < 0.5:
                run_k(n % a.concurrent)
                n += 1
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            rows.append({"stream": f"{w}x{h} {sub} q75 restart {ri}, {a.concurrent} frames in flight", "out": out, "ms_per_frame": round(ms, 4), "fps": round(1e3 / ms, 1)})
            print(f"    {a.concurrent} decoders on {a.concurrent} streams: {ms * 1e3:8.1f} us per frame ({1e3 / ms:7.1f} fps)", flush=True)
            for d_ in decs:
                d_.close()
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
