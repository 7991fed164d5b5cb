#!/usr/bin/env python3
"""DXT decoders at 3840x2160: one frame per launch and 8 frames per launch (frames one picture apart = one image 8 times as tall),
rotating over enough buffers to exceed the Infinity Cache.  ms per frame, algorithmic GB/s (compressed bytes read + pixels written),
fraction of 8 TB/s.  Usage (GPU box): python tools/bench_decode.py [--json out.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ultragrid_amd import lib as L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    a = ap.parse_args()
    l = L.load()
    w, h = 3840, 2160
    rows = []
    for in_name, in_id, bpp_in in (("DXT5-YCoCg", L.DXT5_YCOCG, 1.0), ("DXT1", L.DXT1, 0.5), ("DXT1_YUV", L.DXT1_YUV, 0.5)):
        for out_name, bpp_out in (("RGBA", 4), ("RGB", 3), ("UYVY", 2)):
            for frames in (1, 8):
                hh = h * frames
                n_in, n_out = int(w * hh * bpp_in), w * hh * bpp_out
                nbuf = max(2, int(2.4e9 // (n_in + n_out)) + 1)   # >= 2.4 GB rotating (profiles/r04_rotation_sweep.txt)
                src = torch.randint(0, 256, (nbuf, n_in), dtype=torch.uint8, device="cuda")
                dst = torch.empty((nbuf, n_out), dtype=torch.uint8, device="cuda")

                def run(k):
                    rc = l.ug_hip_dxt_decode(in_id, L.PF_NAMES[out_name], src[k % nbuf].data_ptr(), dst[k % nbuf].data_ptr(), w, hh, 0, 0, 8, 16, torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, L.last_error()
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
                n = max(20, int(120.0 / one))         # timed: >= 120 ms of launches
                torch.cuda.synchronize()
                e0.record()
                for k in range(n):
                    run(k)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
                gbs = (n_in + n_out) / (ms * 1e-3) / 1e9
                rows.append({"row": f"{in_name}->{out_name} x{frames}", "ms_per_frame": round(ms / frames, 5), "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 3)})
                print(f"{in_name:>10}->{out_name:<4} x{frames}: {ms / frames * 1e3:8.2f} us/frame {gbs:8.1f} GB/s {gbs / 8000:.3f}", flush=True)
                del src, dst
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump({"width": w, "height": h, "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
