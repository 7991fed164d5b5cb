#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the hot path as the UltraGrid module drives it: pinned host frame -> H2D -> fused encode kernel
-> D2H of the compressed frame, `--depth` frames in flight on as many streams (ultragrid_amd/pipeline.py, the same code bench.py's `e2e`
leg runs).  This is NOT bench.py's `value` (that one is HBM-resident).  One JSON line per workload.
usage: python tools/e2e_bench.py [--workload 8k-v210|4k-uyvy|1080p-rgb-dxt1|all] [--depth 3] [--seconds 4]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ultragrid_amd import pipeline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="all", choices=sorted(pipeline.WORKLOADS) + ["all"])
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--sweep", action="store_true", help="link probe, copy-only ceiling, depth 2-6 and both stream layouts for every workload")
    a = ap.parse_args()
    node = pipeline.gpu_numa_node(torch.cuda.current_device())
    bound = pipeline.bind_to_numa_node(node)
    wls = sorted(pipeline.WORKLOADS) if a.workload == "all" else [a.workload]
    if a.sweep:
        for streams in (1, 2, 4):
            r = pipeline.link_probe(seconds=min(a.seconds, 1.5), streams=streams)
            r.update({"probe": "link", "gpu_numa_node": node, "cpus_bound": bound})
            print(json.dumps(r), flush=True)
        for wl in wls:
            for mode in ("per-frame-stream", "split"):
                for depth in (2, 3, 4, 6):
                    ceil = pipeline.run(wl, depth=depth, seconds=min(a.seconds, 1.5), mode=mode, encode=False)
                    r = pipeline.run(wl, depth=depth, seconds=a.seconds, mode=mode)
                    r.update({"copy_only_fps": ceil["fps"], "copy_only_pcie_gbs": ceil["pcie_gbs"], "frac_of_copy_only": round(r["fps"] / ceil["fps"], 3)})
                    print(json.dumps(r), flush=True)
        return
    for wl in wls:
        r = pipeline.run(wl, depth=a.depth, seconds=a.seconds)
        r.update({"gpu_numa_node": node, "cpus_bound": bound, "device": torch.cuda.get_device_name(0)})
        print(json.dumps(r), flush=True)
    if a.workload == "all":   # and the receiving direction: JPEG in, raw frame out
        for out in ("UYVY", "RGBA"):
            r = pipeline.run_jpeg_decode(depth=a.depth, seconds=a.seconds, out=out)
            r.update({"gpu_numa_node": node, "cpus_bound": bound})
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
