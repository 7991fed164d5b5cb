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
    a = ap.parse_args()
    node = pipeline.gpu_numa_node(torch.cuda.current_device())
    bound = pipeline.bind_to_numa_node(node)
    for wl in (sorted(pipeline.WORKLOADS) if a.workload == "all" else [a.workload]):
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
