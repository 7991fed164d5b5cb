#!/usr/bin/env python3
"""Random search through the `-c dxt` module inside the reference's own compress framework (oracle/_ref/ug_harness): every input codec the
reference module takes, random frame sizes (multiples of 4; v210 widths that are not multiples of 12 included), DXT1 and DXT5, compared with
the reference's own sequence -- the COMPILED reference's get_best_decoder_from + line decoder, then the DXT oracle.  GPU box.
usage: python tools/find_module_mismatch.py [n]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle import pyoracle as po
from test_module_harness import HARNESS, _random_frame, _ref_best_and_decode

CODECS = ["UYVY", "YUYV", "v210", "RGB", "RGBA", "BGR", "R10k", "R12L", "RG48", "Y216", "Y416", "VUYA", "DVS10"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    bad = 0
    tmp = tempfile.mkdtemp()
    for seed in range(n):
        rng = np.random.default_rng(seed)
        codec = CODECS[seed % len(CODECS)]
        w, h = 4 * int(rng.integers(1, 120)), 4 * int(rng.integers(1, 20))
        if seed % 2:   # round 6: any frame size (dxt_glsl.cpp:150-160 hands any tile size on); 4:2:2 codecs keep an even width
            w, h = max(2, w - int(rng.integers(0, 4))), max(1, h - int(rng.integers(0, 4)))
            if codec in ("UYVY", "YUYV", "Y216", "DVS10", "v210", "R10k", "R12L"):
                w += w & 1
        if codec in ("UYVY", "YUYV", "Y216", "DVS10", "v210") and w % 2:
            w += 4
        cfg = ["dxt:DXT5", "dxt:DXT1"][int(rng.integers(2))]
        src = _random_frame(po, codec, w, h, salt=seed)
        target, conv = _ref_best_and_decode(po, codec, ["RGB", "UYVY"], src, w, h)
        raw, out = os.path.join(tmp, "in.raw"), os.path.join(tmp, "out.bin")
        src.tofile(raw)
        r = subprocess.run([HARNESS, cfg, codec, str(w), str(h), raw, out], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            print("FAILED", seed, codec, w, h, cfg, (r.stdout + r.stderr)[-300:], flush=True)
            bad += 1
            continue
        oid = po.OUT_DXT5YCOCG if cfg.endswith("DXT5") else po.OUT_DXT1
        want = po.dxt_encode(po.IN_RGB if target == "RGB" else po.IN_UYVY, oid, conv, w, h)
        if not np.array_equal(np.fromfile(out, np.uint8), want):
            print("MISMATCH", seed, codec, w, h, cfg, target, flush=True)
            bad += 1
        if bad >= 5:
            break
    print("frames", seed + 1, "problems", bad)


if __name__ == "__main__":
    main()
