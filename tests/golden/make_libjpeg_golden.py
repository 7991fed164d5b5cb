#!/usr/bin/env python3
"""Freeze the libjpeg-turbo pin of the JPEG path into tests/golden/libjpeg_float.npz.

The JPEG stage's pin towards an executable, published implementation is the distribution's libjpeg-turbo run with its FLOAT DCT
(tests/libjpeg_float.py: IJG's jfdctflt.c / its SSE form + the float quantiser + standard Huffman coding).  That library is addressed through
hard-coded struct offsets and may be absent from another image; this script runs it ONCE, where it exists, and commits what it produced:

  fdct_blocks / fdct_out      512 8x8 blocks of samples (noise, two-valued, gaussian, ramps, video) and the BITS of jpeg_fdct_float's output for the
                              level-shifted samples of each
  quality_scaling             jpeg_quality_scaling(1..100)
  meta (JSON) + in_<k> / scan_<j>   compression cases: the input picture (grey plane, packed RGB, or UYVY), size, sampling, quality, restart
                              interval, and libjpeg-turbo's entropy-coded bytes (between the SOS header and EOI) for it --
                                grey   a plane compressed as one component (the oracle's FDCT + quantiser, tests/test_oracle_jpeg.py)
                                rgb    packed RGB kept as R,G,B components 4:4:4 (what GPUJPEG is asked for with RGB input, gpujpeg.cpp:303-305)
                                422 / 420   UYVY; libjpeg gets the planes the reference's converters make of it (uyvy_to_i422 / uyvy_to_i420 as the
                                            oracle has them, pinned to the compiled reference) through jpeg_write_raw_data
                              incl. edge blocks (sizes that are no multiple of the MCU), q = 100 on noise, restart intervals 0..64
  library                     what produced it (version string of the shared object, file name)

tests/test_oracle_jpeg.py (CPU) and tests/test_gpu_jpeg.py (GPU) compare the oracle and the product with this fixture ALWAYS; the live comparison
with the library stays as an extra where the library exists.  libjpeg-turbo is not the library UltraGrid links (libgpujpeg, unobtainable here):
towards the reference the stage stays "parity unpinned".

    python tests/golden/make_libjpeg_golden.py        (needs libjpeg.so.8 = libjpeg-turbo 2.1.x, IJG API 80)
"""
import ctypes as C
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import libjpeg_float as ljf  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from ultragrid_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "libjpeg_float.npz")


def library_version(lj):
    path = None
    for line in open("/proc/self/maps"):
        if "libjpeg" in line:
            path = line.split()[-1]
            break
    ver = "unknown"
    if path:
        m = re.search(rb"libjpeg-turbo version ([0-9.]+)", open(path, "rb").read())
        if m:
            ver = m.group(1).decode()
    return f"libjpeg-turbo {ver} ({os.path.basename(os.path.realpath(path)) if path else '?'}), JPEG_LIB_VERSION 80, dct_method JDCT_FLOAT"


def uyvy_planes(uyvy, w, h, sub):
    if sub == 420:
        return po.uyvy_to_i420(uyvy, w, h)
    a = uyvy.reshape(h, 2 * w)
    return a[:, 1::2], a[:, 0::4], a[:, 2::4]


def main():
    lj = ljf.load()
    assert lj is not None, "needs libjpeg.so.8 (libjpeg-turbo, IJG API 80)"
    po.build()
    g = {}
    # ---- forward DCT ----
    rng = np.random.default_rng(2026)
    kinds = [rng.integers(0, 256, (128, 8, 8)), rng.integers(0, 2, (128, 8, 8)) * 255, np.clip(128 + 30 * rng.standard_normal((128, 8, 8)), 0, 255),
             (np.arange(64).reshape(8, 8)[None] * rng.integers(1, 5, (64, 1, 1)) + rng.integers(0, 256, (64, 1, 1))) % 256]
    video = synth.s2_video("UYVY", 64, 64).reshape(64, 128)
    kinds.append(np.stack([video[8 * (i // 8):8 * (i // 8) + 8, 8 * (i % 8):8 * (i % 8) + 8] for i in range(64)]))
    blocks = np.concatenate(kinds).astype(np.uint8)
    assert blocks.shape == (512, 8, 8)
    out = np.empty((512, 64), np.uint32)
    for b in range(512):
        blk = np.ascontiguousarray(blocks[b].astype(np.float32) - np.float32(128.0))
        lj.jpeg_fdct_float(blk.ctypes.data)
        out[b] = blk.ravel().view(np.uint32)
    g["fdct_blocks"], g["fdct_out"] = blocks, out
    g["quality_scaling"] = np.array([lj.jpeg_quality_scaling(q) for q in range(1, 101)], np.int32)
    # ---- compression cases ----
    inputs, cases = {}, []

    def add_input(name, arr):
        inputs.setdefault(name, np.ascontiguousarray(arr, np.uint8))
        return name

    def add(kind, inp, w, h, q, ri, scan):
        g[f"scan_{len(cases)}"] = np.frombuffer(scan, np.uint8)
        cases.append(dict(kind=kind, input=inp, w=w, h=h, q=q, ri=ri))

    rng = np.random.default_rng(7)
    for i, (h, w) in enumerate(((1, 1), (7, 9), (8, 8), (33, 20), (64, 150), (89, 149))):          # grey planes: the oracle's stage
        contents = {"noise": rng.integers(0, 256, (h, w)), "gauss": np.clip(128 + 40 * rng.standard_normal((h, w)), 0, 255),
                    "ramp": (np.add.outer(np.arange(h), np.arange(w)) * (3 + i)) % 256, "flat": np.full((h, w), 17 * i), "two": rng.integers(0, 2, (h, w)) * 255}
        for j, (cname, plane) in enumerate(contents.items()):
            if (i + j) % 2:
                continue
            name = add_input(f"grey_{h}x{w}_{cname}", plane)
            for q in ((10, 75, 100) if cname in ("noise", "two") else (50, 92)):
                add("grey", name, w, h, q, 0, ljf.scan_bytes(ljf.compress(lj, inputs[name], q)))
    for (w, h) in ((64, 48), (203, 33), (8, 8), (1100, 50), (640, 40)):                              # packed RGB, 4:4:4
        s2 = add_input(f"rgb_{w}x{h}_s2", synth.frame("S2", "RGB", w, h).reshape(h, w, 3))
        for q, ri in ((75, 4), (50, 8), (92, 1), (20, 64)):
            add("rgb", s2, w, h, q, ri, ljf.scan_bytes(ljf.compress(lj, inputs[s2], q, restart=ri)))
        if w * h < 20000:
            nz = add_input(f"rgb_{w}x{h}_noise", rng.integers(0, 256, (h, w, 3)))
            add("rgb", nz, w, h, 100, 16, ljf.scan_bytes(ljf.compress(lj, inputs[nz], 100, restart=16)))
    for sub, sizes in ((422, ((64, 48), (1040, 81), (28, 5), (640, 24))), (420, ((64, 48), (1040, 80), (1036, 90), (48, 16), (640, 32)))):
        for (w, h) in sizes:
            s2 = add_input(f"uyvy_{w}x{h}_s2", synth.s2_video("UYVY", w, h))
            for q, ri in ((75, 4), (92, 1), (50, 8)):
                y, u, v = uyvy_planes(inputs[s2].ravel(), w, h, sub)
                add(str(sub), s2, w, h, q, ri, ljf.scan_bytes(ljf.compress_planes(lj, y, u, v, w, h, sub, q, restart=ri)))
            if w * h < 20000:
                nz = add_input(f"uyvy_{w}x{h}_noise", synth.s1_random("UYVY", w, h, salt=100))
                y, u, v = uyvy_planes(inputs[nz].ravel(), w, h, sub)
                add(str(sub), nz, w, h, 100, 2, ljf.scan_bytes(ljf.compress_planes(lj, y, u, v, w, h, sub, 100, restart=2)))
    for name, arr in inputs.items():
        g["in_" + name] = arr
    g["meta"] = np.array(json.dumps(dict(library=library_version(lj), cases=cases)))
    np.savez_compressed(OUT, **g)
    print(f"{OUT}: {len(cases)} compression cases, {len(inputs)} inputs, {os.path.getsize(OUT)} bytes; {library_version(lj)}")


if __name__ == "__main__":
    main()
