#!/usr/bin/env python3
"""Random search for a picture / parameter set on which the GPU JPEG encoder's byte stream differs from the test writer fed with the
oracle's coefficients (tests/jpeg_bitstream.py): sizes from one MCU up (odd heights included), qualities 1..100, restart intervals 1..40,
4:2:0 / 4:2:2 / R,G,B 4:4:4, smooth to pure-noise content.  Also decodes every stream with the GPU decoder and compares with the oracle.
GPU box.  usage: python tools/find_encode_mismatch.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import pyoracle as po
from ultragrid_amd import codec as hip, lib as L
from jpeg_bitstream import write_jpeg, write_jpeg420

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dec = hip.JpegDecoder()
bad = 0
for seed in range(n):
    rng = np.random.default_rng(seed)
    sub = [420, 422, 444][int(rng.integers(3))]
    w, h = 2 * int(rng.integers(1, 130)), int(rng.integers(1, 150))
    q, ri = int(rng.integers(1, 101)), int(rng.integers(1, 41))
    if seed % 2:  # round 4: half of the cases where the fused kernels run (width % 16 == 0 for UYVY, restart interval a power of two up to 32 / 64) ...
        w = 16 * int(rng.integers(1, 70))
        ri = int(rng.choice([r for r in (1, 2, 4, 8, 16, 32, 64) if sub == 444 or r <= 32]))
    if seed % 11 == 0:     # round 6: no restart intervals at all (restart_interval 0: one segment, coded in parallel by the nori_* kernels)
        ri = 0
    two = seed % 4 >= 2   # ... and half of all cases as a batch of two frames (the two-launch placement; one-frame calls place in one launch)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / (3 + 40 * rng.random())) * np.cos(yy / (3 + 30 * rng.random())), 128 + 90 * np.cos(xx / 33.0 + yy / (5 + 20 * rng.random())),
                     128 + 80 * np.sin(yy / (2 + 9 * rng.random()))], -1)
    rgb = (base + rng.normal(0, [0.0, 2.0, 10.0, 60.0, 200.0][int(rng.integers(5))], base.shape)).clip(0, 255).astype(np.uint8)
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    dl, dc = po.jpeg_divisors(ql), po.jpeg_divisors(qc)
    # round 6: the encoder's other options on a third of the draws -- color_space_internal (1 = RGB as it comes .. 4 = BT.709; 4:2:x: BT.601 / 256 levels) and, 4:4:4 only, one scan per component
    cs = int(rng.integers(2, 5)) if seed % 3 == 0 else 0
    nonint = sub == 444 and seed % 6 < 3
    if sub != 444 and cs == 4:
        cs = 0
    from422 = sub == 444 and seed % 5 == 4      # a 4:4:4 encoder fed UYVY (UG_JPEG_INPUT_UYVY), any internal colour space, either scan layout
    if from422:
        cs = int(rng.integers(0, 5))
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=sub, internal_cs=cs, flags=(L.JPEG_NONINTERLEAVED if nonint else 0) | (L.JPEG_INPUT_UYVY if from422 else 0))
    if from422:
        uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
        dev = torch.from_numpy(uyvy).cuda()
        data = enc.encode_batch(torch.stack([dev, dev]), L.PF_UYVY)[1] if two else enc.encode(dev, L.PF_UYVY)
        comps = po.jpeg_colour_convert("UYVY444", 4, cs or 4, uyvy, w, h).reshape(h, w, 3)
        ycc = cs != 1
        coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(comps[..., c]), dc if ycc and c else dl, (w + 7) // 8, (h + 7) // 8) for c in range(3)]
        if nonint:
            from jpeg_bitstream import write_jpeg_noninterleaved
            want = write_jpeg_noninterleaved(w, h, ql, coefs, restart=ri, qt_chroma=qc if ycc else None)
        else:
            want = write_jpeg(w, h, ql, qc, *coefs, restart=ri, sub=444, ycc=ycc)
    elif sub == 444:
        dev = torch.from_numpy(np.ascontiguousarray(rgb).ravel()).cuda()
        data = enc.encode_batch(torch.stack([dev, dev]), L.PF_RGB)[1] if two else enc.encode(dev, L.PF_RGB)
        comps = po.jpeg_colour_convert("RGB", 1, cs, rgb, w, h).reshape(h, w, 3) if cs else rgb
        coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(comps[..., c]), dc if cs and c else dl, (w + 7) // 8, (h + 7) // 8) for c in range(3)]
        if nonint:
            from jpeg_bitstream import write_jpeg_noninterleaved
            want = write_jpeg_noninterleaved(w, h, ql, coefs, restart=ri, qt_chroma=qc if cs else None)
        else:
            want = write_jpeg(w, h, ql, qc, *coefs, restart=ri, sub=444, ycc=bool(cs))
    else:
        uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
        if cs:
            src709, uyvy = uyvy, po.jpeg_colour_convert("UYVY", 4, cs, uyvy, w, h)       # the samples the stream must hold; the encoder is fed the BT.709 ones
        dev = torch.from_numpy(src709 if cs else uyvy).cuda()
        data = enc.encode_batch(torch.stack([dev, dev]))[1] if two else enc.encode(dev)
        if sub == 422:
            y, u, v = po.uyvy_to_i422(uyvy, w, h)
            mw, mh = (w + 15) // 16, (h + 7) // 8
            want = write_jpeg(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, dl, 2 * mw, mh), po.jpeg_fdct_quant_plane(u, dc, mw, mh), po.jpeg_fdct_quant_plane(v, dc, mw, mh), restart=ri, sub=422)
        else:
            y, u, v = po.uyvy_to_i420(uyvy, w, h)
            mw, mh = (w + 15) // 16, (h + 15) // 16
            want = write_jpeg420(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, dl, 2 * mw, 2 * mh), po.jpeg_fdct_quant_plane(u, dc, mw, mh), po.jpeg_fdct_quant_plane(v, dc, mw, mh), restart=ri)
    enc.close()
    if data != want:
        print("ENCODE MISMATCH seed", seed, sub, w, h, q, ri, "cs", cs, "nonint", nonint, len(data), len(want), flush=True)
        bad += 1
    _, crop, _ = po.jpeg_decode_planes(data)
    for c, pl in enumerate(dec.planes(data)):
        if not np.array_equal(pl.cpu().numpy(), crop[c]):
            print("DECODE MISMATCH seed", seed, sub, w, h, q, ri, "comp", c, flush=True)
            bad += 1
            break
    if bad >= 4:
        break
print("pictures", seed + 1, "mismatches", bad)
