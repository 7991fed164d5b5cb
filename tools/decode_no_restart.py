#!/usr/bin/env python3
"""How the JPEG decoder fares on streams WITHOUT restart intervals (another sender's; UltraGrid's own always carry them, gpujpeg.cpp:345-352): the whole scan is one
segment = one lane.  Pillow-encoded 4:2:0 pictures, decode to UYVY, ms per frame; beside it the same picture with restart intervals.  GPU box."""
import io
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ultragrid_amd import codec as hip, lib as L

for (w, h) in [(640, 360), (1280, 720), (1920, 1080), (3840, 2160)]:
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
    rgb = (rgb + np.random.default_rng(1).normal(0, 6, rgb.shape)).clip(0, 255).astype(np.uint8)
    for rst in (0, 1):
        b = io.BytesIO()
        Image.fromarray(rgb).save(b, "JPEG", quality=75, subsampling=2, **({"restart_marker_rows": 1} if rst else {}))
        data = b.getvalue()
        dec = hip.JpegDecoder()
        dec.decode(data, L.PF_UYVY)
        torch.cuda.synchronize()
        n = 5 if not rst else 50
        t0 = time.perf_counter()
        for _ in range(n):
            dec.decode(data, L.PF_UYVY)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        dec.close()
        print(f"{w}x{h} 4:2:0 q75 {len(data):8d} B  {'one restart interval per MCU row' if rst else 'no restart intervals            '}: {dt * 1e3:9.3f} ms per frame")

# few, long segments: FFmpeg's slice-threaded MJPEG writes one restart interval per slice (8 here), libjpeg users often one per MCU row
for (w, h) in [(1920, 1080), (3840, 2160)]:
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
    rgb = (rgb + np.random.default_rng(1).normal(0, 6, rgb.shape)).clip(0, 255).astype(np.uint8)
    rows = (h + 15) // 16
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", quality=75, subsampling=2, restart_marker_rows=(rows + 7) // 8)
    data = b.getvalue()
    dec = hip.JpegDecoder()
    dec.decode(data, L.PF_UYVY)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dec.decode(data, L.PF_UYVY)
    torch.cuda.synchronize()
    dec.close()
    print(f"{w}x{h} 4:2:0 q75 {len(data):8d} B  eight restart intervals per frame  : {(time.perf_counter() - t0) / 20 * 1e3:9.3f} ms per frame")

# the repository's own 4K 4:2:2 streams with longer restart intervals than the default: where the choice between the two ways falls (run with UG_JPEG_DEC_SYNC=0 for the other side)
from ultragrid_amd import synth
w, h = 3840, 2160
src = torch.from_numpy(synth.s2_video("UYVY", w, h, salt=1)).cuda()
for q, ri in [(75, 4), (75, 32), (95, 16), (95, 32), (95, 128), (75, 256), (75, 2000)]:
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=422)
    data = enc.encode(src, L.PF_UYVY)
    enc.close()
    n_seg = -(-(w // 16 * (h // 8)) // ri)
    dec = hip.JpegDecoder()
    dec.decode(data, L.PF_UYVY)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dec.decode(data, L.PF_UYVY)
    torch.cuda.synchronize()
    dec.close()
    print(f"own 4K 4:2:2 q{q} restart {ri:4d}: {len(data):8d} B, {len(data) // n_seg:7d} B per segment: {(time.perf_counter() - t0) / 20 * 1e3:9.3f} ms per frame")
