import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, numpy as np
from ultragrid_amd import codec as hip, lib as L, synth
for (w, h) in [(1920, 1080), (3840, 2160)]:
    src = torch.from_numpy(synth.s2_video("UYVY", w, h, salt=1)).cuda()
    for ri in (0, 4):
        enc = hip.JpegEncoder(w, h, 75, ri, subsampling=422)
        enc.encode(src, L.PF_UYVY)
        torch.cuda.synchronize()
        n = 5 if ri == 0 else 100
        t0 = time.perf_counter()
        for _ in range(n):
            d = enc.encode(src, L.PF_UYVY)
        dt = (time.perf_counter() - t0) / n
        enc.close()
        print(f"{w}x{h} 4:2:2 q75 restart={ri}: {dt*1e3:8.3f} ms per frame (incl. the stream's copy to the host), {len(d)} B")

# restart intervals too long for the block coder (more than 256 blocks per segment): UG_JPEG_NORI=0 for the wave-per-segment coder they took before
w, h = 3840, 2160
src = torch.from_numpy(synth.s2_video("UYVY", w, h, salt=1)).cuda()
for ri in (65, 200, 1000, 2000, 20000):
    enc = hip.JpegEncoder(w, h, 75, ri, subsampling=422)
    enc.encode(src, L.PF_UYVY)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        d = enc.encode(src, L.PF_UYVY)
    dt = (time.perf_counter() - t0) / 20
    enc.close()
    print(f"{w}x{h} 4:2:2 q75 restart={ri:5d} ({4 * ri:6d} blocks per segment): {dt*1e3:8.3f} ms per frame")
