#!/usr/bin/env python3
"""Launch one DXT decode configuration repeatedly (for rocprofv3 --pmc / --kernel-trace): one_decode.py IN OUT FRAMES [iters]
IN = DXT5 | DXT1 | DXT1_YUV, OUT = RGBA | RGB | UYVY; 3840x2160, FRAMES pictures per launch; blocks = an encoded S2 video-noise frame
(UG_DECODE_RANDOM=1: uniformly random block bytes instead)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ultragrid_amd import codec, lib as L, synth

in_name, out_name, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
l = L.load()
w, h = 3840, 2160
in_id = {"DXT5": L.DXT5_YCOCG, "DXT1": L.DXT1, "DXT1_YUV": L.DXT1_YUV}[in_name]
bpp_in = 1.0 if in_name == "DXT5" else 0.5
bpp_out = {"RGBA": 4, "RGB": 3, "UYVY": 2}[out_name]
hh = h * frames
n_in, n_out = int(w * hh * bpp_in), w * hh * bpp_out
nbuf = max(2, int(600e6 // (n_in + n_out)) + 1)
if os.environ.get("UG_DECODE_RANDOM"):
    src = torch.randint(0, 256, (nbuf, n_in), dtype=torch.uint8, device="cuda")
else:
    uyvy = torch.from_numpy(synth.s2_video("UYVY", w, h)).cuda()
    one = codec.dxt_encode(L.PF_UYVY, L.DXT5_YCOCG if in_name == "DXT5" else L.DXT1, uyvy, w, h)
    src = one.repeat(frames).unsqueeze(0).repeat(nbuf, 1).contiguous()
dst = torch.empty((nbuf, n_out), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def run(k):
    rc = l.ug_hip_dxt_decode(in_id, L.PF_NAMES[out_name], src[k % nbuf].data_ptr(), dst[k % nbuf].data_ptr(), w, hh, 0, 0, 8, 16, st)
    assert rc == 0, L.last_error()


for k in range(5):
    run(k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(iters):
    run(k)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"{in_name}->{out_name} x{frames}: {ms / frames * 1e3:.2f} us/frame, {(n_in + n_out) / (ms * 1e-3) / 1e9:.1f} GB/s, frac {(n_in + n_out) / (ms * 1e-3) / 8e12:.3f}")
