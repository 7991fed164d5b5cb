"""Run the complete JPEG encoder a few times (for rocprofv3 --kernel-trace): python tools/jpeg_profile.py [w h ri]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ultragrid_amd import codec, synth

w, h, ri = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160, 4)
src = torch.from_numpy(synth.s2_video("UYVY", w, h)).cuda()
enc = codec.JpegEncoder(w, h, 75, ri)
for _ in range(30):
    data = enc.encode(src)
print(len(data))
