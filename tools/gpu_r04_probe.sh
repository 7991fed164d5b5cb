cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_jpeg_rtp_compat.py -q 2>&1 | grep -E "passed|failed|^FAILED|assert" | tail -6
timeout 600 python -m pytest tests/test_module_harness.py -k "jpeg" -q 2>&1 | grep -E "passed|failed" | tail -2
python - <<'PY'
import time, torch, numpy as np
from ultragrid_amd import codec as hip, lib as L, synth
from oracle import pyoracle as po
w,h,n=3840,2160,8
uyvy=synth.s2_video("UYVY",w,h)
i420=np.concatenate([p.ravel() for p in po.uyvy_to_i420(uyvy,w,h)])
dev=torch.from_numpy(i420).cuda()
batch=torch.stack([torch.roll(dev, 3840*37*f) for f in range(n)])
for env in ("fused","unfused"):
    import os
    if env=="unfused": os.environ["UG_JPEG_FUSED"]="0"
    e=hip.JpegEncoder(w,h,75,4,subsampling=420)
    for _ in range(5): e.encode_batch(batch, L.PF_I420)
    torch.cuda.synchronize(); t=time.perf_counter(); k=0
    while time.perf_counter()-t<1.0: e.encode_batch(batch, L.PF_I420); k+=1
    dt=time.perf_counter()-t
    print(f"I420 4K 4:2:0 q75 ri4 {env}: {dt/(k*n)*1e6:.1f} us per frame (incl. the python copy of the streams)")
    e.close()
PY
