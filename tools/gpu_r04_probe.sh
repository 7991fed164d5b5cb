cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_deinterlace.py tests/test_module_harness.py -q -k "deinterlace or interlaced" 2>&1 | grep -E "passed|failed|assert|Error" | tail -5
python - <<'PY'
import torch, time
from ultragrid_amd import lib as L
l=L.load()
for (w,h,bpp,name) in ((1920,1080,2,"1080i UYVY"),(1920,1080,3,"1080i RGB"),(3840,2160,2,"4K UYVY")):
    ls=w*bpp
    n=8
    buf=torch.randint(0,256,(n,ls*h),dtype=torch.uint8,device="cuda")
    st=torch.cuda.current_stream().cuda_stream
    for _ in range(3): l.ug_hip_deinterlace_blend_batch(buf.data_ptr(), ls, h, n, ls*h, st)
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): l.ug_hip_deinterlace_blend_batch(buf.data_ptr(), ls, h, n, ls*h, st)
    e1.record(); torch.cuda.synchronize()
    tb=e0.elapsed_time(e1)/20/n*1e3
    e0.record()
    for _ in range(20): l.ug_hip_deinterlace_blend(buf.data_ptr(), ls, h, st)
    e1.record(); torch.cuda.synchronize()
    t1=e0.elapsed_time(e1)/20*1e3
    print(f"deinterlace {name}: {t1:.1f} us one frame, {tb:.1f} us per frame at 8 per launch ({2*ls*h/tb/1e3:.0f} GB/s)")
PY
