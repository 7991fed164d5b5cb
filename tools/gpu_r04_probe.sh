cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_pixfmt.py -q -x -k "jpeg or plane or fused or batched_front or i420 or 444" 2>&1 | grep -v lavc_vid_conv | tail -3
for r in 1 2 3; do for lib in ultragrid_amd/libug_mi355x.so ultragrid_amd/libug_mi355x_storefull.so; do
  UG_MI355X_LIB=$(realpath $lib) python bench.py --workload 4k-uyvy-jpeg420 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['ms_per_launch'], d['roofline']['frac'])"
done; done
for lib in ultragrid_amd/libug_mi355x.so ultragrid_amd/libug_mi355x_storefull.so; do echo $lib; UG_MI355X_LIB=$(realpath $lib) python tools/bench_kernels.py 2>/dev/null | grep -i "fdct\|jpeg encoder" ; done
