cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_jpeg_rtp_compat.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | grep -v "JPEG\]\|APP14" | tail -3
for i in 1 2 3; do
  for lib in libug_mi355x_prev.so libug_mi355x.so; do
    echo -n "$lib  "; UG_MI355X_LIB=$ROOT/ultragrid_amd/$lib timeout 120 python tools/bench_jpeg_batch.py --only batch 2>&1 | grep "frames per call" | tail -1
  done
done
for lib in libug_mi355x_prev.so libug_mi355x.so; do
    echo -n "$lib 422 "; UG_MI355X_LIB=$ROOT/ultragrid_amd/$lib timeout 120 python tools/bench_jpeg_batch.py --sub 422 --only batch 2>&1 | grep "frames per call" | tail -1
    echo -n "$lib 444 "; UG_MI355X_LIB=$ROOT/ultragrid_amd/$lib timeout 120 python tools/bench_jpeg_batch.py --sub 444 --only batch 2>&1 | grep "frames per call" | tail -1
    echo -n "$lib one "; UG_MI355X_LIB=$ROOT/ultragrid_amd/$lib timeout 120 python tools/bench_jpeg_batch.py --only single 2>&1 | grep "per call" | tail -1
done
