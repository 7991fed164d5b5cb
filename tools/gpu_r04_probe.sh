cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_gpu_dxt_decode.py tests/test_module_harness.py -q -x -k "decode or dxt" 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do for lib in libug_mi355x_prev.so libug_mi355x.so; do
  echo "== $lib"; UG_MI355X_LIB=$ROOT/ultragrid_amd/$lib timeout 200 python tools/bench_decode.py 2>&1 | grep "UYVY"
done; done
