cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04z; mkdir -p $OUT
timeout 600 python -m pytest tests/test_deinterlace.py tests/test_module_harness.py -q -k "deinterlace or interlaced" 2>&1 | grep -E "passed|failed" | tail -2
timeout 900 python tools/find_deinterlace_mismatch.py 3000 2>&1 | grep -v amdgpu.ids | tail -12 > $OUT/find_deinterlace.txt; cat $OUT/find_deinterlace.txt
