cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04z; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; grep -E "passed|failed" $OUT/pytest.log | tail -1
timeout 300 python tools/bench_decode.py --json $OUT/decode.json > $OUT/decode.txt 2>&1; grep -v amdgpu $OUT/decode.txt
timeout 900 python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; grep "dxt_decode\|jpeg encoder" $OUT/kernels_table.txt | cut -c1-150
timeout 600 python tools/find_dxt_mismatch.py 1500 2>&1 | tail -1 > $OUT/find_dxt.txt; cat $OUT/find_dxt.txt
