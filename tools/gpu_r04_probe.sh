cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_jpeg_rtp_compat.py -q -x 2>&1 | grep -v lavc_vid_conv | tail -3
timeout 600 python -m pytest tests/test_module_harness.py tests/test_reference_unit_tests.py -k "jpeg or gpujpeg" -q -x 2>&1 | grep -v lavc_vid_conv | tail -2
timeout 500 python tools/find_encode_mismatch.py 1200 2>&1 | tail -2
timeout 100 python tools/bench_jpeg_batch.py 2>&1 | grep "per call"
timeout 100 python tools/bench_jpeg_batch.py --sub 422 --only batch 2>&1 | grep "per call"
timeout 100 python tools/bench_jpeg_batch.py --sub 444 --only batch 2>&1 | grep "per call"
timeout 100 python tools/bench_jpeg_batch.py --n 16 --only batch 2>&1 | grep "per call"
