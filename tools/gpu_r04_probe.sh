cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_jpeg_rtp_compat.py -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 600 python -m pytest tests/test_module_harness.py tests/test_reference_unit_tests.py -k "jpeg or gpujpeg" -q 2>&1 | grep -E "passed|failed" | tail -2
