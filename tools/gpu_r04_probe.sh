cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_jpeg.py -q -x -k "libjpeg_turbo" 2>&1 | grep -v "JPEG\]\|APP14\|lavc_vid" | tail -15
