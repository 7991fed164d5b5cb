cd ${GRAFT_REPO_ROOT:-.}
time (timeout 900 python tools/find_libjpeg_mismatch.py 200 selftest 2>&1 | grep -v "amdgpu.ids\|JPEG\]\|APP14" | tail -4)
