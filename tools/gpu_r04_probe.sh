cd ${GRAFT_REPO_ROOT:-.}
python tools/sync_probe.py 2>&1 | grep -v amdgpu
