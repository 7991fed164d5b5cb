cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04z; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; tail -2 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
