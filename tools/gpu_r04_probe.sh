cd ${GRAFT_REPO_ROOT:-.}
for sub in 420 422; do
for q in 50 75 90 95 98 100; do
  for ri in 2 4 8 16; do
    timeout 60 python tools/bench_jpeg_batch.py --sub $sub --q $q --ri $ri --only batch --seconds 0.3 2>&1 | grep "per call" | tail -1
  done
done
done
