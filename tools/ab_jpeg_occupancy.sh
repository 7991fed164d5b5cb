#!/bin/bash
# A/B of the register allocator's occupancy target for jpeg_code_kernel (libug_ab_e<tag>.so): every call form of tools/bench_jpeg_batch.py
for r in 1 2; do
for t in 5b 3a 2a 1a; do
  for args in "--sub 420" "--sub 422" "--sub 444" "--sub 422 --size 1920x1080 --n 16"; do
    UG_MI355X_LIB=$(realpath ultragrid_amd/libug_ab_e$t.so) timeout 100 python tools/bench_jpeg_batch.py $args --seconds 0.6 2>&1 | grep "per call\|per frame" | sed "s/^/e$t [$args] /"
  done
done
done
