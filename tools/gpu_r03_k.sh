#!/bin/bash
# r03k: end-of-round tables that the final evidence call did not re-take: all decoders[] pairs at 8K, soak through the modules, JPEG decode table
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03k; mkdir -p $OUT
python tools/bench_pixfmt_all.py --json $OUT/pixfmt_all_8k.json > $OUT/pixfmt_all_8k.txt 2>&1; tail -2 $OUT/pixfmt_all_8k.txt | cut -c1-300
bash tools/soak.sh > $OUT/soak.txt 2>&1; cat $OUT/soak.txt
python tools/bench_jpeg_decode.py --json $OUT/jpeg_decode.json > $OUT/jpeg_decode.txt 2>&1; grep -v amdgpu $OUT/jpeg_decode.txt | head -8
