#!/bin/bash
# Round 4, session E: de-interlace (C-ABI + module), phase clock of the JPEG kernels, the small end of the rotation sweep.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_deinterlace.py "tests/test_module_harness.py" -k "deinterlace or interlaced or batched_dxt or compress_frame_through" -q -x 2>&1 | grep -v lavc_vid_conv | tail -8 > $OUT/pytest.log; tail -8 $OUT/pytest.log
UG_JPEG_PROF=1 timeout 120 python tools/bench_jpeg_batch.py --only batch --calls 40 > $OUT/jpeg_prof_fused.txt 2>&1; grep "UG_JPEG_PROF\|frames per call" $OUT/jpeg_prof_fused.txt
UG_JPEG_PROF=1 UG_JPEG_FUSED=0 timeout 120 python tools/bench_jpeg_batch.py --only batch --calls 40 > $OUT/jpeg_prof_unfused.txt 2>&1; grep "UG_JPEG_PROF\|frames per call" $OUT/jpeg_prof_unfused.txt
timeout 900 python tools/rotation_sweep.py > $OUT/rotation_sweep.txt 2> $OUT/rotation_sweep.err; cat $OUT/rotation_sweep.txt; tail -2 $OUT/rotation_sweep.err
