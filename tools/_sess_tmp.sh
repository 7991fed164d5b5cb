cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r05d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_jpeg.py -m gpu -q -k "fused_encoder_equals or full_4k_frame" 2>&1 | tail -8
for i in 1 2; do
for ab in UG_JPEG_FLAT=0 UG_JPEG_FLAT=1; do for sub in 420 422; do env $ab timeout 100 python tools/bench_jpeg_batch.py --sub $sub --only single 2>&1 | grep "per frame" | tail -1 | sed "s/^/$ab /"; done; done
done | tee $OUT/flat_ab.txt
for ab in UG_JPEG_FLAT=0 UG_JPEG_FLAT=1; do env $ab timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 7680x4320 --only single 2>&1 | grep "per frame" | tail -1 | sed "s/^/$ab /"; env $ab timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 1920x1080 --only single 2>&1 | grep "per frame" | tail -1 | sed "s/^/$ab /"; done | tee -a $OUT/flat_ab.txt
UG_JPEG_FLAT=1 UG_JPEG_PROF=1 timeout 100 python tools/bench_jpeg_batch.py --only single --calls 200 2>&1 | grep "UG_JPEG_PROF" | tee $OUT/prof_flat.txt
( cd /tmp && export TMPDIR=/tmp && UG_JPEG_FLAT=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/jt1 -o t -- python $GRAFT_REPO_ROOT/tools/bench_jpeg_batch.py --only single --calls 400 > $GRAFT_REPO_ROOT/$OUT/jt1.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/jt1 -name "*.db") 2>&1 | grep -v "copyBuffer\|roll\|elementwise\|fillBuffer\|CatArray\|at::native" | head -3 | cut -c1-200 | tee $OUT/trace_single_flat.txt
rm -rf $OUT/jt1
