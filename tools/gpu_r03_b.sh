# round 3, GPU session B: fixed-point DXT5 decoder (equivalence test, timing), batched JPEG encode tests, full GPU suite
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_gpu_dxt_decode.py tests/test_gpu_jpeg.py tests/test_jpeg_rtp_compat.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_new.log; tail -15 $O/pytest_new.log
for cfg in "DXT5 RGBA 8" "DXT5 RGB 8" "DXT5 UYVY 8" "DXT5 RGBA 1"; do python tools/one_decode.py $cfg 200; done 2>&1 | tee $O/decode_timing.txt
( time python -m pytest tests -m gpu -q --maxfail=20 ) 2>&1 | grep -v "lavc_vid_conv\|Using CUDA FFmpeg" | tail -30 > $O/pytest.log; tail -6 $O/pytest.log
