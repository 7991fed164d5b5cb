O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests/test_gpu_dxt_decode.py tests/test_module_harness.py -m gpu -q 2>&1 | tail -5 > $O/pytest_dec.log; tail -5 $O/pytest_dec.log
for cfg in "DXT5 RGBA 8" "DXT5 RGB 8" "DXT5 UYVY 8" "DXT5 RGBA 1" "DXT5 RGB 1" "DXT5 UYVY 1" "DXT1 RGBA 8" "DXT1 UYVY 8"; do python tools/one_decode.py $cfg 200 2>&1 | grep -v amdgpu.ids; done | tee $O/decode_timing.txt
