O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests/test_gpu_dxt_decode.py -m gpu -q 2>&1 | tail -5 > $O/pytest_dec.log; tail -5 $O/pytest_dec.log
for rpw in 1 2 4 8; do for cfg in "DXT5 RGBA 8" "DXT5 RGB 8" "DXT5 UYVY 8" "DXT5 RGBA 1"; do echo -n "rpw=$rpw "; UG_DXT5_DEC_RPW=$rpw python tools/one_decode.py $cfg 200 2>&1 | grep -v amdgpu.ids; done; done | tee $O/decode_timing_rpw.txt
for cfg in "DXT5 RGBA 8" "DXT5 RGBA 1" "DXT5 UYVY 8"; do echo -n "default "; python tools/one_decode.py $cfg 200 2>&1 | grep -v amdgpu.ids; done | tee -a $O/decode_timing_rpw.txt
