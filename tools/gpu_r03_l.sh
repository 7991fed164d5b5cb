#!/bin/bash
# r03l: whole-line stores for the multi-word units of the pixfmt fast paths (ug::WaveWords) -- parity, bytes written per row, interleaved timing
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03l
python -m pytest tests/test_gpu_pixfmt.py tests/test_gpu_pixfmt_ext.py tests/test_gpu_dxt_decode.py tests/test_planar_api.py tests/test_lavc_conv.py -q -x -m gpu 2>&1 | grep -v lavc_vid_conv | tail -2
for r in 1 2; do
for lib in ${LIBS:-tools/ab/libA_plain_stores.so tools/ab/libB_product.so tools/ab/libC_wavewords.so}; do
  n=$(basename $lib .so)
  for cfg in "v210 UYVY 3840 2160 1" "UYVY RGB 3840 2160 1" "UYVY RGBA 3840 2160 1" "v210 RGB 3840 2160 1" "RGBA RGB 3840 2160 1" "UYVY v210 3840 2160 1" "UYVY RGB 3840 2160 8" "v210 RGB 3840 2160 8" "UYVY RGB 7680 4320 1" "v210 UYVY 7680 4320 1"; do echo -n "$n "; UG_MI355X_LIB=$(realpath $lib) python tools/one_pixfmt.py $cfg 2>&1 | grep -v amdgpu.ids; done
done
done | tee gpurun_out/r03l/ab.txt
bash tools/pmc_write_by_row.sh ${WLIB:-tools/ab/libC_wavewords.so} > /dev/null 2>&1
