#!/bin/bash
# r03m: ug::WaveWords in planar_api (group stores), the decoders' RGB rows and v210_to_p010le -- parity, tables of both builds, bytes written per row
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03m
python -m pytest tests/test_gpu_pixfmt.py tests/test_gpu_pixfmt_ext.py tests/test_gpu_dxt_decode.py tests/test_planar_api.py tests/test_lavc_conv.py tests/test_lavc_hook.py tests/test_module_harness.py -q -x -m gpu 2>&1 | grep -v lavc_vid_conv | tail -2
for r in 1 2; do
for lib in tools/ab/libC_wavewords.so tools/ab/libD_wavewords2.so; do
  n=$(basename $lib .so)
  UG_MI355X_LIB=$(realpath $lib) python tools/bench_decode.py 2>&1 | grep -E "RGB " | sed "s/^/$n /"
  UG_MI355X_LIB=$(realpath $lib) python tools/bench_kernels.py 2>&1 | grep -E "from_planar|rgba_to_bgra|v210_to_p010le  |av_to_uv gbrp|uv_to_av v210->p010le" | sed "s/^/$n /"
done
done | tee gpurun_out/r03m/ab.txt
bash tools/pmc_write_by_row.sh tools/ab/libD_wavewords2.so > /dev/null 2>&1
