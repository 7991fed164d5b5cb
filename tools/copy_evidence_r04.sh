#!/bin/bash
# gpurun_out/r04m/ (tools/gpu_r04_m.sh on the GPU box) -> profiles/r04_*
set -e
S=${1:-gpurun_out/r04m}; D=profiles
cp $S/bench_line.json $D/r04_bench_line.json
cp $S/bench_4k-uyvy-jpeg420.json $D/r04_bench_4k_jpeg420.json
cp $S/8k-v210.txt $D/r04_pmc_8k_v210.txt; cp $S/1080p-rgb-dxt1.txt $D/r04_pmc_1080p_rgb_dxt1.txt; cp $S/4k-uyvy-jpeg420.txt $D/r04_pmc_4k_uyvy_jpeg420.txt
cp $S/pmc_traffic.json $D/pmc_traffic.json
cp $S/kernels.json $D/r04_kernels.json; grep -v amdgpu.ids $S/kernels_table.txt > $D/r04_all_kernels_table.txt
cp $S/decode.json $D/r04_decode.json; grep -v amdgpu.ids $S/decode.txt > $D/r04_decode.txt
cp $S/pixfmt_all_8k.json $D/r04_pixfmt_all_8k.json
tail -2 $S/pytest.log | head -1 > $D/r04_gpu_tests.txt
grep -E "^==|THROUGHPUT|^OK" $S/soak.txt | cut -c1-200 > $D/r04_soak_numa_ab.txt
grep -v "CatArray\|at::native" $S/jpeg_batch_traffic.txt | sed 's#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#' > $D/r04_jpeg_batch_traffic.txt
ls $D | grep -c r04
# the end-of-round call (tools/gpu_r04_final.sh -> gpurun_out/r04n/): bench line + the kernel trace of the same command on the same box, the
# other workloads' lines, the JPEG encoder's rates at every sampling / size, the per-kernel table, the random searches
N=gpurun_out/r04n
if [ -d $N ]; then
  cp $N/bench_line.json $D/r04_bench_line.json
  sed 's#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#' $N/kernel_trace.txt > $D/r04_kernel_trace.txt
  cp $N/bench_4k-uyvy-jpeg420.json $D/r04_bench_4k_jpeg420.json; cp $N/bench_8k-v210.json $D/r04_bench_8k_v210.json; cp $N/bench_1080p-rgb-dxt1.json $D/r04_bench_1080p_rgb_dxt1.json
  cp $N/jpeg_batch_all.txt $D/r04_jpeg_batch_all.txt
  cp $N/kernels.json $D/r04_kernels.json; grep -v amdgpu.ids $N/kernels_table.txt > $D/r04_all_kernels_table.txt
  tail -2 $N/pytest.log | head -1 > $D/r04_gpu_tests.txt
  { echo "# end of round 4: tools/find_encode_mismatch.py 1500 (half of the cases on the fused kernels, half as two-frame batches), tools/find_dxt_mismatch.py 1500, tools/find_module_mismatch.py"; grep -v amdgpu.ids $N/find_encode.txt; grep -v amdgpu.ids $N/find_dxt.txt; grep -v amdgpu.ids $N/find_module.txt; } > $D/r04_random_searches.txt
fi
# the closing call of the round (tools/gpu_r04_final.sh as it stands -> gpurun_out/r04z/), on the build the round ends on: it replaces what the
# earlier calls left wherever both have the file
Z=gpurun_out/r04z
if [ -d $Z ]; then
  cp $Z/bench_line.json $D/r04_bench_line.json
  cp $Z/kernel_trace.txt $D/r04_kernel_trace.txt
  cp $Z/bench_4k-uyvy-jpeg420.json $D/r04_bench_4k_jpeg420.json; cp $Z/bench_8k-v210.json $D/r04_bench_8k_v210.json; cp $Z/bench_1080p-rgb-dxt1.json $D/r04_bench_1080p_rgb_dxt1.json
  cp $Z/bench_4k-uyvy-jpeg-encode.json $D/r04_bench_4k_jpeg_encode.json
  cp $Z/pmc_uyvy_dxt5_4k_x16.txt $D/r04_pmc_uyvy_dxt5_4k_x16.txt
  cp $Z/8k-v210.txt $D/r04_pmc_8k_v210.txt; cp $Z/1080p-rgb-dxt1.txt $D/r04_pmc_1080p_rgb_dxt1.txt; cp $Z/4k-uyvy-jpeg420.txt $D/r04_pmc_4k_uyvy_jpeg420.txt
  cp $Z/pmc_traffic.json $D/pmc_traffic.json
  cp $Z/jpeg_batch_all.txt $D/r04_jpeg_batch_all.txt
  cp $Z/jpeg_batch_pmc.txt $D/r04_jpeg_batch_pmc.txt; cp $Z/jpeg_batch_trace_422_444.txt $D/r04_jpeg_batch_trace_422_444.txt
  { echo "# phase clock of the fused encoder kernel on the closing build of round 4 (UG_JPEG_PROF=1 python tools/bench_jpeg_batch.py --sub S --only batch --calls 40): cycles per"
    echo "# workgroup, wave 0's view: front end | tables + DC values (waiting for the other waves' front ends) | walk | positions | merge | byte counts + prefix | slot / look-back | write-out"
    echo "# (the mid-round file with the look-back A/B and the phase clock of the segment-by-segment write-out: r04_jpeg_batch_midround.txt)"
    cat $Z/jpeg_phase_clock.txt; } > $D/r04_jpeg_batch.txt
  cp $Z/kernels.json $D/r04_kernels.json; grep -v amdgpu.ids $Z/kernels_table.txt > $D/r04_all_kernels_table.txt
  cp $Z/decode.json $D/r04_decode.json; grep -v amdgpu.ids $Z/decode.txt > $D/r04_decode.txt
  cp $Z/pixfmt_all_8k.json $D/r04_pixfmt_all_8k.json
  grep -v amdgpu.ids $Z/deinterlace.txt > $D/r04_deinterlace.txt
  tail -2 $Z/pytest.log | head -1 > $D/r04_gpu_tests.txt
  { echo "# end of round 4: tools/find_encode_mismatch.py 2000 (half of the cases on the fused kernels, half as two-frame batches), tools/find_dxt_mismatch.py 1500, tools/find_module_mismatch.py, tools/find_decode_mismatch_valid.py 3000"
    grep -v amdgpu.ids $Z/find_encode.txt; grep -v amdgpu.ids $Z/find_dxt.txt; grep -v amdgpu.ids $Z/find_module.txt; grep -v amdgpu.ids $Z/find_decode_valid.txt
    if [ -f $Z/find_deinterlace.txt ]; then echo "# after the closing call: tools/find_deinterlace_mismatch.py 3000 (random geometries of ug_hip_deinterlace_blend[_batch] against the oracle), tools/find_encode_mismatch.py 6000"; tail -1 $Z/find_deinterlace.txt; tail -1 $Z/find_encode_6000.txt; fi
    if [ -f $Z/find_libjpeg.txt ]; then echo "# tools/find_libjpeg_mismatch.py 3000: the product's RGB 4:4:4 / UYVY 4:2:2 / 4:2:0 streams against libjpeg-turbo (float DCT) on random sizes, qualities 1..100, restart intervals 1..64"; tail -1 $Z/find_libjpeg.txt; fi; } > $D/r04_random_searches.txt
fi
