#!/bin/bash
# gpurun_out/r03/ (written by tools/gpu_r03_final.sh on the GPU box) -> profiles/r03_*
set -e
S=gpurun_out/r03; D=profiles
cp $S/bench_line.json $D/r03_bench_line.json
cp $S/bench_8k-v210.json $D/r03_bench_8k_v210.json
cp $S/bench_1080p-rgb-dxt1.json $D/r03_bench_1080p_rgb_dxt1.json
cp $S/bench_4k-uyvy-jpeg420.json $D/r03_bench_4k_jpeg420.json
cp $S/kernel_trace.txt $D/r03_kernel_trace.txt
sed 's#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#' $S/pmc_summary.txt > $D/r03_pmc_uyvy_dxt5_4k_x16.txt
cp $S/8k-v210.txt $D/r03_pmc_8k_v210.txt; cp $S/1080p-rgb-dxt1.txt $D/r03_pmc_1080p_rgb_dxt1.txt; cp $S/4k-uyvy-jpeg420.txt $D/r03_pmc_4k_uyvy_jpeg420.txt
cp $S/pmc_traffic.json $D/pmc_traffic.json
cp $S/kernels.json $D/r03_kernels.json; grep -v amdgpu.ids $S/kernels_table.txt > $D/r03_all_kernels_table.txt
cp $S/decode.json $D/r03_decode.json; grep -v amdgpu.ids $S/decode.txt > $D/r03_decode.txt
tail -2 $S/pytest.log | head -1 > $D/r03_gpu_tests.txt
{ echo "# end of round 3 (after the fast index stages): tools/find_dxt_mismatch.py 8000, tools/find_module_mismatch.py"; grep -v amdgpu.ids $S/find_dxt.txt; grep -v amdgpu.ids $S/find_module.txt; } > $D/r03_random_searches_final.txt
ls $D | grep -c r03_
