#!/bin/bash
# gpurun_out/<tag>/ (written by tools/gpu_evidence.sh on the GPU box) -> profiles/r02_*: usage tools/copy_evidence.sh <tag>
set -e
S=gpurun_out/$1; D=profiles
cp $S/bench_line.json $D/r02_bench_line.json
cp $S/bench_8k-v210.json $D/r02_bench_8k_v210.json
cp $S/bench_1080p-rgb-dxt1.json $D/r02_bench_1080p_rgb_dxt1.json
cp $S/bench_4k-uyvy-jpeg420.json $D/r02_bench_4k_jpeg420.json
cp $S/kernel_trace.txt $D/r02_kernel_trace.txt
cp $S/kernel_trace_jpeg420.txt $D/r02_kernel_trace_jpeg420.txt
cp $S/pmc_summary.txt $D/r02_pmc_uyvy_dxt5_4k_x16.txt
cp $S/kernels.json $D/r02_kernels.json
cp $S/kernels_table.txt $D/r02_all_kernels_table.txt
cp $S/all_kernels_trace.txt $D/r02_all_kernels_trace.txt
cp $S/pixfmt_all_8k.json $D/r02_pixfmt_all_8k.json
cp $S/decode.json $D/r02_decode.json; cp $S/decode.txt $D/r02_decode.txt
cp $S/jpeg_encoder_trace.txt $D/r02_jpeg_encoder_trace.txt
cp $S/e2e_bench.txt $D/r02_e2e_bench.jsonl
cp $S/module_fps.txt $D/r02_module_fps.txt
cp $S/soak.txt $D/r02_soak.txt
cp $S/cpu_reference_pixfmt.json $D/r02_cpu_reference_pixfmt.json; cp $S/cpu_reference_pixfmt.txt $D/r02_cpu_reference_pixfmt.txt
cp $S/jpeg_decode.json $D/r02_jpeg_decode.json; [ -f $S/jpeg_decode_8k.json ] && cp $S/jpeg_decode_8k.json $D/r02_jpeg_decode_8k.json; grep -v amdgpu.ids $S/jpeg_decode.txt > $D/r02_jpeg_decode.txt
sed 's#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#' $S/jpeg_decoder_pmc.txt > $D/r02_jpeg_decoder_pmc.txt
[ -f $S/16lane.txt ] && cp $S/16lane.txt $D/r02_16lane_experiment.txt
tail -2 $S/pytest.log | head -1 > $D/r02_gpu_tests.txt
ls $D | grep -c r02_
