#!/bin/bash
# Round 4, end-of-round evidence in one gpurun call (the build the round ends on): the GPU suite; the counter passes of the five bench
# workloads' kernels (-> profiles/pmc_traffic.json); the bench lines + the rocprofv3 kernel trace of the default command; the JPEG encoder's
# rates, kernel trace, SQ / TCC counters and phase clock; the per-kernel, decoder and pixel-format tables; de-interlace times; random
# searches.  Lands in gpurun_out/r04z/; tools/copy_evidence_r04.sh copies what is kept.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04z; mkdir -p $OUT
FILTER="copyBuffer\|roll_cuda\|elementwise\|fillBuffer\|CatArray\|at::native"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log
# ---- counters: the headline kernel, the other DXT workloads, the JPEG front end, the JPEG encoder ----
bash tools/pmc_collect.sh r04 > $OUT/pmc.log 2>&1; sed "s#$ROOT/##" gpurun_out/pmc_r04/summary.txt > $OUT/pmc_uyvy_dxt5_4k_x16.txt; rm -f gpurun_out/pmc_r04/*.db
python tools/pmc_to_json.py uyvy_dxt5_4k_x16 "dxt_encode_kernel<2, 6" "rocprof passes of round 4 (profiles/r04_pmc_uyvy_dxt5_4k_x16.txt), dxt_encode_kernel<UYVY,DXT5,ties even>" $OUT/pmc_uyvy_dxt5_4k_x16.txt
bash tools/pmc_workloads.sh > $OUT/pmc_workloads.log 2>&1; cp gpurun_out/pmc_workloads/*.txt $OUT/
python tools/pmc_to_json.py v210_dxt5_8k_x4 "dxt_encode_kernel<6, 6" "rocprof passes of round 4 (profiles/r04_pmc_8k_v210.txt), dxt_encode_kernel<v210,DXT5,ties even>" $OUT/8k-v210.txt
python tools/pmc_to_json.py rgb_dxt1_1080p_x64 "dxt_encode_kernel<4, 1" "rocprof passes of round 4 (profiles/r04_pmc_1080p_rgb_dxt1.txt), dxt_encode_kernel<RGB,DXT1,ties even>" $OUT/1080p-rgb-dxt1.txt
python tools/pmc_to_json.py uyvy_jpeg420_4k_x8 "uyvy_jpeg_fast_kernel" "rocprof passes of round 4 (profiles/r04_pmc_4k_uyvy_jpeg420.txt), uyvy_jpeg_fast_kernel<420> batched (the configs[3] front end; the -c jpeg module's fused encoder kernel never writes the coefficients)" $OUT/4k-uyvy-jpeg420.txt
( cd /tmp && export TMPDIR=/tmp
  CMD="python $ROOT/tools/bench_jpeg_batch.py --only batch --calls 40"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt -o t -- $CMD > $OUT/jt.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/jp0 -o p -- $CMD > $OUT/jp0.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/jp1 -o p -- $CMD > $OUT/jp1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/jp3 -o p -- $CMD > $OUT/jp3.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/jp2 -o p -- $CMD > $OUT/jp2.log 2>&1
  for sub in 422 444; do timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt$sub -o t -- $CMD --sub $sub > $OUT/jt$sub.log 2>&1; done )
python tools/pmc_summary.py $(find $OUT/jt $OUT/jp0 $OUT/jp1 $OUT/jp2 $OUT/jp3 -name "*.db") 2>&1 | grep -v "$FILTER" | sed "s#$ROOT/##" > $OUT/jpeg_batch_pmc.txt
python tools/pmc_summary.py $(find $OUT/jt422 $OUT/jt444 -name "*.db") 2>&1 | grep -v "$FILTER" | sed "s#$ROOT/##" > $OUT/jpeg_batch_trace_422_444.txt
grep "^pmc" $OUT/jpeg_batch_pmc.txt | cut -c1-12,60-150; rm -rf $OUT/jt $OUT/jp0 $OUT/jp1 $OUT/jp2 $OUT/jp3 $OUT/jt422 $OUT/jt444
python tools/pmc_to_json.py uyvy_jpeg_encode_4k_x8 "jpeg_code_kernel<3, 420>" "rocprof passes of round 4 (profiles/r04_jpeg_batch_pmc.txt), jpeg_code_kernel<3,420>: the fused encoder kernel of ug_hip_jpeg_encoder_encode_batch, 8 frames per launch (jpeg_gather_kernel beside it moves the stream bytes once more)" $OUT/jpeg_batch_pmc.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
for sub in 420 422; do UG_JPEG_PROF=1 timeout 120 python tools/bench_jpeg_batch.py --sub $sub --only batch --calls 40 2>&1 | grep "UG_JPEG_PROF" | sed "s/^/$sub /"; done > $OUT/jpeg_phase_clock.txt; cat $OUT/jpeg_phase_clock.txt
# ---- the bench lines (after the counters: the lines quote pmc_traffic.json) + the kernel trace of the default command ----
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_line.json
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-e2e > $OUT/trace.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/trace -name "*.db") 2>&1 | sed "s#$ROOT/##" > $OUT/kernel_trace.txt; head -4 $OUT/kernel_trace.txt | cut -c1-160; tail -1 $OUT/trace.log > $OUT/trace_bench_line.json
rm -rf $OUT/trace
for wl in 4k-uyvy-jpeg420 8k-v210 1080p-rgb-dxt1 4k-uyvy-jpeg-encode; do python bench.py --workload $wl --no-e2e > $OUT/bench_$wl.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_$wl.json; done
{ timeout 100 python tools/bench_jpeg_batch.py; timeout 100 python tools/bench_jpeg_batch.py --n 16 --only batch; timeout 100 python tools/bench_jpeg_batch.py --sub 422; timeout 100 python tools/bench_jpeg_batch.py --sub 444;
  timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 7680x4320 --n 4; timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 1920x1080 --n 16;
  UG_JPEG_LOOKBACK=1 timeout 100 python tools/bench_jpeg_batch.py --only batch | sed 's/^/UG_JPEG_LOOKBACK=1 /'; UG_JPEG_FUSED=0 timeout 100 python tools/bench_jpeg_batch.py --only batch | sed 's/^/UG_JPEG_FUSED=0 /'; } 2>&1 | grep "per call" > $OUT/jpeg_batch_all.txt; cat $OUT/jpeg_batch_all.txt
# ---- tables ----
timeout 900 python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; grep -c . $OUT/kernels_table.txt
timeout 300 python tools/bench_decode.py --json $OUT/decode.json > $OUT/decode.txt 2>&1; tail -2 $OUT/decode.txt
timeout 600 python tools/bench_pixfmt_all.py --json $OUT/pixfmt_all_8k.json > $OUT/pixfmt_all_8k.txt 2>&1; tail -2 $OUT/pixfmt_all_8k.txt
timeout 120 python tools/bench_deinterlace.py > $OUT/deinterlace.txt 2>&1; cat $OUT/deinterlace.txt
# ---- random searches against the oracles ----
timeout 600 python tools/find_encode_mismatch.py 2000 2>&1 | tail -2 > $OUT/find_encode.txt; cat $OUT/find_encode.txt
timeout 600 python tools/find_dxt_mismatch.py 1500 2>&1 | tail -2 > $OUT/find_dxt.txt; cat $OUT/find_dxt.txt
timeout 300 python tools/find_module_mismatch.py 2>&1 | tail -2 > $OUT/find_module.txt; cat $OUT/find_module.txt
timeout 300 python tools/find_decode_mismatch_valid.py 3000 2>&1 | tail -2 > $OUT/find_decode_valid.txt; cat $OUT/find_decode_valid.txt
ls $OUT
