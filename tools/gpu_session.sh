#!/bin/bash
# One parameterised GPU session (replaces the per-session gpu_r0x_*.sh scripts of the earlier rounds; what those produced is in profiles/).
#   gpurun --timeout <s> -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'
# Everything lands in gpurun_out/<tag>/ (scratch); what is kept is copied into profiles/ by hand, named per round.
# steps:
#   tests            the whole GPU suite
#   tests:<expr>     pytest -m gpu -k <expr>
#   runtime          tests/test_runtime_conventions.py on the product modules, then the same under the ASan / TSan builds of harness + modules
#   bench            the default bench.py line
#   bench:<workload> bench.py --workload <workload> --no-e2e
#   trace            rocprofv3 --kernel-trace --stats of the default bench command (no CPU legs) -> kernel_trace.txt
#   pmc              rocprofv3 --pmc passes of the five bench workloads' kernels -> profiles-ready summaries + pmc_traffic.json (what bench.py quotes as `traffic`)
#   jpeg             the JPEG encoder's rates in every call form (tools/bench_jpeg_batch.py), kernel trace and phase clock
#   tables           the per-kernel, decoder, pixel-format and de-interlace tables (2.4 GB of rotating buffers per row)
#   search           random searches against the oracles (encoder, decoder, module)
#   latency          one frame in flight through the reference's framework, bands=1..16 (tools/module_latency.sh)
#   cmd:<shell>      anything else
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); TAG=${1:-session}; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for step in "$@"; do
  echo "=== $step"
  case "$step" in
    tests)    timeout 2400 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -25 > $OUT/pytest.log; tail -4 $OUT/pytest.log ;;
    tests:*)  timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 -k "${step#tests:}" 2>&1 | tail -40 > $OUT/pytest_k.log; tail -15 $OUT/pytest_k.log ;;
    runtime)
      timeout 1500 python -m pytest tests/test_runtime_conventions.py -q --maxfail=20 2>&1 | tail -40 > $OUT/runtime_plain.log; tail -5 $OUT/runtime_plain.log
      timeout 1500 python -m pytest tests/test_runtime_conventions.py -m gpu -q -k "asan or tsan" --maxfail=40 2>&1 | tail -120 > $OUT/runtime_sanitized.log; tail -8 $OUT/runtime_sanitized.log ;;
    bench)    python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_line.json; tail -3 $OUT/bench.err ;;
    bench:*)  wl=${step#bench:}; python bench.py --workload $wl --no-e2e > $OUT/bench_$wl.json 2>> $OUT/bench.err; cut -c1-300 $OUT/bench_$wl.json ;;
    trace)
      ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-e2e > $OUT/trace.log 2>&1 )
      python tools/pmc_summary.py $(find $OUT/trace -name "*.db") 2>&1 | sed "s#$ROOT/##" > $OUT/kernel_trace.txt; head -6 $OUT/kernel_trace.txt | cut -c1-170
      grep "^{" $OUT/trace.log | tail -1 > $OUT/trace_bench_line.json; rm -rf $OUT/trace ;;
    pmc-jpeg)  # the two JPEG workloads only
      R=${TAG}; WORKLOADS="4k-uyvy-jpeg420 4k-uyvy-jpeg-encode" bash tools/pmc_workloads.sh > $OUT/pmc_workloads.log 2>&1; cp gpurun_out/pmc_workloads/*.txt $OUT/
      python tools/pmc_to_json.py uyvy_jpeg420_4k_x8 "uyvy_jpeg_fast_batch_kernel" "rocprof passes of session $R, uyvy_jpeg_fast_batch_kernel<420>" $OUT/4k-uyvy-jpeg420.txt
      python tools/pmc_to_json.py uyvy_jpeg_encode_4k_x8 "jpeg_code_kernel<3, 420" "rocprof passes of session $R, jpeg_code_kernel<3,420>: the fused encoder kernel of ug_hip_jpeg_encoder_encode_batch, 8 frames per launch (jpeg_gather_kernel beside it moves the stream bytes once more)" $OUT/4k-uyvy-jpeg-encode.txt
      cp profiles/pmc_traffic.json $OUT/pmc_traffic.json ;;
    pmc)
      R=${TAG}; bash tools/pmc_collect.sh $R > $OUT/pmc.log 2>&1; sed "s#$ROOT/##" gpurun_out/pmc_$R/summary.txt > $OUT/pmc_uyvy_dxt5_4k_x16.txt; rm -f gpurun_out/pmc_$R/*.db
      python tools/pmc_to_json.py uyvy_dxt5_4k_x16 "dxt_encode_kernel<2, 6" "rocprof passes of session $R, dxt_encode_kernel<UYVY,DXT5,ties even>" $OUT/pmc_uyvy_dxt5_4k_x16.txt
      bash tools/pmc_workloads.sh > $OUT/pmc_workloads.log 2>&1; cp gpurun_out/pmc_workloads/*.txt $OUT/
      python tools/pmc_to_json.py v210_dxt5_8k_x4 "dxt_encode_kernel<6, 6" "rocprof passes of session $R, dxt_encode_kernel<v210,DXT5,ties even>" $OUT/8k-v210.txt
      python tools/pmc_to_json.py rgb_dxt1_1080p_x64 "dxt_encode_kernel<4, 1" "rocprof passes of session $R, dxt_encode_kernel<RGB,DXT1,ties even>" $OUT/1080p-rgb-dxt1.txt
      python tools/pmc_to_json.py uyvy_jpeg420_4k_x8 "uyvy_jpeg_fast_batch_kernel" "rocprof passes of session $R, uyvy_jpeg_fast_batch_kernel<420>" $OUT/4k-uyvy-jpeg420.txt
      python tools/pmc_to_json.py uyvy_jpeg_encode_4k_x8 "jpeg_code_kernel<3, 420" "rocprof passes of session $R, jpeg_code_kernel<3,420>: the fused encoder kernel of ug_hip_jpeg_encoder_encode_batch, 8 frames per launch (jpeg_gather_kernel beside it moves the stream bytes once more)" $OUT/4k-uyvy-jpeg-encode.txt
      cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; grep -c . $OUT/pmc_traffic.json ;;
    jpeg)
      { timeout 100 python tools/bench_jpeg_batch.py; timeout 100 python tools/bench_jpeg_batch.py --n 16 --only batch; timeout 100 python tools/bench_jpeg_batch.py --sub 422; timeout 100 python tools/bench_jpeg_batch.py --sub 444;
        timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 7680x4320 --n 4; timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 1920x1080 --n 16;
        for ab in UG_JPEG_FLAT=0 UG_JPEG_LOOKBACK=0 UG_JPEG_TICKET=1; do env $ab timeout 100 python tools/bench_jpeg_batch.py --only single | sed "s/^/$ab /"; done; } 2>&1 | grep "per call\|per frame" > $OUT/jpeg_batch_all.txt; cat $OUT/jpeg_batch_all.txt
      ( cd /tmp && export TMPDIR=/tmp; for only in single batch; do timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt_$only -o t -- python $ROOT/tools/bench_jpeg_batch.py --only $only --calls 200 > $OUT/jt_$only.log 2>&1; done )
      for only in single batch; do echo "== --only $only"; python tools/pmc_summary.py $(find $OUT/jt_$only -name "*.db") 2>&1 | grep -v "copyBuffer\|roll\|elementwise\|fillBuffer\|CatArray\|at::native" | head -6 | cut -c1-200; done > $OUT/jpeg_kernel_trace.txt; rm -rf $OUT/jt_single $OUT/jt_batch
      for only in single batch; do UG_JPEG_PROF=1 timeout 120 python tools/bench_jpeg_batch.py --only $only --calls 100 2>&1 | grep "UG_JPEG_PROF" | sed "s/^/$only /"; done > $OUT/jpeg_phase_clock.txt; cat $OUT/jpeg_kernel_trace.txt $OUT/jpeg_phase_clock.txt ;;
    tables)
      timeout 900 python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; grep -c . $OUT/kernels_table.txt
      timeout 300 python tools/bench_decode.py --json $OUT/decode.json > $OUT/decode.txt 2>&1; tail -2 $OUT/decode.txt
      timeout 600 python tools/bench_pixfmt_all.py --json $OUT/pixfmt_all_8k.json > $OUT/pixfmt_all_8k.txt 2>&1; tail -2 $OUT/pixfmt_all_8k.txt
      timeout 120 python tools/bench_deinterlace.py > $OUT/deinterlace.txt 2>&1; cat $OUT/deinterlace.txt ;;
    search)
      timeout 600 python tools/find_encode_mismatch.py 2000 2>&1 | tail -2 > $OUT/find_encode.txt; cat $OUT/find_encode.txt
      timeout 600 python tools/find_dxt_mismatch.py 1500 2>&1 | tail -2 > $OUT/find_dxt.txt; cat $OUT/find_dxt.txt
      timeout 300 python tools/find_module_mismatch.py 2>&1 | tail -2 > $OUT/find_module.txt; cat $OUT/find_module.txt
      timeout 300 python tools/find_decode_mismatch_valid.py 3000 2>&1 | tail -2 > $OUT/find_decode_valid.txt; cat $OUT/find_decode_valid.txt ;;
    latency)  bash tools/module_latency.sh 2>&1 | tee $OUT/module_latency.txt | grep "bands=[148]:" ;;
    cmd:*)    bash -c "${step#cmd:}" 2>&1 | tail -60 ;;
    *)        echo "unknown step $step" ;;
  esac
done
ls $OUT
