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
#   jpeg1            the one-frame JPEG call: tools/bench_jpeg_batch.py --only single variants (+ UG_JPEG_* A/B switches given in $JPEG_AB)
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
      tail -1 $OUT/trace.log > $OUT/trace_bench_line.json; rm -rf $OUT/trace ;;
    jpeg1)
      { for sub in 420 422; do timeout 100 python tools/bench_jpeg_batch.py --sub $sub --only single; done
        for ab in $JPEG_AB; do env $ab timeout 100 python tools/bench_jpeg_batch.py --only single | sed "s/^/$ab /"; done; } 2>&1 | grep "per call\|us" > $OUT/jpeg_one_frame.txt; cat $OUT/jpeg_one_frame.txt ;;
    cmd:*)    bash -c "${step#cmd:}" 2>&1 | tail -60 ;;
    *)        echo "unknown step $step" ;;
  esac
done
ls $OUT
