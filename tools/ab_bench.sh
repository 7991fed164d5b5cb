#!/bin/bash
# Interleaved A/B of kernel-library builds on the GPU box: tools/ab_bench.sh libA.so libB.so [...]
# WORKLOAD=<bench.py --workload name> selects another configuration.  Prints ms per launch (HIP events) of the headline bench for each library, ROUNDS times, interleaved.
ROUNDS=${ROUNDS:-3}
STEPS=${STEPS:-300}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    UG_MI355X_LIB=$(realpath $lib) python bench.py --steps $STEPS --warmup 20 --no-cpu-baseline --no-e2e --no-configs --no-parity-check ${WORKLOAD:+--workload $WORKLOAD} 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['ms_per_launch'], d['roofline']['frac'])"
  done
done
