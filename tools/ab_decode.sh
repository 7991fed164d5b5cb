#!/bin/bash
# Interleaved A/B of kernel-library builds on the DXT decoders (GPU box): tools/ab_decode.sh "<IN OUT FRAMES>" libA.so libB.so [...]
ROUNDS=${ROUNDS:-3}
CFG=$1; shift
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    echo -n "$(basename $lib) "; UG_MI355X_LIB=$(realpath $lib) python tools/one_decode.py $CFG 300 2>&1 | grep -v amdgpu.ids
  done
done
