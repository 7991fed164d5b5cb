#!/bin/bash
# Interleaved A/B: product (plain loads) vs -DUG_STREAM_LOADS
A=tools/ab/libB_product.so; B=tools/ab/libC_stream_loads.so
for r in 1 2; do
for lib in $A $B; do
  n=$(basename $lib .so)
  for cfg in "UYVY RGB 3840 2160 1" "UYVY RGBA 3840 2160 1" "RGB UYVY 3840 2160 1" "UYVY RGB 3840 2160 8" "RG48 RGB 3840 2160 1" "R10k RGBA 3840 2160 1" "Y416 UYVY 3840 2160 8"; do echo -n "$n "; UG_MI355X_LIB=$(realpath $lib) python tools/one_pixfmt.py $cfg 2>&1 | grep -v amdgpu.ids; done
  for cfg in "DXT5 RGBA 1" "DXT5 RGBA 8" "DXT5 UYVY 8"; do echo -n "$n "; UG_MI355X_LIB=$(realpath $lib) python tools/one_decode.py $cfg 200 2>&1 | grep -v amdgpu.ids; done
done
done
