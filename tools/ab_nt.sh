#!/bin/bash
# Interleaved A/B of the streaming-store library against the plain-store build over the kernel families (GPU box)
A=tools/ab/libA_plain_stores.so; B=tools/ab/libB_product.so
for r in 1 2; do
for lib in $A $B; do
  n=$(basename $lib .so)
  for cfg in "v210 UYVY 3840 2160 1" "UYVY RGB 3840 2160 1" "UYVY RGBA 3840 2160 1" "RGB UYVY 3840 2160 1" "UYVY RGB 3840 2160 8" "UYVY RGBA 3840 2160 8" "v210 UYVY 7680 4320 1" "RG48 RGB 3840 2160 1" "v210 RGB 3840 2160 1" "RGBA RGB 3840 2160 1"; do echo -n "$n "; UG_MI355X_LIB=$(realpath $lib) python tools/one_pixfmt.py $cfg 2>&1 | grep -v amdgpu.ids; done
done
done
