#!/bin/bash
# Interleaved A/B of the streaming-store library against the plain-store build over the kernel families (GPU box)
A=tools/ab/libA_plain_stores.so; B=tools/ab/libB_product.so
for r in 1 2; do
for lib in $A $B; do
  n=$(basename $lib .so)
  UG_MI355X_LIB=$(realpath $lib) python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', 'UYVY->DXT5 4K x16', d['roofline']['ms_per_launch'], 'ms/launch', d['roofline']['frac'])"
  for cfg in "UYVY DXT5 3840 2160 1" "RGB DXT1 1920 1080 1" "UYVY DXT5 7680 4320 1"; do echo -n "$n "; UG_MI355X_LIB=$(realpath $lib) python tools/one_kernel.py $cfg 300 2>&1 | grep -v amdgpu.ids; done
  for cfg in "v210 UYVY 3840 2160 1" "UYVY RGB 3840 2160 1" "UYVY RGBA 3840 2160 1" "RGB UYVY 3840 2160 1" "UYVY RGB 3840 2160 8" "v210 UYVY 7680 4320 1" "RG48 RGB 3840 2160 1"; do echo -n "$n "; UG_MI355X_LIB=$(realpath $lib) python tools/one_pixfmt.py $cfg 2>&1 | grep -v amdgpu.ids; done
  for cfg in "DXT5 RGBA 1" "DXT5 RGBA 8" "DXT5 UYVY 8" "DXT1 RGBA 1" "DXT1 RGBA 8"; do echo -n "$n "; UG_MI355X_LIB=$(realpath $lib) python tools/one_decode.py $cfg 200 2>&1 | grep -v amdgpu.ids; done
done
done
