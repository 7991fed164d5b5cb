#!/bin/bash
# validation of the -c jpeg scan layouts on the GPU: option tests, the random search, 4K RGB throughput per layout (writes gpurun_out/nonint_check.txt)
set -u
out=gpurun_out/nonint_check.txt
mkdir -p gpurun_out
{
python -m pytest tests/test_jpeg_colour_options.py tests/test_gpu_jpeg.py -q 2>&1 | tail -3
python -m pytest tests/test_module_harness.py tests/test_runtime_conventions.py -q -k jpeg 2>&1 | tail -3
timeout 600 python tools/find_encode_mismatch.py 2500 2>&1 | tail -3
python - <<'PY'
import numpy as np
from ultragrid_amd import synth
fr = [synth.s2_video("RGB", 3840, 2160, salt=i) for i in range(2)]
np.concatenate([fr[i % 2] for i in range(8)]).tofile("/tmp/4k_rgb.raw")
PY
for cfg in "jpeg:q=75:restart=4" "jpeg:q=75:restart=4:interleaved" "jpeg:q=75:restart=4:Y601full" "jpeg:q=75:restart=4:Y601full:interleaved" "jpeg:q=75:restart=4:workers=1" "jpeg:q=75:restart=4:interleaved:workers=1"; do
  echo "== $cfg"
  timeout 300 oracle/_ref/ug_harness "$cfg" RGB 3840 2160 /tmp/4k_rgb.raw /tmp/o.bin 1 host 8 50 2>&1 | grep THROUGHPUT
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
} > $out 2>&1
cat $out
