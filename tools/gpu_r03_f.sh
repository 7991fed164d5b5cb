O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests/test_module_harness.py tests/test_abi.py tests/test_gpu_jpeg.py -m gpu -q 2>&1 | tail -8 > $O/pytest_mod.log; tail -8 $O/pytest_mod.log
python - <<'PY'
import numpy as np
from ultragrid_amd import synth
fr = [synth.s2_video("UYVY", 7680, 4320, salt=i) for i in range(2)]
np.concatenate([fr[i % 2] for i in range(4)]).tofile("/tmp/8k.raw")
PY
for lanes in 0 1 0 1; do
  echo "== UG_MI355X_COPY_LANES=$lanes"
  UG_MI355X_COPY_LANES=$lanes CFGS="dxt:DXT5 jpeg:q=75:restart=4 jpeg:q=75:restart=4:batch=8:workers=1 jpeg:q=75:restart=4:batch=8" REPEAT=300 bash tools/soak.sh 2>&1 | grep -E "==|THROUGHPUT"
  for cfg in dxt:DXT5 dxt:DXT5:workers=3; do UG_MI355X_COPY_LANES=$lanes oracle/_ref/ug_harness $cfg UYVY 7680 4320 /tmp/8k.raw /tmp/o.bin 1 host 4 200 2>&1 | grep THROUGHPUT | sed "s/^/8K $cfg /"; done
done 2>&1 | tee $O/module_lanes_ab.txt
