O=gpurun_out/r03g; mkdir -p $O
python -m pytest tests/test_module_harness.py -m gpu -q 2>&1 | tail -4 > $O/pytest_mod.log; tail -4 $O/pytest_mod.log
for sub in 420 422; do python tools/bench_jpeg_batch.py --sub $sub 2>&1 | grep -v amdgpu.ids; done | tee $O/jpeg_batch.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/tr -o jb -- python $GRAFT_REPO_ROOT/tools/bench_jpeg_batch.py --sub 420 --seconds 0.5 > $GRAFT_REPO_ROOT/$O/tr.log 2>&1)
python tools/pmc_summary.py $O/tr/jb_results.db > $O/jpeg_batch_trace.txt 2>&1; grep -A1 "^kernel" $O/jpeg_batch_trace.txt | grep -v "^--" | cut -c1-150 | head -24; rm -rf $O/tr
python - <<'PY'
import numpy as np
from ultragrid_amd import synth
fr = [synth.s2_video("UYVY", 7680, 4320, salt=i) for i in range(2)]
np.concatenate([fr[i % 2] for i in range(4)]).tofile("/tmp/8k.raw")
PY
for round in 1 2; do
  CFGS="dxt:DXT5 dxt:DXT5:batch=8 dxt:DXT5:workers=1 dxt:DXT5:workers=4 jpeg:q=75:restart=4 jpeg:q=75:restart=4:batch=8 jpeg:q=75:restart=4:batch=8:workers=1" REPEAT=400 bash tools/soak.sh 2>&1 | grep -E "==|THROUGHPUT"
  for cfg in dxt:DXT5 dxt:DXT5:batch=4 dxt:DXT5:workers=1 dxt:DXT5:workers=4; do oracle/_ref/ug_harness $cfg UYVY 7680 4320 /tmp/8k.raw /tmp/o.bin 1 host 4 250 2>&1 | grep THROUGHPUT | sed "s/^/8K $cfg /"; done
done 2>&1 | tee $O/module_fps.txt
