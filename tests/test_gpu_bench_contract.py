"""bench.py's driver contract on the GPU box: the one-line JSON at N = 1, and the N > 1 launch path (torch.distributed.run, one rank
per GPU) smoke-tested with two ranks on the one GPU there is (--all-ranks-on-device0; gloo for the barrier so that two ranks can share
a device), kernel-timed region and the PCIe-inclusive e2e leg included."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_gpu_line(hip):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--launches-per-step", "8", "--e2e-seconds", "0.3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _line(r.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "configs[2]" in d["config"]["workload"] and d["config"]["launches_per_step"] == 8
    roof = d["roofline"]
    assert roof["bound"] == "valu" and 0 < roof["frac"] < 1 and 0 < roof["hbm_read_frac"] < roof["frac"] and roof["peak"] == 8000.0
    assert abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-3
    assert roof["traffic"] and roof["traffic_source"] and roof["valu"]["waves_per_launch"] == 129600 and 0 < roof["valu_frac"] < 1
    # value (wall clock over the job) can only be below what the kernel-only launch time allows
    px = 16 * 3840 * 2160
    assert d["value"] <= px / (roof["ms_per_launch"] * 1e-3) / 1e6 * 1.02
    e = d["e2e"]
    assert e["8k-v210"]["fps_total"] > 60 and len(e["8k-v210"]["fps_per_gpu"]) == 1 and e["4k-uyvy"]["fps_total"] > 60   # the north star's floor, PCIe included


def test_two_ranks_on_one_gpu_smoke(hip):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 1000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--launches-per-step", "8", "--e2e-seconds", "0.3",
                        "--dist-backend", "gloo", "--all-ranks-on-device0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert len(d["e2e"]["8k-v210"]["fps_per_gpu"]) == 2 and abs(sum(d["e2e"]["8k-v210"]["fps_per_gpu"]) - d["e2e"]["8k-v210"]["fps_total"]) < 0.2
    assert d["config"]["parallelism"].startswith("frames sharded over 2 GPU")
