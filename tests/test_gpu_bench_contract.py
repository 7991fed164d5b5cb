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
                        "--configs-seconds", "0.25", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _line(r.stdout)
    # the additive keys of round 5: an in-run oracle check of the timed kernel's output, every other BASELINE configuration
    # under the same clock, one-frame-in-flight latency, the VALU rate against the measured issue peak
    assert d["parity_check"] == dict(d["parity_check"], frames=1, bytes_compared=3840 * 2160, bytes_differing=0)
    assert set(d["configs"]) == {"1080p-rgb-dxt1", "4k-uyvy-jpeg420", "4k-uyvy-jpeg-encode", "8k-v210"}
    for name, c in d["configs"].items():
        assert c["ms_per_launch"] > 0 and c["launches_timed"] >= 4 and 0.05 < c["frac"] < 1, (name, c)
        assert abs(c["algorithmic_bytes_per_launch"] / (c["ms_per_launch"] * 1e-3) / 1e9 / 8000.0 - c["frac"]) < 2e-3
    assert d["configs"]["1080p-rgb-dxt1"]["parity_check"]["bytes_differing"] == 0 and d["configs"]["8k-v210"]["parity_check"]["bytes_differing"] == 0
    assert d["configs"]["8k-v210"]["parity_check"]["bytes_compared"] == 7680 * 4320
    # round 6: the two JPEG configurations carry an oracle check of what the TIMED launches produced too -- the front end's
    # coefficients, and the encoder's stream entropy-decoded back to the coefficients it codes -- and the e2e ceiling is taken beside the leg it bounds
    pj, pe = d["configs"]["4k-uyvy-jpeg420"]["parity_check"], d["configs"]["4k-uyvy-jpeg-encode"]["parity_check"]
    assert pj["coefficients_differing"] == 0 and pj["coefficients_compared"] == 240 * 135 * 6 * 64
    assert pe["coefficients_differing"] == 0 and pe["coefficients_compared"] == 240 * 135 * 6 * 64 and pe["ends_with_eoi"]
    assert len(pe["lens"]) == 8 and pe["stream_bytes"] == pe["lens"][0] and all(100_000 < n < 3840 * 2160 for n in pe["lens"])
    for wl in ("8k-uyvy", "8k-v210", "4k-uyvy"):
        runs = d["e2e"][wl]["copy_only_runs"]
        assert len(runs) == 3 and max(runs) == d["e2e"][wl]["copy_only_fps_per_gpu"][0]
        assert 0.5 < d["e2e"][wl]["frac_of_copy_only"] <= 1.10, (wl, d["e2e"][wl])    # (0.3 s legs here: the driver's default 2 s legs are held to 1.02, profiles/)
    lat = d["e2e"]["latency_ms_depth1"]
    assert set(lat) == {"8k-uyvy", "8k-v210", "4k-uyvy"}
    for name, v in lat.items():
        assert v["ms"] >= max(v["h2d_ms"], v["d2h_ms"]) and v["ms"] < 16.7 and v["kernel_ms"] < v["h2d_ms"], (name, v)   # one frame period at 60 fps is 16.7 ms
        assert 0 < v["bands4_ms"] < 16.7 and 0 < v["bands8_ms"] < 16.7
    assert lat["8k-uyvy"]["bands4_ms"] < lat["8k-uyvy"]["ms"] and lat["8k-v210"]["bands4_ms"] < lat["8k-v210"]["ms"]    # row bands overlap the copies of ONE frame
    assert 0 < d["roofline"]["valu_frac"] < d["roofline"]["valu"]["frac_of_measured_peak"] < 1.05
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
    assert e["8k-uyvy"]["fps_total"] > 60 and e["8k-uyvy"]["bytes_in_per_frame"] == 7680 * 4320 * 2                       # the target's literal configuration
    assert "dist" not in d["config"]


def test_rccl_group_at_world_size_one(hip):
    """the `nccl` (= RCCL) branch of bench.py -- init_process_group with a device id, barrier, all_reduce(MAX) of the step size
    and of the wall time, all_gather of the e2e rates, all on cuda tensors -- runs here at world size 1, so that the driver's 8-GPU run is
    not the first time any of it executes; and the CPU baseline is in the line whenever a group is up."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--launches-per-step", "8", "--e2e-seconds", "0.3",
                        "--force-dist"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["dist"]["backend"].startswith("nccl") and d["config"]["dist"]["world"] == 1
    assert "cuda" in d["config"]["dist"]["collectives"]
    assert d["value"] > 0 and len(d["e2e"]["8k-uyvy"]["fps_per_gpu"]) == 1
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"


def test_shard_collectives_on_cuda_tensors(hip):
    """ultragrid_amd/shard.py's three collectives straight on an RCCL group of one rank (cuda tensors)."""
    code = ("import os, torch, torch.distributed as dist; from ultragrid_amd import shard; "
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29500 + os.getpid() % 2000), RANK='0', WORLD_SIZE='1'); "
            "torch.cuda.set_device(0); dist.init_process_group('nccl', device_id=torch.device('cuda', 0)); "
            "assert shard.agree_max(7, dist, 'cuda') == 7; assert shard.gather_rates([1.5, 2.5], dist, 'cuda') == [[1.5, 2.5]]; "
            "w = shard.timed_steps(lambda: None, 3, torch.cuda.synchronize, dist, device='cuda'); assert 0 <= w < 5; "
            "dist.destroy_process_group(); print('RCCL-OK')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_two_ranks_on_one_gpu_smoke(hip):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 1000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--launches-per-step", "8", "--e2e-seconds", "0.3",
                        "--dist-backend", "gloo", "--all-ranks-on-device0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["cpu_baseline"]["value"] > 0   # rank 0 times the CPU oracle at every world size
    assert d["config"]["dist"]["world"] == 2 and d["config"]["dist"]["backend"] == "gloo"
    assert len(d["e2e"]["8k-v210"]["fps_per_gpu"]) == 2 and abs(sum(d["e2e"]["8k-v210"]["fps_per_gpu"]) - d["e2e"]["8k-v210"]["fps_total"]) < 0.2
    assert d["config"]["parallelism"].startswith("frames sharded over 2 GPU")


def test_gpus_flag_without_a_launcher_starts_its_own_ranks(hip):
    """`python3 bench.py --gpus 2` as the driver might invoke it -- no torch.distributed.run, no WORLD_SIZE -- launches its two
    ranks itself and still prints exactly one JSON line with n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--launches-per-step", "8", "--e2e-seconds", "0.3",
                        "--dist-backend", "gloo", "--all-ranks-on-device0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["dist"]["world"] == 2 and d["value"] > 0


def test_whole_jpeg_encoder_workload(hip):
    """`--workload 4k-uyvy-jpeg-encode`: the whole encoder (fused kernel + stream assembly) as a bench line of its own, so that its rate is
    reproducible with the driver's tool and not only with tools/bench_jpeg_batch.py."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "4k-uyvy-jpeg-encode", "--steps", "2", "--warmup", "1", "--launches-per-step", "4",
                        "--no-e2e"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _line(r.stdout)
    assert REQUIRED <= set(d) and "JPEG" in d["metric"] and d["config"]["frames_per_launch"] == 8
    roof = d["roofline"]
    assert roof["bound"] == "valu" and 5 < roof["us_per_frame"] < 60 and 0 < roof["frac"] < 1
    assert 1_000_000 < d["config"]["frame_bytes_out"] < 3_000_000            # a 4K q75 4:2:0 stream of video noise: ~1.5 MB
    assert roof["algorithmic_bytes_per_launch"] == 8 * 3840 * 2160 * 2 + 8 * d["config"]["frame_bytes_out"] or abs(roof["algorithmic_bytes_per_launch"] - (8 * 3840 * 2160 * 2 + 8 * d["config"]["frame_bytes_out"])) < 8
