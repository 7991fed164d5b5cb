"""N > 1 path on CPU: world_size-2 gloo processes shard frames exactly as bench.py / the module would across
GPUs (no data-path collective), results are merged in sequence order and equal the single-process result.
The per-frame work here is the CPU oracle (this is a test of the sharding logic, not of the HIP path)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ultragrid_amd import shard, synth  # noqa: E402

W, H, N_FRAMES = 64, 16, 7


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    mine = shard.frames_for_rank(N_FRAMES, rank, world)
    out = []

    def step():
        out.clear()
        for seq in mine:
            out.append((seq, po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, synth.s1_random("UYVY", W, H, salt=seq), W, H).tobytes()))

    wall = shard.timed_steps(step, 2, lambda: None, dist)
    # what bench.py exchanges besides the barrier: the agreed step size and the per-rank rates of the e2e leg
    agreed = shard.agree_max(100 + 7 * rank, dist)
    rates = shard.gather_rates([10.0 * (rank + 1), 0.5 * (rank + 1)], dist)
    assert agreed == 100 + 7 * (world - 1)
    assert rates == [[10.0 * (r + 1), 0.5 * (r + 1)] for r in range(world)]
    gathered = [None] * world
    dist.all_gather_object(gathered, out)   # test-only gather; the product never exchanges frame data between ranks
    if rank == 0:
        q.put((wall, gathered))
    dist.destroy_process_group()


def test_rate_helpers_without_a_process_group():
    assert shard.gather_rates([3.0, 1.5]) == [[3.0, 1.5]] and shard.agree_max(42) == 42


def test_round_robin_assignment():
    assert shard.frames_for_rank(7, 0, 2) == [0, 2, 4, 6] and shard.frames_for_rank(7, 1, 2) == [1, 3, 5]
    allf = sorted(sum((shard.frames_for_rank(60, r, 8) for r in range(8)), []))
    assert allf == list(range(60))
    with pytest.raises(ValueError):
        shard.frames_for_rank(4, 2, 2)
    with pytest.raises(ValueError):
        shard.reorder([[(0, "a")], [(0, "b")]])
    with pytest.raises(ValueError):
        shard.reorder([[(0, "a")], [(2, "b")]])


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_sharding_matches_single_process(po, world):
    """world_size 2 (the contract's CPU case) and 4 (7 frames over 4 ranks: uneven shares)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7 * world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    wall, gathered = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    merged = shard.reorder(gathered)
    assert len(merged) == N_FRAMES and wall > 0
    for seq, blob in enumerate(merged):
        want = po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, synth.s1_random("UYVY", W, H, salt=seq), W, H).tobytes()
        assert blob == want, seq


def test_bench_self_launch_decision():
    """`python bench.py --gpus N` without a launcher re-executes itself as N ranks under torch.distributed.run on
    127.0.0.1; under a launcher (WORLD_SIZE / RANK present) and at N = 1 it runs in place."""
    argv = ["/x/bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = shard.self_launch_command(8, {}, argv, "/usr/bin/python3", 29512)
    assert cmd[:3] == ["/usr/bin/python3", "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29512" and "--nnodes=1" in cmd and cmd[-len(argv):] == argv
    assert shard.self_launch_command(1, {}, argv, "python") is None
    assert shard.self_launch_command(8, {"WORLD_SIZE": "8", "RANK": "3"}, argv, "python") is None
    assert shard.self_launch_command(2, {"LOCAL_RANK": "0"}, argv, "python") is None
    p = shard.self_launch_command(2, {"PATH": "/bin"}, argv, "python")   # port picked here: a free one
    assert 1024 < int(p[p.index("--master-port") + 1]) < 65536


def test_bench_self_launch_runs_the_ranks(tmp_path):
    """The re-exec itself, on CPU: bench.py --gpus 2 with no launcher starts two ranks; each gets as far as the GPU check (there is no GPU
    here), i.e. both ran main() past the launch decision with WORLD_SIZE=2 -- and the parent returns the job's non-zero exit code."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the re-exec; the GPU form is tests/test_gpu_bench_contract.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600, cwd=str(tmp_path), env=env)
    assert r.returncode != 0
    out = r.stdout + r.stderr
    # both ranks say so -- unless the launcher's agent saw the first one fail and sent SIGTERM to the second while that was still
    # importing torch (a loaded box: pytest -n 4); the agent's log line then names the second process it had started
    n = out.count("bench.py needs a GPU")
    assert n >= 2 or (n == 1 and "closing signal SIGTERM" in out), out[-3000:]
