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
