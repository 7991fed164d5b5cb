"""Frame sharding across GPUs: frames are independent, so N ranks encode disjoint frames
with NO data-path collective.  Assignment and in-order delivery follow the reference's multi-GPU JPEG module
(sequence numbers handed out at push, round-robin over devices, reorder on pop:
src/video_compress/gpujpeg.cpp:643-676,688-722).  torch.distributed (RCCL on GPUs, gloo in CPU tests) is used
only for the timing barrier and the max-over-ranks reduction."""
from __future__ import annotations

import time
from typing import Callable, Iterable, List, Sequence, Tuple


def self_launch_command(gpus: int, environ, argv: Sequence[str], python: str, port: int | None = None) -> List[str] | None:
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run (no WORLD_SIZE / RANK in the environment) must still run N ranks:
    returns the command that re-executes `argv` (script + its arguments) as one rank per GPU under torch.distributed.run on this node, or
    None when no re-launch is needed (N = 1, or the launcher's environment is already there -- then WORLD_SIZE must equal N, checked by
    the caller).  One process per GPU is the reference's one-worker-per-device scheme (gpujpeg.cpp:446-466) at process granularity."""
    if gpus <= 1:
        return None
    if "WORLD_SIZE" in environ or "RANK" in environ or "LOCAL_RANK" in environ:
        return None
    if port is None:
        port = free_port()
    return [python, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), *argv]


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Sequence numbers handled by `rank`: seq % world == rank (round-robin, gpujpeg.cpp:661-675)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_frames, world))


def reorder(parts: Iterable[Sequence[Tuple[int, object]]]) -> List[object]:
    """Merge per-rank [(seq, payload), ...] lists back into sequence order (gpujpeg.cpp:688-722);
    raises if a frame is missing or was encoded twice."""
    merged = {}
    for part in parts:
        for seq, payload in part:
            if seq in merged:
                raise ValueError(f"frame {seq} encoded twice")
            merged[seq] = payload
    n = len(merged)
    if sorted(merged) != list(range(n)):
        raise ValueError("missing frames: " + str(sorted(set(range(max(merged, default=-1) + 1)) - set(merged))))
    return [merged[i] for i in range(n)]


def timed_steps(step: Callable[[], None], steps: int, sync: Callable[[], None], dist=None, device=None) -> float:
    """Bench contract: barrier + device sync on both sides of exactly `steps` steps; returns the MAX wall time
    over ranks (seconds)."""
    import torch

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall


def gather_rates(local: Sequence[float], dist=None, device=None) -> List[List[float]]:
    """Per-rank rate vectors (fps, GB/s ...) of a leg every rank ran at the same time -> [[rank0...], [rank1...], ...] on every
    rank.  The only thing ranks ever exchange besides the timing barrier: a handful of numbers, never frame data."""
    import torch
    if dist is None:
        return [list(map(float, local))]
    t = torch.tensor(list(local), dtype=torch.float64, device=device or "cpu")
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [[float(x) for x in p.tolist()] for p in parts]


def agree_max(value: int, dist=None, device=None) -> int:
    """The same integer on every rank (the maximum): ranks calibrate their step size locally and must then run the SAME step."""
    import torch
    if dist is None:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())
