"""Frame-sharded data parallelism for the 2Haff path: one process per GPU, `torch.distributed` over RCCL
(backend "nccl" on ROCm) — or gloo on a GPU-less host for the functional tests.

Frames are independent units: every rank runs the whole model replica on its contiguous block of
frames; there is NO collective on the data path. Collectives are used only for (a) the timing fence/max of the
benchmark and (b) gathering per-rank results (frame counts, mask checksums) at the end.
"""
import os
import time

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Rendezvous from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of rank; blocks differ by at most one item and cover 0..n_items exactly."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def fence(device=None):
    """barrier + device sync on both sides of a timed region (bench.py contract)."""
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)


def max_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_steps(step_fn, steps, device=None):
    """EXACTLY `steps` calls bracketed by fence(); returns the max-over-ranks wall time in seconds."""
    fence(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    fence(device)
    return max_over_ranks(time.perf_counter() - t0, device)


def evaluate_sharded(evaluate_fn, n_frames, device=None):
    """Run `evaluate_fn(lo, hi)` on this rank's block of frames and gather (lo, hi, result) from every rank.
    `result` must be picklable (e.g. per-frame mask checksums); returns the list ordered by rank on every rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_bounds(n_frames, rank, world)
    local = (lo, hi, evaluate_fn(lo, hi) if hi > lo else None)
    if world == 1:
        return [local]
    out = [None] * world
    dist.all_gather_object(out, local)
    return out
