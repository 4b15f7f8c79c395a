"""N>1 path on CPU: world_size-2 gloo processes exercise the frame sharding, the timing fence/max and the result
gather that bench.py and multi-GPU inference use (RCCL replaces gloo on the GPU node; the logic is identical)."""
import os
import socket

import torch
import torch.multiprocessing as mp

import haff  # noqa: F401
from haff import dist as hdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _checksum_eval(frames):
    def fn(lo, hi):
        # stand-in for model.evaluate on frames[lo:hi]: a per-frame deterministic checksum
        return [int(frames[i].to(torch.int64).sum()) for i in range(lo, hi)]
    return fn


def _worker(rank, world, port, n_frames, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = hdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (n_frames, 8, 8, 3), generator=g, dtype=torch.uint8)
    gathered = hdist.evaluate_sharded(_checksum_eval(frames), n_frames)
    calls = []
    t = hdist.timed_steps(lambda: calls.append(1) or torch.ones(10).sum(), 3)
    slow = hdist.max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((gathered, len(calls), t, slow))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [hdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharded_evaluate_matches_single_process():
    n_frames, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, n_calls, t, slow = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (n_frames, 8, 8, 3), generator=g, dtype=torch.uint8)
    single = _checksum_eval(frames)(0, n_frames)
    merged = [c for lo, hi, res in gathered for c in res]
    assert [(lo, hi) for lo, hi, _ in gathered] == [(0, 4), (4, 7)]
    assert merged == single
    assert n_calls == 3 and t > 0 and slow == 2.0
