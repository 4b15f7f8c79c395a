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


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from haff import train_ops as T
    from haff.train_ds import AverageMeter, all_reduce_meters
    hdist.init_from_env("gloo")
    g = torch.Generator().manual_seed(100 + rank)
    # per-rank "gradients" of a trainable set with mixed dtypes and sizes (several buckets at 1 KB buckets)
    grads = [torch.randn((257, 3), generator=g), torch.randn((1000,), generator=g).to(torch.bfloat16), torch.randn((5,), generator=g)]
    T.allreduce_mean_(grads, bucket_bytes=1024)
    meters = [AverageMeter("Loss"), AverageMeter("Time")]
    meters[0].update(1.0 + rank, 2)
    meters[1].update(0.5, 1)
    all_reduce_meters(meters, "cpu")
    if rank == 0:
        q.put(([t.float() for t in grads], [(m.sum, m.count) for m in meters]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_mean_of_rank_gradients():
    """DDP semantics of the fine-tune loop: averaged gradients == mean over ranks (gloo stand-in for RCCL)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    grads, meters = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per_rank = []
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        per_rank.append([torch.randn((257, 3), generator=g), torch.randn((1000,), generator=g).to(torch.bfloat16).float(),
                         torch.randn((5,), generator=g)])
    for i, got in enumerate(grads):
        ref = (per_rank[0][i] + per_rank[1][i]) / 2
        tol = 1e-6 if i != 1 else 2e-2
        assert (got - ref).abs().max().item() <= tol
    assert meters[0] == (2.0 + 4.0, 4.0) and meters[1] == (1.0, 2.0)
