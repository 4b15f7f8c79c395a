"""Optimiser-side helpers of the fine-tune loop (reference: DeepSpeed engine built at train_ds.py:344-393 —
AdamW lr/betas(0.9,0.95)/wd 0, WarmupDecayLR, gradient clipping 1.0, bf16 with fp32 master weights).
Fused HIP kernels for the update and the gradient norm; RCCL all-reduce of the trainable gradients (DDP
semantics) through torch.distributed on flat buckets."""
import math

import torch

from .autograd import _dt, _s
from .lib import check, load_library


class AdamWState:
    """fp32 master copy + first/second moments of one parameter tensor."""

    def __init__(self, param):
        self.master = param.detach().to(torch.float32).clone().contiguous()
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        self.step = 0


def adamw_step(state, grad, lr, betas=(0.9, 0.95), eps=1e-8, wd=0.0, gscale=1.0, param_lp=None):
    """One fused AdamW update (torch.optim.AdamW semantics); optionally refreshes a bf16 copy of the parameter."""
    lib = load_library()
    state.step += 1
    grad = grad.contiguous()
    lp_ptr, lp_dt = 0, -1
    if param_lp is not None and param_lp.dtype == torch.bfloat16:
        lp_ptr, lp_dt = param_lp.data_ptr(), 0
    check(lib.haff_adamw_step(state.master.data_ptr(), state.m.data_ptr(), state.v.data_ptr(), grad.data_ptr(), lp_ptr,
                              state.master.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps), float(wd),
                              state.step, float(gscale), _dt(grad), lp_dt, _s()), "haff_adamw_step")
    if param_lp is not None and param_lp.dtype == torch.float32:
        param_lp.copy_(state.master)  # fp32 parameters alias the master values (plain copy)


def grad_norm(grads):
    """Global L2 norm of a list of gradient tensors (device scalar)."""
    lib = load_library()
    acc = torch.zeros((1,), dtype=torch.float32, device=grads[0].device)
    for g in grads:
        g = g.contiguous()
        check(lib.haff_sumsq(g.data_ptr(), acc.data_ptr(), g.numel(), _dt(g), _s()), "haff_sumsq")
    return acc.sqrt()


def warmup_decay_lr(step, total_steps, base_lr, warmup_steps=100, warmup_min_lr=0.0):
    """DeepSpeed WarmupDecayLR (train_ds.py:361-369): linear warm-up to base_lr, then linear decay to 0."""
    if step < warmup_steps:
        return warmup_min_lr + (base_lr - warmup_min_lr) * step / max(warmup_steps, 1)
    return base_lr * max(0.0, (total_steps - step) / max(total_steps - warmup_steps, 1))


def allreduce_mean_(tensors, bucket_bytes=64 << 20):
    """Average gradient tensors over ranks with bucketed flat all-reduces (RCCL when the tensors live in HBM;
    gloo on CPU in the tests). Buckets of ~64 MB: xGMI rings are per-link bound, few large collectives win."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1).to(torch.float32) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
            off += n
        bucket, size = [], 0
    for t in tensors:
        bucket.append(t)
        size += t.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()
