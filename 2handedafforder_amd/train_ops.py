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


def adamw_step(state, grad, lr, betas=(0.9, 0.95), eps=1e-8, wd=0.0, gscale=1.0, param_lp=None, gscale_dev=None):
    """One fused AdamW update (torch.optim.AdamW semantics); optionally refreshes a bf16 copy of the parameter.
    gscale_dev: device fp32 scalar multiplied into gscale inside the kernel (clip_coef_device: no host read of the norm)."""
    lib = load_library()
    state.step += 1
    grad = grad.contiguous()
    lp_ptr, lp_dt = 0, -1
    if param_lp is not None and param_lp.dtype == torch.bfloat16:
        lp_ptr, lp_dt = param_lp.data_ptr(), 0
    if gscale_dev is not None:
        assert gscale_dev.dtype == torch.float32 and gscale_dev.numel() == 1
        check(lib.haff_adamw_step_dev(state.master.data_ptr(), state.m.data_ptr(), state.v.data_ptr(), grad.data_ptr(), lp_ptr,
                                      state.master.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps), float(wd),
                                      state.step, float(gscale), gscale_dev.data_ptr(), _dt(grad), lp_dt, _s()), "haff_adamw_step_dev")
    else:
        check(lib.haff_adamw_step(state.master.data_ptr(), state.m.data_ptr(), state.v.data_ptr(), grad.data_ptr(), lp_ptr,
                                  state.master.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps), float(wd),
                                  state.step, float(gscale), _dt(grad), lp_dt, _s()), "haff_adamw_step")
    if param_lp is not None and param_lp.dtype == torch.float32:
        param_lp.copy_(state.master)  # fp32 parameters alias the master values (plain copy)


def grad_norm(grads):
    """Global L2 norm of a list of gradient tensors (device scalar)."""
    lib = load_library()
    from . import autograd as A
    acc = torch.zeros((1,), dtype=torch.float32, device=grads[0].device)
    if not A.ORDERED_REDUCTIONS:
        for g in grads:
            g = g.contiguous()
            check(lib.haff_sumsq(g.data_ptr(), acc.data_ptr(), g.numel(), _dt(g), _s()), "haff_sumsq")
        return acc.sqrt()
    # ordered: <= 1024 block partials per tensor, added in index order, tensors in list order (bitwise repeatable)
    import ctypes
    partials = torch.empty((1024,), dtype=torch.float32, device=grads[0].device)
    n_parts = ctypes.c_int(0)
    for g in grads:
        g = g.contiguous()
        check(lib.haff_sumsq_partials(g.data_ptr(), partials.data_ptr(), g.numel(), _dt(g), ctypes.byref(n_parts), _s()), "haff_sumsq_partials")
        A._reduce_partials(partials, acc, 1, n_parts.value, 1, 1, 0, True)
    return acc.sqrt()


def clip_coef_device(norm, max_norm=1.0):
    """min(1, max_norm / (norm + 1e-6)) of a device-scalar gradient norm, as a device fp32 scalar (torch.nn.utils.clip_grad_norm_'s
    coefficient; gradient clipping 1.0 of the reference's engine config, train_ds.py:381) — for adamw_step(gscale_dev=)."""
    return torch.clamp(max_norm / (norm.to(torch.float32) + 1e-6), max=1.0).reshape(1)


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


def _ready_rank(name):
    """Order in which the trainable tensors' gradients become final during backward (output side first):
    lm_head, then text_hidden_fcs and the mask decoders, then the LoRA pairs from the LAST Llama layer down to the first,
    embed_tokens last (its gradient is only complete once the whole backward has reached the input embeddings)."""
    import re
    if name.startswith("lm_head"):
        return (0, 0)
    if "embed_tokens" in name:
        return (3, 0)
    m = re.search(r"layers\.(\d+)\.", name)
    if m and "lora_" in name:
        return (2, -int(m.group(1)))
    return (1, 0)


class GradBucketReducer:
    """Gradient all-reduce of the trainable set, overlapped with backward (the reference's engine does
    the same with overlap_comm / reduce_scatter per micro-step, train_ds.py:372-379 — here ONCE per optimizer step).

    * parameters are assigned to buckets of ~bucket_bytes in the order their gradients become ready (_ready_rank), per
      dtype; every bucket owns ONE flat buffer in the gradients' own dtype and each p.grad is a VIEW into it: backward
      accumulates straight into the bucket (no cat, no up-cast, no copy back);
    * post-accumulate hooks count a bucket's arrivals; on the LAST micro-step of an accumulation window (begin(sync=True))
      the bucket's async all-reduce (RCCL over xGMI on the GPU node, gloo in the CPU tests) is launched once its last
      gradient has landed AND every lower-indexed bucket has been launched, while backward continues into the earlier layers;
    * collectives are issued in BUCKET-INDEX ORDER ONLY, on every rank, whatever its autograd graph reached (the reference's
      ZeRO-2 engine reduces fixed buckets in a fixed order, train_ds.py:372-393): a rank whose micro-batch has no [SEG] or no
      left hand never sees some hooks fire, and a hook-arrival order would make it issue its all-reduces in a different
      sequence than its peers — on RCCL a hang or silently mismatched reductions. A bucket that never completes on this
      rank simply holds back the ones behind it until finish();
    * finish() launches the rest in index order, waits, and turns sums into means.
    """

    def __init__(self, named_params, bucket_bytes=64 << 20):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        items = sorted(named_params, key=lambda kv: _ready_rank(kv[0]))
        self.buckets = []          # each: {"flat", "names", "pending", "n", "work"}
        self.bucket_of = {}
        open_bucket = {}           # ONE open bucket per dtype: bf16 matrices and fp32 vectors alternate in state-dict order,
        for name, p in items:      # and a bucket per dtype CHANGE would be a bucket (an all-reduce) per tensor
            nbytes = p.numel() * p.element_size()
            cur = open_bucket.get(p.dtype)
            if cur is None or (cur["bytes"] + nbytes > bucket_bytes and cur["bytes"] > 0):
                cur = {"dtype": p.dtype, "bytes": 0, "params": [], "names": []}
                open_bucket[p.dtype] = cur
                self.buckets.append(cur)
            cur["params"].append(p)
            cur["names"].append(name)
            cur["bytes"] += nbytes
        for bi, b in enumerate(self.buckets):
            n = sum(p.numel() for p in b["params"])
            b["flat"] = torch.zeros((n,), dtype=b["dtype"], device=b["params"][0].device)
            off = 0
            for p in b["params"]:
                p.grad = b["flat"][off:off + p.numel()].view_as(p)
                off += p.numel()
                self.bucket_of[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._on_grad)
            b["n"], b["pending"], b["work"] = len(b["params"]), len(b["params"]), None
        self.sync = False
        self.next_launch = 0       # lowest bucket index whose all-reduce has not been issued in this window
        self.launch_order = []     # bucket indices in the order their all-reduce was issued (always 0, 1, 2, ...; tests / tracing)

    def zero(self):
        for b in self.buckets:
            b["flat"].zero_()

    def begin(self, sync):
        """Call before each backward: sync=True on the micro-step whose gradients complete the accumulation window."""
        self.sync = bool(sync) and self.world > 1
        self.launch_order = []
        self.next_launch = 0
        for b in self.buckets:
            b["pending"], b["work"] = b["n"], None

    def _launch(self, bi):
        assert bi == self.next_launch, "all-reduces leave in bucket-index order on every rank"
        b = self.buckets[bi]
        b["work"] = self.dist.all_reduce(b["flat"], op=self.dist.ReduceOp.SUM, async_op=True)
        self.launch_order.append(bi)
        self.next_launch = bi + 1

    def _drain_ready(self):
        while self.next_launch < len(self.buckets) and self.buckets[self.next_launch]["pending"] <= 0:
            self._launch(self.next_launch)

    def _on_grad(self, p):
        if not self.sync:
            return
        b = self.buckets[self.bucket_of[id(p)]]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._drain_ready()

    def finish(self):
        """After the last backward of the window: every bucket reduced and averaged in place (p.grad views see it)."""
        if self.world == 1:
            return
        while self.next_launch < len(self.buckets):   # buckets the loss did not (fully) reach on this rank, and all behind them
            self._launch(self.next_launch)
        for b in self.buckets:
            b["work"].wait()
            b["flat"].div_(self.world)
        self.sync = False

    def grads(self):
        return [b["flat"] for b in self.buckets]


class _StateView:
    """One parameter's slice of a BucketAdamW bucket: the fields train_ds.py's checkpoint code reads and writes per key."""

    def __init__(self, opt, master, m, v):
        self._opt, self.master, self.m, self.v = opt, master, m, v

    @property
    def step(self):
        return self._opt.step_count

    @step.setter
    def step(self, value):
        self._opt.step_count = int(value)


class BucketAdamW:
    """AdamW over the flat gradient buckets of a GradBucketReducer: ONE fused launch per bucket (<= 16 for the 7B trainable set)
    instead of one per tensor (237). The reference's engine steps a flattened fp32 partition the same way (DeepSpeed's bf16
    optimizer behind train_ds.py:344-393: AdamW lr / betas (0.9, 0.95) / wd 0, fp32 master weights, gradient clipping 1.0).

    * per bucket: fp32 master / m / v as flat buffers in the bucket's parameter order;
    * bf16 parameters are RE-POINTED (p.data) at views of one flat bf16 buffer per bucket, which the kernel refreshes from the
      master values; fp32 parameters alias their master slice directly. Build it before anything caches parameter storage;
    * `states[name]` exposes each parameter's {master, m, v, step} as views (checkpoint / resume code keeps working per key)."""

    def __init__(self, reducer, named_params):
        names = {id(p): k for k, p in named_params}
        self.step_count = 0
        self.buckets, self.states = [], {}
        for b in reducer.buckets:
            n = b["flat"].numel()
            dev = b["flat"].device
            master = torch.empty((n,), dtype=torch.float32, device=dev)
            off = 0
            for p in b["params"]:
                master[off:off + p.numel()] = p.detach().reshape(-1).to(torch.float32)
                off += p.numel()
            m, v = torch.zeros_like(master), torch.zeros_like(master)
            lp = None if b["dtype"] == torch.float32 else master.to(b["dtype"])
            off = 0
            for p in b["params"]:
                k = p.numel()
                src = master if lp is None else lp
                p.data = src[off:off + k].view(p.shape)
                self.states[names[id(p)]] = _StateView(self, master[off:off + k].view(p.shape), m[off:off + k].view(p.shape),
                                                       v[off:off + k].view(p.shape))
                off += k
            self.buckets.append({"master": master, "m": m, "v": v, "lp": lp, "grad": b["flat"]})

    def refresh_lp(self):
        """After master values were written from outside (resume): the bf16 parameter copies follow."""
        for b in self.buckets:
            if b["lp"] is not None:
                b["lp"].copy_(b["master"])

    def step(self, lr, betas=(0.9, 0.95), eps=1e-8, wd=0.0, gscale=1.0, gscale_dev=None):
        lib = load_library()
        self.step_count += 1
        for b in self.buckets:
            g = b["grad"]
            lp_ptr, lp_dt = (b["lp"].data_ptr(), 0) if b["lp"] is not None else (0, -1)
            args = (b["master"].data_ptr(), b["m"].data_ptr(), b["v"].data_ptr(), g.data_ptr(), lp_ptr, g.numel(), float(lr),
                    float(betas[0]), float(betas[1]), float(eps), float(wd), self.step_count, float(gscale))
            if gscale_dev is not None:
                check(lib.haff_adamw_step_dev(*args, gscale_dev.data_ptr(), _dt(g), lp_dt, _s()), "haff_adamw_step_dev")
            else:
                check(lib.haff_adamw_step(*args, _dt(g), lp_dt, _s()), "haff_adamw_step")
