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
        q.put(([t.float().numpy().copy() for t in grads], [(m.sum, m.count) for m in meters]))  # numpy: no fd passing
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
        got = torch.from_numpy(got)
        ref = (per_rank[0][i] + per_rank[1][i]) / 2
        tol = 1e-6 if i != 1 else 2e-2
        assert (got - ref).abs().max().item() <= tol
    assert meters[0] == (2.0 + 4.0, 4.0) and meters[1] == (1.0, 2.0)


# ---- DDP semantics on the MODEL's gradients: two ranks on half-batches + overlapped bucketed ----
# ---- all-reduce == one process on the concatenated batch. CPU stand-in for the HIP model: the oracle under autograd.   ----
def _trainable_leaves(cfg, sd, seed=3, r=4):
    """The reference's trainable set (train_ds.py:192-244) as fp32 leaves: LoRA A/B on q_proj / v_proj of every Llama layer,
    embed_tokens, lm_head, text_hidden_fcs, both mask decoders."""
    g = torch.Generator().manual_seed(seed)
    osd = {k: v.clone() for k, v in sd.items()}
    named, lora = [], {}
    for k in sorted(sd):
        if k in ("lm_head.weight", "model.embed_tokens.weight") or "text_hidden_fcs" in k or "mask_decoder_" in k:
            osd[k] = osd[k].clone().requires_grad_(True)
            named.append((k, osd[k]))
    H = cfg.llm.hidden
    for i in range(cfg.llm.layers):
        for n in ("q_proj", "v_proj"):
            base = f"model.layers.{i}.self_attn.{n}"
            lora[base + ".lora_A"] = (torch.randn((r, H), generator=g) * 0.05).requires_grad_(True)
            lora[base + ".lora_B"] = (torch.randn((H, r), generator=g) * 0.05).requires_grad_(True)
            named += [(base + ".lora_A", lora[base + ".lora_A"]), (base + ".lora_B", lora[base + ".lora_B"])]
    return osd, lora, named


def _half(batch, lo, hi):
    out = {}
    for k, v in batch.items():
        if k == "offset":
            out[k] = torch.arange(hi - lo + 1)
        elif torch.is_tensor(v):
            out[k] = v[lo:hi]
        elif isinstance(v, list):
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def _model_grad_worker(rank, world, port, q, divergent=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from haff import config as hcfg, train_ops as T, weights as hw
    from oracle import lisa_oracle as O
    from test_train_gpu import make_batch
    hdist.init_from_env("gloo")
    cfg = hcfg.tiny()
    sd = hw.make_state_dict(cfg, 21)
    osd, lora, named = _trainable_leaves(cfg, sd)
    reducer = T.GradBucketReducer(named, bucket_bytes=64 << 10)
    fired_in_backward = []
    launch = reducer._launch
    reducer._launch = lambda bi: (fired_in_backward.append(bi), launch(bi))[1]
    batch = make_batch(cfg, b=2, hw=(40, 36))
    reducer.zero()
    # two micro-steps (gradient accumulation): only the second one may talk to the other rank
    for micro in range(2):
        reducer.begin(sync=micro == 1)
        out = O.lisa_model_forward(osd, cfg, _half(batch, rank, rank + 1), lora=lora, lora_alpha=8.0)
        # divergent: rank 1's graph is the one a micro-batch without [SEG] / without masks produces — the language-model loss
        # only, so text_hidden_fcs and both mask decoders never see a gradient (their hooks never fire on this rank)
        loss = out["ce_loss"] if (divergent and rank == 1) else out["loss"]
        (loss * 0.5).backward()
        if micro == 0:
            assert fired_in_backward == []
    n_hook = len(fired_in_backward)
    reducer.finish()
    names = [b["names"] for b in reducer.buckets]
    if divergent:
        q.put((rank, {k: p.grad.detach().numpy().copy() for k, p in named}, n_hook, list(reducer.launch_order), names))
    elif rank == 0:
        q.put(({k: p.grad.detach().numpy().copy() for k, p in named}, n_hook, list(reducer.launch_order), names))  # numpy: no fd passing
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_model_gradients_equal_single_process_on_the_concatenated_batch():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from haff import config as hcfg, weights as hw
    from oracle import lisa_oracle as O
    from test_train_gpu import make_batch
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, n_hook, order, names = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single process, whole batch, the same two accumulation micro-steps
    cfg = hcfg.tiny()
    sd = hw.make_state_dict(cfg, 21)
    osd, lora, named = _trainable_leaves(cfg, sd)
    batch = make_batch(cfg, b=2, hw=(40, 36))
    for _ in range(2):
        out = O.lisa_model_forward(osd, cfg, batch, lora=lora, lora_alpha=8.0)
        (out["loss"] * 0.5).backward()
    worst = 0.0
    for k, p in named:
        gk = torch.from_numpy(got[k])
        if p.grad is None:
            assert float(gk.abs().max()) == 0.0, k
            continue
        scale = p.grad.abs().max().item()
        err = (gk - p.grad).abs().max().item()
        worst = max(worst, err / (scale + 1e-12) if scale > 1e-7 else 0.0)
        assert err <= 2e-4 * scale + 1e-7, (k, err, scale)
    # overlap: buckets were launched from inside backward, output side before input side
    assert n_hook >= 2 and len(order) == len(names)
    flat_names = [n for bi in order for n in names[bi]]
    assert flat_names.index("lm_head.weight") < flat_names.index("model.layers.0.self_attn.q_proj.lora_A")
    assert flat_names.index("model.layers.1.self_attn.q_proj.lora_A") < flat_names.index("model.layers.0.self_attn.q_proj.lora_A")


def test_two_rank_reducer_with_rank_divergent_graphs_issues_the_same_collective_sequence():
    """A rank whose micro-batch reaches a different parameter set (no [SEG], a missing hand: here rank 1 back-propagates the
    language-model loss only) must issue its all-reduces in the same order as its peers: bucket-index order, whatever the hooks'
    arrival order was (the reference's ZeRO-2 engine reduces fixed buckets in a fixed order, train_ds.py:372-393). Gradients ==
    the mean of the two ranks' gradients computed in one process."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from haff import config as hcfg, weights as hw
    from oracle import lisa_oracle as O
    from test_train_gpu import make_batch
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_grad_worker, args=(r, world, port, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, got, n_hook, order, names = q.get(timeout=600)
        res[rank] = (got, n_hook, order, names)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n_buckets = len(res[0][3])
    assert n_buckets >= 4
    assert res[0][2] == list(range(n_buckets)) and res[1][2] == list(range(n_buckets))   # same sequence on both ranks
    assert res[0][3] == res[1][3]
    # rank 0 overlapped (its graph reaches everything); rank 1 could launch from hooks only up to the first bucket that holds a
    # tensor its graph never reached (the decoders / fcs sit right behind lm_head): the rest left in finish()
    assert res[0][1] >= 2 and res[1][1] < res[0][1]
    cfg = hcfg.tiny()
    sd = hw.make_state_dict(cfg, 21)
    batch = make_batch(cfg, b=2, hw=(40, 36))
    ref = {}
    for rank in range(world):
        osd, lora, named = _trainable_leaves(cfg, sd)
        for _ in range(2):
            out = O.lisa_model_forward(osd, cfg, _half(batch, rank, rank + 1), lora=lora, lora_alpha=8.0)
            ((out["ce_loss"] if rank == 1 else out["loss"]) * 0.5).backward()
        for k, p in named:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            ref[k] = ref.get(k, 0) + g / world
    untouched = 0
    for rank in range(world):
        for k, want in ref.items():
            gk = torch.from_numpy(res[rank][0][k])
            scale = want.abs().max().item()
            assert (gk - want).abs().max().item() <= 2e-4 * scale + 1e-7, (rank, k)
    osd, lora, named = _trainable_leaves(cfg, sd)
    out = O.lisa_model_forward(osd, cfg, _half(batch, 1, 2), lora=lora, lora_alpha=8.0)
    out["ce_loss"].backward()
    untouched = sum(1 for k, p in named if p.grad is None)
    assert untouched >= 50, untouched     # the divergence is real: the decoders' and text_hidden_fcs' tensors saw no gradient on rank 1


def _bench_stub_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import contextlib
    import io
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--stub-step-ms", str(20 + 30 * rank), "--batch", "5"])
    q.put((rank, buf.getvalue()))


def test_bench_multi_rank_path_dry_run_with_stub_model():
    """bench.py --gpus 2 (rendezvous from the torchrun environment, fence, EXACTLY K timed steps, max over ranks, one JSON
    line on rank 0, barrier + teardown) with the model replaced by a sleep: rank 1 is the slow rank (50 ms vs 20 ms per
    step), so value must be whole-job frames / the SLOW rank's time."""
    import json
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_stub_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[1].strip() == ""                       # only rank 0 prints
    lines = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["ms_per_step"] >= 50.0 and line["ms_per_step"] < 120.0
    assert abs(line["value"] - 2 * 5 * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    assert line["cpu_baseline"] is None and line["config"]["stub"] is True


def test_bench_bare_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun environment must not print an n_gpus: 1 line: it starts the two ranks itself
    (child `python -m torch.distributed.run`, before any GPU call in the parent) and forwards rank 0's line; a launcher whose
    WORLD_SIZE disagrees with --gpus is refused with a non-zero exit (the reference is one command too: 2Haff/README.md:69)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--stub-step-ms", "30", "--batch", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 3
    assert abs(line["value"] - 2 * 4 * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    env["WORLD_SIZE"] = "3"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-step-ms", "5"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_bucket_layout_with_alternating_dtypes():
    """LisaTrainable keeps matrices in bf16 and biases / norm vectors in fp32, alternating in state-dict order: buckets are
    filled per dtype (one open bucket each), so the count is ~ total bytes / bucket bytes per dtype, not one per tensor."""
    from haff import train_ops as T
    named = []
    for i in range(40):
        named.append((f"model.visual_model.mask_decoder_left.blk{i}.weight", torch.nn.Parameter(torch.zeros(256, 256, dtype=torch.bfloat16))))
        named.append((f"model.visual_model.mask_decoder_left.blk{i}.bias", torch.nn.Parameter(torch.zeros(256, dtype=torch.float32))))
    named.append(("lm_head.weight", torch.nn.Parameter(torch.zeros(2000, 256, dtype=torch.bfloat16))))
    bucket = 1 << 20
    red = T.GradBucketReducer(named, bucket_bytes=bucket)
    by_dtype = {}
    for b in red.buckets:
        by_dtype.setdefault(b["dtype"], []).append(b)
        assert all(p.dtype == b["dtype"] for p in b["params"])
        assert sum(p.numel() for p in b["params"]) == b["flat"].numel()
    bf16_bytes = (40 * 256 * 256 + 2000 * 256) * 2
    assert len(by_dtype[torch.float32]) == 1                      # 40 KiB of biases: one bucket
    assert len(by_dtype[torch.bfloat16]) <= bf16_bytes // bucket + 2
    assert len(red.buckets) <= 8, len(red.buckets)                # (81 tensors gave 80 buckets before)
    for name, p in named:                                         # every gradient is a view into its bucket
        assert p.grad is not None and p.grad.shape == p.shape
        b = red.buckets[red.bucket_of[id(p)]]
        assert p.grad.untyped_storage().data_ptr() == b["flat"].untyped_storage().data_ptr()


def test_clip_coefficient_on_the_device_matches_the_host_rule():
    """train_ops.clip_coef_device = min(1, max_norm / (norm + 1e-6)) as a one-element fp32 tensor (the clip coefficient stays
    on the device so that the optimizer launches do not wait for a host read of the gradient norm); lr schedule sanity."""
    from haff import train_ops as T
    for norm in (0.0, 0.3, 1.0, 4.0, 123.5):
        want = min(1.0, 1.0 / (norm + 1e-6))
        got = T.clip_coef_device(torch.tensor(norm, dtype=torch.float64), 1.0)
        assert got.dtype == torch.float32 and got.shape == (1,)
        assert abs(float(got) - want) <= 1e-6 * max(want, 1.0)
    assert abs(float(T.clip_coef_device(torch.tensor(10.0), 5.0)) - 0.5) < 1e-6
    assert T.warmup_decay_lr(0, 1000, 3e-4) == 0.0 and abs(T.warmup_decay_lr(100, 1000, 3e-4) - 3e-4) < 1e-12
