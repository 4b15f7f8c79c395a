"""Backward kernels of the fine-tune path: each autograd.Function (HIP forward + HIP backward through the C-ABI)
against torch autograd on a plain fp32 PyTorch reference of the same op. fp32 mode ~1e-5, bf16 mode ~2e-2 of scale."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _ag():
    import haff  # noqa: F401
    from haff import autograd as A
    return A


def _rand(shape, dev, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def _tol(dtype):
    return 3e-5 if dtype == torch.float32 else 3e-2


def _close(got, ref, tol, what):
    got, ref = got.float(), ref.float()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert math.isfinite(err) and err <= tol * scale, f"{what}: err {err:.3g} scale {scale:.3g}"


def _leaf(t):
    return t.clone().requires_grad_(True)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(200, 96, 64), (37, 43, 72), (5, 8, 256), (2100, 256, 64), (1300, 128, 32)])   # the last two: 16-byte column sums for db
def test_linear_fn(dev, dtype, M, N, K):
    A = _ag()
    x, w = _leaf(_rand((M, K), dev, dtype, 1)), _leaf(_rand((N, K), dev, dtype, 2, K ** -0.5))
    b = _leaf(_rand((N,), dev, torch.float32, 3))
    r = _leaf(_rand((M, N), dev, dtype, 4))
    gy = _rand((M, N), dev, dtype, 5)
    y = A.linear(x, w, b, r)
    y.backward(gy)
    xr, wr, br, rr = (_leaf(t.detach().float()) for t in (x, w, b, r))
    yr = F.linear(xr, wr, br) + rr
    yr.backward(gy.float())
    tol = _tol(dtype)
    _close(y, yr, tol, "y")
    _close(x.grad, xr.grad, tol, "dx")
    _close(w.grad, wr.grad, tol, "dw")
    _close(b.grad, br.grad, tol, "db")
    _close(r.grad, rr.grad, tol, "dres")
    # frozen weight with a resident transposed copy
    x2 = _leaf(x.detach())
    wt = A.transpose(w.detach(), Rp=(N + 7) // 8 * 8)[0]
    A.linear(x2, w.detach(), None, None, wt).backward(gy)
    x3 = _leaf(x.detach().float())
    F.linear(x3, w.detach().float()).backward(gy.float())
    _close(x2.grad, x3.grad, tol, "dx frozen")


@pytest.mark.parametrize("dtype", DT)
def test_act_swiglu_norms(dev, dtype):
    A = _ag()
    tol = _tol(dtype)
    x = _rand((70, 96), dev, dtype, 10)
    gy = _rand((70, 96), dev, dtype, 11)
    for code, ref in ((1, F.gelu), (3, F.relu), (4, F.silu)):
        a = _leaf(x)
        A.act(a, code).backward(gy)
        b = _leaf(x.float())
        ref(b).backward(gy.float())
        _close(a.grad, b.grad, tol, f"act {code}")
    Fh = 48
    gu = _leaf(_rand((70, 2 * Fh), dev, dtype, 12))
    gy2 = _rand((70, Fh), dev, dtype, 13)
    y = A.swiglu(gu)
    y.backward(gy2)
    gr = _leaf(gu.detach().float())
    g3 = gr.view(70, Fh // 16, 2, 16)
    yr = (F.silu(g3[:, :, 0]) * g3[:, :, 1]).reshape(70, Fh)
    yr.backward(gy2.float())
    _close(y, yr, tol, "swiglu")
    _close(gu.grad, gr.grad, tol, "swiglu grad")
    C = 256
    x = _rand((33, C), dev, dtype, 14, 2.0) + 0.5
    w, b = _leaf(_rand((C,), dev, torch.float32, 15) + 1.0), _leaf(_rand((C,), dev, torch.float32, 16))
    gy = _rand((33, C), dev, dtype, 17)
    a = _leaf(x)
    A.layernorm(a, w, b, 1e-5).backward(gy)
    ar, wr, br = _leaf(x.float()), _leaf(w.detach()), _leaf(b.detach())
    F.layer_norm(ar, (C,), wr, br, 1e-5).backward(gy.float())
    _close(a.grad, ar.grad, tol, "ln dx")
    _close(w.grad, wr.grad, tol, "ln dw")
    _close(b.grad, br.grad, tol, "ln db")
    a = _leaf(x)
    A.rmsnorm(a, w.detach(), 1e-5).backward(gy)
    ar = _leaf(x.float())
    (ar * torch.rsqrt(ar.pow(2).mean(-1, keepdim=True) + 1e-5) * w.detach()).backward(gy.float())
    _close(a.grad, ar.grad, tol, "rms dx")


@pytest.mark.parametrize("M,N1,N2,strided", [(300, 8, 8, False), (1000, 256, 128, False), (4133, 136, 264, True), (70000, 256, 256, False),
                                             (513, 32, 2048, True)])
def test_gemm_tn(dev, M, N1, N2, strided):
    """haff_gemm_tn_bf16: a^T @ b with the contraction over the rows (the weight gradient of a trainable Linear) against fp32
    torch; ragged row counts (partial last slab / split), tiles cut by N1 / N2, operands that are column slices of wider
    tensors; a second launch gives the same bits (split partials are added in index order); LinearFn takes it for dW."""
    A = _ag()
    a = _rand((M, N1 + (24 if strided else 0)), dev, torch.bfloat16, 80)
    b = _rand((M, N2 + (8 if strided else 0)), dev, torch.bfloat16, 81)
    av, bv = (a[:, 16:16 + N1], b[:, :N2]) if strided else (a, b)
    assert A.gemm_tn_supported(av, bv)
    out = A.gemm_tn(av, bv)
    ref = av.float().t() @ bv.float()
    _close(out, ref, 1.2e-2, "gemm_tn bf16 out")
    out32 = A.gemm_tn(av, bv, torch.float32)
    _close(out32, ref, 2e-3 if M < 5000 else 4e-3, "gemm_tn f32 out")   # fp32 accumulation of bf16 products
    assert torch.equal(A.gemm_tn(av, bv), out)
    assert not A.gemm_tn_supported(a[:, 1:1 + N1], b[:, :N2])   # a misaligned column slice goes the transposing way
    # through LinearFn: dW by the TN product equals dW by transposes + the NT product to bf16 rounding
    if N1 <= 512:
        x, w = _leaf(bv.contiguous()), _leaf(_rand((N1, N2), dev, torch.bfloat16, 82, N2 ** -0.5))
        gy = av.contiguous()
        A.linear(x, w).backward(gy)
        x2, w2 = _leaf(x.detach()), _leaf(w.detach())
        A.TN_WEIGHT_GRADIENTS = False
        try:
            A.linear(x2, w2).backward(gy)
        finally:
            A.TN_WEIGHT_GRADIENTS = True
        _close(w.grad, w2.grad, 1.2e-2, "LinearFn dW: TN product vs transposes")
        _close(w.grad, gy.float().t() @ x.detach().float(), 1.2e-2, "LinearFn dW vs fp32")


@pytest.mark.parametrize("C", [4096, 5120])
def test_norm_adjoints_llama_width(dev, C):
    """bf16 rows of 4096 / 5120 take the one-pass register-resident adjoint (norm_bwd_vec_kernel): RMSNorm and LayerNorm
    (with the weight / bias gradients) against fp32 torch autograd, 37 rows (a partial last workgroup), and against the generic
    kernel's result on a column count next to it."""
    A = _ag()
    dtype, R = torch.bfloat16, 37
    x = _rand((R, C), dev, dtype, 14, 2.0) + 0.5
    w, b = _leaf(_rand((C,), dev, torch.float32, 15) + 1.0), _leaf(_rand((C,), dev, torch.float32, 16))
    gy = _rand((R, C), dev, dtype, 17)
    a = _leaf(x)
    A.layernorm(a, w, b, 1e-5).backward(gy)
    ar, wr, br = _leaf(x.float()), _leaf(w.detach()), _leaf(b.detach())
    F.layer_norm(ar, (C,), wr, br, 1e-5).backward(gy.float())
    _close(a.grad, ar.grad, 3e-2, "ln dx")
    _close(w.grad, wr.grad, 3e-2, "ln dw")
    _close(b.grad, br.grad, 3e-2, "ln db")
    a = _leaf(x)
    A.rmsnorm(a, w.detach(), 1e-5).backward(gy)
    ar = _leaf(x.float())
    (ar * torch.rsqrt(ar.pow(2).mean(-1, keepdim=True) + 1e-5) * w.detach()).backward(gy.float())
    _close(a.grad, ar.grad, 3e-2, "rms dx")
    assert (a.grad.float() - ar.grad).abs().mean().item() <= 4e-3 * ar.grad.abs().mean().item() + 1e-6   # bf16 rounding of dx only


@pytest.mark.parametrize("C,dtype", [(4096, torch.bfloat16), (5120, torch.bfloat16), (256, torch.bfloat16), (192, torch.float32)])
def test_resid_rmsnorm_fn(dev, C, dtype):
    """(x, rmsnorm(x)) as one node (ResidRMSNormFn: haff_norm_bwd_add): a pre-norm residual block y = x + f(norm(x)) built on it has
    the gradients of the plain rmsnorm node + autograd's own accumulation add (bit for bit where the bf16 sum rounds once either
    way is not guaranteed: the fused form adds in fp32 and rounds ONCE, so it is held to the fp32 reference and must be at least
    as close as the two-node form), with only one branch used, and for both row kernels (Llama widths / generic)."""
    A = _ag()
    R = 37
    x = _rand((R, C), dev, dtype, 31, 2.0) + 0.5
    w = _rand((C,), dev, torch.float32, 32) + 1.0
    m = _rand((C, C), dev, dtype, 33, C ** -0.5)
    gy = _rand((R, C), dev, dtype, 34)

    def block(x_, fused):
        if fused:
            xs, h = A.resid_rmsnorm(x_, w, 1e-5)
        else:
            xs, h = x_, A.rmsnorm(x_, w, 1e-5)
        return A.linear(h, m, None, xs)          # y = x + norm(x) @ m.T (the residual enters LinearFn's epilogue)
    a, b = _leaf(x), _leaf(x)
    ya, yb = block(a, True), block(b, False)
    assert torch.equal(ya, yb)
    ya.backward(gy)
    yb.backward(gy)
    xr = _leaf(x.float())
    (xr + (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * w) @ m.float().t()).backward(gy.float())
    tol = 3e-2 if dtype == torch.bfloat16 else 1e-5
    _close(a.grad, xr.grad, tol, "fused residual + rmsnorm dx")
    e_f = (a.grad.float() - xr.grad).abs().mean().item()
    e_s = (b.grad.float() - xr.grad).abs().mean().item()
    assert e_f <= 1.05 * e_s + 1e-7, (e_f, e_s)
    # one branch only: the stream output unused / the norm output unused
    c = _leaf(x)
    A.resid_rmsnorm(c, w, 1e-5)[1].backward(gy)
    d = _leaf(x)
    A.rmsnorm(d, w, 1e-5).backward(gy)
    assert torch.equal(c.grad, d.grad)
    e = _leaf(x)
    A.resid_rmsnorm(e, w, 1e-5)[0].backward(gy)
    assert torch.equal(e.grad, gy)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,Nq,Nk,d,causal", [(2, 4, 37, 37, 32, True), (2, 8, 6, 200, 16, False), (1, 8, 130, 6, 16, False),
                                                (2, 2, 50, 50, 128, True)])
def test_attention_fn(dev, dtype, B, H, Nq, Nk, d, causal):
    A = _ag()
    tol = _tol(dtype)
    q, k, v = (_leaf(_rand((B, n, H * d), dev, dtype, s)) for n, s in ((Nq, 20), (Nk, 21), (Nk, 22)))
    go = _rand((B, Nq, H * d), dev, dtype, 23)
    scale = d ** -0.5
    o = A.attention(q, k, v, H, scale, causal)
    o.backward(go)
    qr, kr, vr = (_leaf(t.detach().float()) for t in (q, k, v))
    q4, k4, v4 = (t.view(B, -1, H, d).permute(0, 2, 1, 3) for t in (qr, kr, vr))
    s = (q4 @ k4.transpose(-1, -2)) * scale
    if causal:
        m = torch.arange(Nk, device=dev)[None, :] > torch.arange(Nq, device=dev)[:, None] + (Nk - Nq)
        s = s.masked_fill(m, float("-inf"))
    orf = (torch.softmax(s, -1) @ v4).permute(0, 2, 1, 3).reshape(B, Nq, H * d)
    orf.backward(go.float())
    _close(o, orf, tol, "attn out")
    _close(q.grad, qr.grad, tol, "dq")
    _close(k.grad, kr.grad, tol, "dk")
    _close(v.grad, vr.grad, tol, "dv")


@pytest.mark.parametrize("B,H,Nq,Nk,causal", [(2, 4, 351, 351, True), (1, 3, 200, 200, False), (2, 2, 64, 64, True),
                                              (1, 2, 65, 130, True), (1, 2, 130, 70, False)])
def test_flash_attention_fn(dev, B, H, Nq, Nk, causal):
    """The flash pair of the fine-tune path (haff_attention_lse_bf16 + haff_attention_bwd_bf16, d = 128, bf16: no probabilities
    in HBM) against (1) fp32 torch autograd of the definition and (2) the materialised AttentionFn it replaces; ragged last
    blocks (351 = 5 x 64 + 31), Nk != Nq (causal offset q_pos0 = Nk - Nq), one block, non-causal; a second backward gives the
    same bits (dq is summed by the owning workgroup in a fixed order: no atomics)."""
    A = _ag()
    d, dtype = 128, torch.bfloat16
    q, k, v = (_leaf(_rand((B, n, H * d), dev, dtype, s)) for n, s in ((Nq, 40), (Nk, 41), (Nk, 42)))
    go = _rand((B, Nq, H * d), dev, dtype, 43)
    scale = d ** -0.5
    assert A.FLASH_TRAINING_ATTENTION
    o = A.attention(q, k, v, H, scale, causal)
    assert o.grad_fn is not None and "Flash" in type(o.grad_fn).__name__
    o.backward(go)
    # (1) the definition in fp32
    qr, kr, vr = (_leaf(t.detach().float()) for t in (q, k, v))
    q4, k4, v4 = (t.view(B, -1, H, d).permute(0, 2, 1, 3) for t in (qr, kr, vr))
    s_ = (q4 @ k4.transpose(-1, -2)) * scale
    if causal:
        m = torch.arange(Nk, device=dev)[None, :] > torch.arange(Nq, device=dev)[:, None] + (Nk - Nq)
        s_ = s_.masked_fill(m, float("-inf"))
    orf = (torch.softmax(s_, -1) @ v4).permute(0, 2, 1, 3).reshape(B, Nq, H * d)
    orf.backward(go.float())
    for got, ref, what in ((o, orf, "out"), (q.grad, qr.grad, "dq"), (k.grad, kr.grad, "dk"), (v.grad, vr.grad, "dv")):
        _close(got, ref, 3e-2, f"flash {what} vs fp32 autograd")
    # (2) the materialised form on the same inputs
    qm, km, vm = (_leaf(t.detach()) for t in (q, k, v))
    A.FLASH_TRAINING_ATTENTION = False
    try:
        om = A.attention(qm, km, vm, H, scale, causal)
        om.backward(go)
    finally:
        A.FLASH_TRAINING_ATTENTION = True
    for got, ref, what in ((o, om, "out"), (q.grad, qm.grad, "dq"), (k.grad, km.grad, "dk"), (v.grad, vm.grad, "dv")):
        _close(got, ref, 2e-2, f"flash {what} vs materialised")
    # the flash gradients are no further from fp32 than the materialised ones (mean error)
    e_f = sum((g.float() - r).abs().mean().item() for g, r in ((q.grad, qr.grad), (k.grad, kr.grad), (v.grad, vr.grad)))
    e_m = sum((g.float() - r).abs().mean().item() for g, r in ((qm.grad, qr.grad), (km.grad, kr.grad), (vm.grad, vr.grad)))
    assert e_f <= 1.25 * e_m, (e_f, e_m)
    # repeatable to the bit
    q2, k2, v2 = (_leaf(t.detach()) for t in (q, k, v))
    A.attention(q2, k2, v2, H, scale, causal).backward(go)
    assert torch.equal(q2.grad, q.grad) and torch.equal(k2.grad, k.grad) and torch.equal(v2.grad, v.grad)


@pytest.mark.parametrize("B,T,heads,K,r,drop", [(2, 37, 2, 256, 8, False), (3, 50, 3, 384, 8, True), (1, 129, 2, 256, 4, True),
                                                (2, 16, 1, 128, 1, False), (1, 40, 40, 5120, 8, True),   # the last: 13B widths
                                                (3, 50, 3, 384, 8, "two"), (1, 129, 2, 256, 4, "two"), (2, 291, 32, 4096, 8, "two")])
def test_lora_qkv_rope_fn(dev, B, T, heads, K, r, drop):
    """The adapted q|k|v projection + RoPE as one node (csrc/lora.hip: haff_lora_qkv_rope_fwd / _bwd, haff_lora_dx, haff_lora_tn
    and the role-swapped weight-streaming products) against (1) fp32 torch autograd of the definition (peft LoRA on q_proj /
    v_proj, rotate-half RoPE) and (2) the chain of separate nodes it replaces; row counts that are not multiples of 16 or 8,
    ranks below 8 (zero-padded adapters), without a dropout mask, with ONE mask for both adapters and with peft's TWO independent
    masks ("two": each adapted Linear has its own lora_dropout, train_ds.py:218-230); repeatable to the bit."""
    A = _ag()
    dtype, d = torch.bfloat16, 128
    H, M = heads * d, B * T
    scale = 16.0 / r
    x = _leaf(_rand((M, K), dev, dtype, 50))
    w = _rand((3 * H, K), dev, dtype, 51, K ** -0.5)
    wt = A.transpose(w)[0]
    aq, av = (_leaf(_rand((r, K), dev, dtype, s_, K ** -0.5)) for s_ in (52, 53))
    bq, bv = (_leaf(_rand((H, r), dev, dtype, s_, 0.3)) for s_ in (54, 55))
    keep = keep_v = None
    if drop:
        g = torch.Generator(device="cpu").manual_seed(56)
        keep = ((torch.rand((M, K), generator=g) >= 0.25).float() / 0.75).to(dtype).to(dev)
        keep_v = keep
        if drop == "two":   # peft's semantics: the q and the v adapter drop their inputs independently (train_ds.py:218-230)
            keep_v = ((torch.rand((M, K), generator=g) >= 0.25).float() / 0.75).to(dtype).to(dev)
            keep = (keep, keep_v)
    keep_q = keep[0] if isinstance(keep, tuple) else keep
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    ang = torch.arange(T + 3, dtype=torch.float32)[:, None] * inv[None, :]
    cs = torch.cat([ang.cos(), ang.sin()], 1).contiguous().to(dev)
    gq, gk, gv = (_rand((M, H), dev, dtype, s_) for s_ in (57, 58, 59))
    assert A.lora_qkv_rope_supported(x, w, aq, heads)
    q, k, v = A.lora_qkv_rope(x, w, wt, aq, bq, av, bv, cs, T, heads, scale, keep)
    assert "LoraQKVRope" in type(q.grad_fn).__name__
    torch.autograd.backward([q, k, v], [gq, gk, gv])

    # (1) the definition in fp32
    def rope(t):
        t4 = t.view(B, T, heads, d)
        co, si = cs[:T, :d // 2].view(1, T, 1, d // 2), cs[:T, d // 2:].view(1, T, 1, d // 2)
        t1, t2 = t4[..., :d // 2], t4[..., d // 2:]
        return torch.cat([t1 * co - t2 * si, t2 * co + t1 * si], -1).reshape(M, H)

    xr, aqr, avr, bqr, bvr = (_leaf(t.detach().float()) for t in (x, aq, av, bq, bv))
    xd = xr if keep is None else xr * keep_q.float()
    xdv = xr if keep is None else xr * keep_v.float()
    qkv = xr @ w.float().t()
    qr = rope(qkv[:, :H] + scale * (xd @ aqr.t()) @ bqr.t())
    kr = rope(qkv[:, H:2 * H])
    vr = qkv[:, 2 * H:] + scale * (xdv @ avr.t()) @ bvr.t()
    torch.autograd.backward([qr, kr, vr], [gq.float(), gk.float(), gv.float()])
    for got, ref, what in ((q, qr, "q"), (k, kr, "k"), (v, vr, "v"), (x.grad, xr.grad, "dx"), (aq.grad, aqr.grad, "dAq"),
                           (av.grad, avr.grad, "dAv"), (bq.grad, bqr.grad, "dBq"), (bv.grad, bvr.grad, "dBv")):
        assert got.shape == ref.shape, what
        _close(got, ref, 3e-2, f"fused lora {what} vs fp32 autograd")
    # (2) the separate nodes on the same inputs
    xs, aqs, avs, bqs, bvs = (_leaf(t.detach()) for t in (x, aq, av, bq, bv))

    def padk(b_):
        if b_.shape[1] % 8 == 0:
            return b_
        return torch.cat([b_, torch.zeros((b_.shape[0], 8 - b_.shape[1] % 8), dtype=b_.dtype, device=b_.device)], 1)

    def pada(a_):   # rank rows padded to 8 so that the next product's K is a multiple of 8
        if a_.shape[0] % 8 == 0:
            return a_
        return torch.cat([a_, torch.zeros((8 - a_.shape[0] % 8, a_.shape[1]), dtype=a_.dtype, device=a_.device)], 0)

    qkvs = A.linear(xs, w, None, None, wt)
    hl = xs if keep is None else xs * keep_q
    hv = xs if keep is None else xs * keep_v
    qs = A.rope(A.add(qkvs[:, :H], A.scale(A.linear(A.linear(hl, pada(aqs)), padk(bqs)), scale)), cs, T, heads, d)
    ks = A.rope(qkvs[:, H:2 * H], cs, T, heads, d)
    vs = A.add(qkvs[:, 2 * H:], A.scale(A.linear(A.linear(hv, pada(avs)), padk(bvs)), scale))
    torch.autograd.backward([qs, ks, vs], [gq, gk, gv])
    for got, ref, what in ((q, qs, "q"), (k, ks, "k"), (v, vs, "v"), (x.grad, xs.grad, "dx"), (aq.grad, aqs.grad, "dAq"),
                           (av.grad, avs.grad, "dAv"), (bq.grad, bqs.grad, "dBq"), (bv.grad, bvs.grad, "dBv")):
        _close(got, ref, 3e-2, f"fused lora {what} vs separate nodes")
    # the fused node is no further from fp32 than the chain it replaces
    pairs_f = ((x.grad, xr.grad), (aq.grad, aqr.grad), (bq.grad, bqr.grad), (av.grad, avr.grad), (bv.grad, bvr.grad))
    pairs_s = ((xs.grad, xr.grad), (aqs.grad, aqr.grad), (bqs.grad, bqr.grad), (avs.grad, avr.grad), (bvs.grad, bvr.grad))
    e_f = sum(((g_.float() - r_).abs().mean() / r_.abs().mean()).item() for g_, r_ in pairs_f)
    e_s = sum(((g_.float() - r_).abs().mean() / r_.abs().mean()).item() for g_, r_ in pairs_s)
    assert e_f <= 1.25 * e_s, (e_f, e_s)
    # repeatable to the bit
    x2, aq2, av2, bq2, bv2 = (_leaf(t.detach()) for t in (x, aq, av, bq, bv))
    q2, k2, v2 = A.lora_qkv_rope(x2, w, wt, aq2, bq2, av2, bv2, cs, T, heads, scale, keep)
    torch.autograd.backward([q2, k2, v2], [gq, gk, gv])
    assert torch.equal(q2, q) and torch.equal(v2, v)
    for a_, b_ in ((x2.grad, x.grad), (aq2.grad, aq.grad), (av2.grad, av.grad), (bq2.grad, bq.grad), (bv2.grad, bv.grad)):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("dtype", DT)
def test_rope_bmm_embed_ce(dev, dtype):
    A = _ag()
    tol = _tol(dtype)
    B, T, H, d = 2, 9, 3, 32
    x = _leaf(_rand((B * T, H * d), dev, dtype, 30))
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, device=dev).float() / d))
    ang = torch.arange(T, device=dev).float()[:, None] * inv[None, :]
    cs = torch.cat([ang.cos(), ang.sin()], 1).contiguous()
    gy = _rand((B * T, H * d), dev, dtype, 31)
    y = A.rope(x, cs, T, H, d)
    y.backward(gy)
    xr = _leaf(x.detach().float())
    x4 = xr.view(B, T, H, d)
    cos = torch.cat([ang.cos(), ang.cos()], 1).view(1, T, 1, d)
    sin = torch.cat([ang.sin(), ang.sin()], 1).view(1, T, 1, d)
    yr = (x4 * cos + torch.cat([-x4[..., d // 2:], x4[..., :d // 2]], -1) * sin).reshape(B * T, H * d)
    yr.backward(gy.float())
    _close(y, yr, tol, "rope")
    _close(x.grad, xr.grad, tol, "rope grad")
    # batched NT product (hypernetwork x upscaled embedding)
    a, b = _leaf(_rand((3, 1, 32), dev, dtype, 32)), _leaf(_rand((3, 100, 32), dev, dtype, 33))
    gc = _rand((3, 1, 100), dev, dtype, 34)
    c = A.bmm_nt(a, b)
    c.backward(gc)
    ar, br = _leaf(a.detach().float()), _leaf(b.detach().float())
    cr = ar @ br.transpose(1, 2)
    cr.backward(gc.float())
    _close(c, cr, tol, "bmm")
    _close(a.grad, ar.grad, tol, "bmm da")
    _close(b.grad, br.grad, tol, "bmm db")
    # embedding
    Wt = _leaf(_rand((50, 64), dev, dtype, 35))
    ids = torch.tensor([[1, 5, 5, -200, 7], [0, 49, 3, 3, 3]], device=dev)
    ge = _rand((2, 5, 64), dev, dtype, 36)
    A.embed(Wt, ids).backward(ge)
    Wr = _leaf(Wt.detach().float())
    mask = (ids >= 0)
    (F.embedding(ids.clamp(min=0), Wr) * mask[..., None]).backward(ge.float())
    _close(Wt.grad, Wr.grad, tol, "embed grad")
    # cross entropy with ignore_index
    lg = _leaf(_rand((40, 323), dev, dtype, 37, 2.0))
    lab = torch.randint(0, 323, (40,), device=dev)
    lab[::3] = -100
    loss = A.cross_entropy(lg, lab)
    loss.backward()
    lr = _leaf(lg.detach().float())
    lossr = F.cross_entropy(lr, lab, ignore_index=-100)
    lossr.backward()
    _close(loss, lossr, 1e-5 if dtype == torch.float32 else 1e-2, "ce")
    _close(lg.grad, lr.grad, tol, "ce grad")


def test_losses_and_bilinear(dev):
    A = _ag()
    n, hw = 2, 3000
    x = _leaf(_rand((n, hw), dev, torch.float32, 40, 2.0))
    t = (torch.rand((n, hw), device=dev) > 0.6).float()
    w = [1.0, 0.0]
    out = A.mask_losses(x, t, w)
    coef = torch.tensor([[2.0, 0.5], [2.0, 0.5]], device=dev)
    (out * coef).sum().backward()
    xr = _leaf(x.detach())
    tot = 0
    for i in range(n):
        z = (w[i] * xr[i]).view(1, 1, hw)
        bce = F.binary_cross_entropy_with_logits(z, t[i].view(1, 1, hw), reduction="none").flatten(1, 2).mean(1).sum()
        p = z.sigmoid().flatten(1, 2)
        tt = t[i].view(1, hw)
        dice = (1 - (2 * (p / 1000 * tt).sum(-1) + 1e-6) / ((p / 1000).sum(-1) + (tt / 1000).sum(-1) + 1e-6)).sum()
        _close(out[i, 0], bce, 1e-5, "bce")
        _close(out[i, 1], dice, 1e-5, "dice")
        tot = tot + 2.0 * bce + 0.5 * dice
    tot.backward()
    _close(x.grad, xr.grad, 1e-4, "mask loss grad")
    # taxonomy CE on already soft-maxed probabilities
    z = _leaf(_rand((3, 4), dev, torch.float32, 41))
    tgt = torch.tensor([[1., 0, 0, 0], [0, 0, 1., 0], [0, 1., 0, 0]], device=dev)
    loss, probs = A.taxonomy_ce(z, tgt)
    loss.sum().backward()
    zr = _leaf(z.detach())
    pr = torch.softmax(zr, -1)
    lr = torch.stack([F.cross_entropy(pr[i:i + 1], tgt[i:i + 1]) for i in range(3)])
    lr.sum().backward()
    _close(loss, lr, 1e-5, "tax ce")
    _close(probs, pr, 1e-6, "tax probs")
    _close(z.grad, zr.grad, 1e-5, "tax grad")
    # bilinear adjoint
    m = _leaf(_rand((2, 56, 56), dev, torch.float32, 42))
    g = _rand((2, 120, 90), dev, torch.float32, 43)
    up = A.resize_bilinear(m, (56, 56), (224, 224))
    A.resize_bilinear(up, (224, 168), (120, 90)).backward(g)
    mr = _leaf(m.detach())
    u = F.interpolate(mr[:, None], (224, 224), mode="bilinear", align_corners=False)
    F.interpolate(u[..., :224, :168], (120, 90), mode="bilinear", align_corners=False)[:, 0].backward(g)
    _close(m.grad, mr.grad, 1e-5, "bilinear grad")
    # the gather adjoint at a non-integer up-scale of a crop (rows outside the crop get no gradient), twice: same bits
    m2 = _leaf(_rand((3, 40, 48), dev, torch.float32, 44))
    g2 = _rand((3, 101, 77), dev, torch.float32, 45)
    A.resize_bilinear(m2, (33, 48), (101, 77)).backward(g2)
    m2r = _leaf(m2.detach())
    F.interpolate(m2r[:, None, :33, :48], (101, 77), mode="bilinear", align_corners=False)[:, 0].backward(g2)
    _close(m2.grad, m2r.grad, 1e-5, "bilinear grad (crop, 3.06x / 1.6x)")
    assert m2.grad[:, 33:].abs().max().item() == 0.0
    m3 = _leaf(m2.detach())
    A.resize_bilinear(m3, (33, 48), (101, 77)).backward(g2)
    assert torch.equal(m3.grad, m2.grad)


def test_adamw_matches_torch(dev):
    import haff  # noqa: F401
    from haff import train_ops as T
    p = _rand((1000,), dev, torch.float32, 50)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0)
    state = T.AdamWState(p.clone())
    for step in range(1, 4):
        g = _rand((1000,), dev, torch.float32, 60 + step)
        ref.grad = g.clone()
        opt.step()
        T.adamw_step(state, g, lr=1e-3, betas=(0.9, 0.95), eps=1e-8, wd=0.0, gscale=1.0)
    _close(state.master, ref.detach(), 1e-6, "adamw")
    # the clip coefficient from the device (haff_adamw_step_dev) = the same update with the host float
    g = _rand((1000,), dev, torch.float32, 70, 5.0)
    s_host, s_dev = T.AdamWState(p.clone()), T.AdamWState(p.clone())
    norm = T.grad_norm([g])
    coef = T.clip_coef_device(norm, 1.0)
    T.adamw_step(s_host, g, lr=1e-3, gscale=0.5 * min(1.0, 1.0 / (float(norm) + 1e-6)))
    T.adamw_step(s_dev, g, lr=1e-3, gscale=0.5, gscale_dev=coef)
    assert float(coef) < 1.0
    _close(s_dev.master, s_host.master, 1e-6, "adamw with a device clip coefficient")
    _close(s_dev.v, s_host.v, 1e-5, "adamw v with a device clip coefficient")
    assert abs(T.grad_norm([g, g]).item() - math.sqrt(2) * g.norm().item()) < 1e-2


def test_loss_upstream_scalars_stay_in_fp32(dev):
    """the gradient a loss weight sends into CrossEntropyFn / TaxonomyCEFn is applied by haff_scale_dev in fp32 from
    DEVICE memory — not rounded to bf16 first (0.3 -> 0.30078125: 0.26 % off against the mask gradients of the same step)."""
    import haff  # noqa: F401
    from haff import autograd as A
    g = torch.Generator(device="cpu").manual_seed(5)
    # (4096 rows: the gradient's big entries — the -1 at each label — are rounded to bf16 one by one, +-0.2 % each; their mean
    # error over 3277 valid rows is ~4e-5, far below the 0.26 % a bf16 upstream scalar would add to every element)
    logits = (torch.randn((4096, 323), generator=g) * 2).to(torch.bfloat16).to(dev).requires_grad_(True)
    labels = torch.randint(0, 323, (4096,), generator=g).to(dev)
    labels[::5] = -100
    w = torch.tensor(0.3, device=dev)
    (A.cross_entropy(logits, labels) * w).backward()
    ref_logits = logits.detach().float().requires_grad_(True)
    (torch.nn.functional.cross_entropy(ref_logits, labels, ignore_index=-100) * 0.3).backward()
    got, ref = logits.grad.float(), ref_logits.grad
    # against the fp32 gradient: only the bf16 rounding of each stored element (2^-9 relative), no common 0.26 % scale error
    ratio = (got * ref).sum() / (ref * ref).sum()
    assert abs(float(ratio) - 1.0) < 4e-4, float(ratio)
    assert (got - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()   # (two bf16 roundings: the stored dlogits, the scaled result)
    # per-row form
    a = torch.randn((6, 40), generator=g).to(dev)
    alpha = torch.tensor([0.3, -1.7, 0.0, 2.5, 1e-3, 1.0], device=dev)
    assert torch.equal(A.scale_dev(a, alpha, per_row=True), a * alpha[:, None])
    assert torch.equal(A.scale_dev(a, alpha[3]), a * 2.5)


def test_bucket_adamw_equals_the_per_tensor_update(dev):
    """train_ops.BucketAdamW (one fused launch per gradient bucket, parameters re-pointed at flat buffers) against
    train_ops.adamw_step per tensor on the same gradients: bit-identical master weights, moments and bf16 parameter copies
    over several steps, with the device clip coefficient; `states[name]` are views the checkpoint code can read and write."""
    import haff  # noqa: F401
    from haff import train_ops as T
    shapes = [("lm_head.weight", (37, 64), torch.bfloat16), ("model.text_hidden_fcs.0.0.bias", (64,), torch.float32),
              ("model.layers.1.self_attn.q_proj.lora_A", (8, 64), torch.bfloat16), ("model.layers.0.self_attn.q_proj.lora_B", (64, 8), torch.bfloat16),
              ("model.visual_model.mask_decoder_left.x.bias", (5,), torch.float32), ("model.embed_tokens.weight", (323, 64), torch.bfloat16)]
    named = [(k, _rand(s, dev, torch.float32, 100 + i).to(dt).requires_grad_(True)) for i, (k, s, dt) in enumerate(shapes)]
    ref_p = {k: p.detach().clone() for k, p in named}
    ref_s = {k: T.AdamWState(p.detach().clone()) for k, p in named}
    red = T.GradBucketReducer(named, bucket_bytes=4096)
    opt = T.BucketAdamW(red, named)
    assert len(opt.buckets) == len(red.buckets) >= 3
    for k, p in named:                                        # re-pointed, values kept
        assert torch.equal(p.detach(), ref_p[k]) and torch.equal(opt.states[k].master, ref_p[k].float())
    for step in range(1, 4):
        for i, (k, p) in enumerate(named):
            p.grad.copy_(_rand(p.shape, dev, torch.float32, 200 + 10 * step + i).to(p.dtype))
        clip = T.clip_coef_device(T.grad_norm(red.grads()), 1.0)
        for k, p in named:
            T.adamw_step(ref_s[k], p.grad.clone(), lr=1e-3, gscale=0.5, param_lp=ref_p[k], gscale_dev=clip)
        opt.step(lr=1e-3, gscale=0.5, gscale_dev=clip)
        for k, p in named:
            st = opt.states[k]
            assert torch.equal(st.master, ref_s[k].master) and torch.equal(st.m, ref_s[k].m) and torch.equal(st.v, ref_s[k].v), (step, k)
            assert torch.equal(p.detach(), ref_p[k]), (step, k)      # the parameter the model reads follows the master copy
            assert st.step == step
    # resume: values written through the per-key views reach the flat buffers and, after refresh_lp, the parameters
    opt.states["lm_head.weight"].master.fill_(0.25)
    opt.refresh_lp()
    assert float(named[0][1].detach().float().mean()) == 0.25


def test_ordered_reductions_are_bitwise_repeatable_and_right(dev):
    """Round 4: column sums, the mask-loss sums, the embedding scatter and the gradient norm without atomics (per-block partials
    added in index order / rows of one id added in row order): equal to fp64 references to fp32 rounding and IDENTICAL over
    repeated launches (the atomic forms of rounds 1-3 differed run to run)."""
    import haff  # noqa: F401
    from haff import autograd as A, train_ops as T
    assert A.ORDERED_REDUCTIONS
    g = torch.Generator(device="cpu").manual_seed(9)
    # column sums: tall bf16 (the decoders' 65536 x 256 image-token rows), short fp32, ragged sizes
    for (R, C, dt) in ((65536, 256, torch.bfloat16), (300, 37, torch.float32), (5000, 1280, torch.bfloat16), (1, 8, torch.float32)):
        x = (torch.randn((R, C), generator=g) + 0.3).to(dt).to(dev)
        outs = [A.colsum(x) for _ in range(3)]
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        ref = x.double().sum(0)
        assert (outs[0].double() - ref).abs().max().item() <= 2e-6 * x.double().abs().sum(0).max().item() + 1e-6
    # gradient norm over several buckets of different dtypes
    grads = [torch.randn((1 << 20,), generator=g).to(torch.bfloat16).to(dev), torch.randn((777,), generator=g).to(dev),
             torch.randn((3_000_001,), generator=g).to(torch.bfloat16).to(dev)]
    norms = [T.grad_norm(grads) for _ in range(3)]
    assert torch.equal(norms[0], norms[1]) and torch.equal(norms[0], norms[2])
    ref = sum(float(t.double().pow(2).sum()) for t in grads) ** 0.5
    assert abs(float(norms[0]) - ref) <= 2e-6 * ref
    # mask-loss statistics
    x = (torch.randn((3, 1024 * 1024), generator=g) * 3).to(dev)
    t = (torch.rand((3, 1024 * 1024), generator=g) > 0.5).float().to(dev)
    wg = [1.0, 0.0, 2.0]
    a = [A.mask_losses(x, t, wg) for _ in range(3)]
    assert torch.equal(a[0], a[1]) and torch.equal(a[0], a[2])
    z = x.double() * torch.tensor(wg, dtype=torch.float64, device=dev)[:, None]
    bce = torch.nn.functional.binary_cross_entropy_with_logits(z, t.double(), reduction="none").mean(1)
    p = torch.sigmoid(z)
    dice = 1 - (2 * (p / 1000 * t.double()).sum(1) + 1e-6) / ((p / 1000).sum(1) + (t.double() / 1000).sum(1) + 1e-6)
    assert (a[0][:, 0].double() - bce).abs().max().item() <= 1e-5 and (a[0][:, 1].double() - dice).abs().max().item() <= 1e-5
    # embedding scatter with repeated ids and ignored rows
    ids = torch.tensor([[5, 7, 5, 5, 0, 319, 7, 5]] * 4).to(dev)
    w = torch.randn((320, 64), generator=g).to(torch.bfloat16).to(dev).requires_grad_(True)
    dy = torch.randn((4, 8, 64), generator=g).to(torch.bfloat16).to(dev)
    gs = []
    for _ in range(3):
        w.grad = None
        A.embed(w, ids).backward(dy)
        gs.append(w.grad.clone())
    assert torch.equal(gs[0], gs[1]) and torch.equal(gs[0], gs[2])
    ref = torch.zeros((320, 64), dtype=torch.float64, device=dev).index_add_(0, ids.reshape(-1), dy.double().reshape(-1, 64))
    assert (gs[0].double() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
