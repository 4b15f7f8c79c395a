"""Op-level numerics: every HIP kernel (through the C-ABI) against a plain PyTorch fp32 reference of the same op.

Tolerances: bf16 kernels read bf16-rounded inputs and round their output to bf16 once (rel 2^-8 per element);
the reference is computed in fp32 from the SAME bf16-rounded inputs, so the bound is a few bf16 ulps of the
output scale. fp32 (parity-mode) kernels must agree to ~1e-5 relative.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    import haff  # noqa: F401
    from haff import ops
    return ops


def _close(got, ref, rel, what):
    got = got.float()
    ref = ref.float()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert math.isfinite(err), f"{what}: non-finite output"
    assert err <= rel * scale, f"{what}: max|err|={err:.4g} vs scale {scale:.4g} (rel {err / scale:.3g} > {rel})"


def _rand(shape, dev, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


ACT_REF = {
    0: lambda x: x,
    1: lambda x: F.gelu(x),
    2: lambda x: x * torch.sigmoid(1.702 * x),
    3: lambda x: F.relu(x),
    4: lambda x: F.silu(x),
}


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 72), (1000, 384, 1280), (37, 1003, 256), (4900, 256, 128)])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_bf16_epilogues(dev, M, N, K, act):
    ops = _ops()
    x = _rand((M, K), dev, torch.bfloat16, 1)
    w = _rand((N, K), dev, torch.bfloat16, 2, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 3)
    resid = _rand((M, N), dev, torch.bfloat16, 4)
    ref = ACT_REF[act](x.float() @ w.float().T + bias) + resid.float()
    got = ops.linear(x, w, bias=bias, act=act, resid=resid)
    _close(got, ref, 1.2e-2, f"gemm_bf16 {M}x{N}x{K} act{act}")
    got32 = ops.linear(x, w, bias=bias, act=act, out_dtype=torch.float32)
    _close(got32, ACT_REF[act](x.float() @ w.float().T + bias), 2e-3, "gemm_bf16 f32-out")


@pytest.mark.parametrize("M", [1, 3, 4, 5, 8])
@pytest.mark.parametrize("H,N2", [(4096, 12288), (512, 1000), (5120, 27648)])
def test_linear_rms_carries_the_norm(dev, M, H, N2):
    """haff_gemm_bf16_rms: a residual product that also emits per-workgroup sums of squares of its bf16 output, and a product
    on norm-weight-folded weights that turns those partials into 1/rms — against RMSNorm + product in fp32, and the
    residual stream bit-equal to the plain kernel's."""
    ops = _ops()
    eps = 1e-5
    a = _rand((M, H), dev, torch.bfloat16, 41)
    x0 = _rand((M, H), dev, torch.bfloat16, 42)
    wo = _rand((H, H), dev, torch.bfloat16, 43, H ** -0.5)
    gamma = (1.0 + 0.3 * _rand((H,), dev, torch.float32, 44))
    w2 = _rand((N2, H), dev, torch.bfloat16, 45, H ** -0.5)
    w2f = (w2.float() * gamma[None, :]).to(torch.bfloat16).contiguous()
    parts = torch.full((H // 16, 16), float("nan"), device=dev)
    x1 = ops.linear_rms(a, wo, resid=x0, out=x0.clone(), ssq_out=parts)
    assert torch.equal(x1, ops.linear(a, wo, resid=x0))
    ssq = (x1.float() ** 2).view(M, H // 16, 16).sum(-1).T            # [H/16, M]
    _close(parts[:, :M], ssq, 1e-5, "per-workgroup sums of squares")
    rstd = torch.rsqrt((x1.float() ** 2).mean(-1, keepdim=True) + eps)
    swiglu = N2 % 32 == 0
    y = ops.linear_rms(x1, w2f, ssq_in=parts, eps=eps, swiglu=swiglu)
    lin = (x1.float() * rstd) @ w2f.float().T
    if swiglu:
        lin = lin.view(M, N2 // 32, 2, 16)
        lin = (F.silu(lin[:, :, 0]) * lin[:, :, 1]).reshape(M, N2 // 2)
    _close(y, lin, 1.2e-2, "folded rms product")
    assert torch.equal(y, ops.linear_rms(x1, w2f, ssq_in=parts, eps=eps, swiglu=swiglu))
    if not swiglu:   # against the unfused kernels (norm weight applied to the activations instead of the weights)
        _close(y, ops.linear(ops.rmsnorm(x1, gamma, eps), w2), 2.5e-2, "vs norm kernel + product")


@pytest.mark.parametrize("M,N,K", [(288, 4096, 4096), (288, 4096, 11008), (257, 1024, 4096), (257, 3072, 1024), (257, 1003, 1024),
                                   (100, 520, 2048), (1000, 256, 1280), (291, 12288, 4096), (320, 640, 1056), (129, 5120, 5120),
                                   (65, 512, 512), (321, 4096, 4096)])
def test_gemm_bf16_split_k_tiles(dev, M, N, K):
    """Few-tile products (one-frame prefill o_proj / down_proj, the CLIP tower): ops.linear hands the library a workspace and
    the 128x128 kernel runs K slices as a batched launch, summed in slice order by the reduce kernel with the whole epilogue
    (bias, activation, residual, row map, fp32 / bf16 output, ragged N). Same tolerance as the unsplit kernel; repeatable
    bit for bit; identical to the unsplit kernel's result up to fp32 summation order."""
    ops = _ops()
    x = _rand((M, K), dev, torch.bfloat16, 21)
    w = _rand((N, K), dev, torch.bfloat16, 22, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 23)
    resid = _rand((M, N), dev, torch.bfloat16, 24)
    y = x.float() @ w.float().T + bias
    got = ops.linear(x, w, bias=bias, act=2, resid=resid)
    _close(got, ACT_REF[2](y) + resid.float(), 1.2e-2, "split-K quick-gelu+resid")
    assert torch.equal(got, ops.linear(x, w, bias=bias, act=2, resid=resid))
    got32 = ops.linear(x, w, bias=bias, out_dtype=torch.float32)
    _close(got32, y, 2e-3, "split-K f32 out")
    unsplit = ops.linear(x, w, bias=bias, out_dtype=torch.float32, tile_cfg=1)
    _close(got32, unsplit.float(), 1e-4, "split vs unsplit")
    perm = torch.randperm(M, device=dev).to(torch.int32)
    perm[::7] = -1
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    ops.linear(x, w, bias=bias, resid=resid, row_map=perm, out=out)
    ref = torch.zeros((M, N), dtype=torch.float32, device=dev)
    keep = perm >= 0
    ref[perm[keep].long()] = y[keep] + resid.float()[perm[keep].long()]
    _close(out, ref, 1.2e-2, "split-K row_map")
    xs = _rand((M, K + 64), dev, torch.bfloat16, 25)[:, 32:32 + K]     # strided view, 64-B aligned rows
    if (K + 64) % 8 == 0:
        _close(ops.linear(xs, w, out_dtype=torch.float32), xs.float() @ w.float().T, 2e-3, "split-K strided A")


@pytest.mark.parametrize("M,N,K", [(64, 22016, 4096), (40, 16384, 4096), (100, 2048, 4096), (64, 32000, 4096), (288, 22016, 4096),
                                   (320, 1024, 1024), (65, 4096, 256)])
def test_gemm_bf16_split_k_swiglu(dev, M, N, K):
    """33..64-row products on very wide weights (batch-64 decode gate/up, lm_head) and SwiGLU products with few tiles: K
    slices of the 128x128 tile, (gate, up) pairs formed in the reduce kernel. Against the fp32 reference and the unsplit tile."""
    ops = _ops()
    x = _rand((M, K), dev, torch.bfloat16, 51)
    w = _rand((N, K), dev, torch.bfloat16, 52, K ** -0.5)
    F_ = N // 2
    wi = torch.stack([w[:F_].reshape(F_ // 16, 16, K), w[F_:].reshape(F_ // 16, 16, K)], dim=1).reshape(N, K).contiguous()
    y = x.float() @ w.float().T
    ref = F.silu(y[:, :F_]) * y[:, F_:]
    got = ops.linear(x, wi, swiglu=True)
    _close(got, ref, 1.5e-2, "split-K swiglu")
    assert torch.equal(got, ops.linear(x, wi, swiglu=True))
    _close(got, ops.linear(x, wi, swiglu=True, tile_cfg=1).float(), 1.2e-2, "split vs unsplit swiglu")
    got32 = ops.linear(x, wi, swiglu=True, out_dtype=torch.float32)
    _close(got32, ref, 3e-3, "split-K swiglu f32 out")
    plain = ops.linear(x, w, out_dtype=torch.float32)     # the same shapes without SwiGLU (lm_head-like)
    _close(plain, y, 2e-3, "split-K wide plain")


@pytest.mark.parametrize("tile_cfg", [1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(512, 512, 128), (700, 1003, 256), (300, 520, 64), (1111, 256, 1280), (5000, 4500, 64)])
def test_gemm_bf16_forced_tiles(dev, tile_cfg, M, N, K):
    """The three tiles of the template (128^2 / 4 waves; 256^2 and 192 x 256 / 8 waves with the ping-pong ring loop) through
    every epilogue feature, ragged edges included (K = 64: the single-K-tile path of the ring loop, with and without a next
    tile); the auto heuristic only picks the 8-wave tiles on large problems."""
    ops = _ops()
    x = _rand((M, K), dev, torch.bfloat16, 11)
    w = _rand((N, K), dev, torch.bfloat16, 12, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 13)
    resid = _rand((M, N), dev, torch.bfloat16, 14)
    y = x.float() @ w.float().T + bias
    got = ops.linear(x, w, bias=bias, act=1, resid=resid, tile_cfg=tile_cfg)
    _close(got, F.gelu(y) + resid.float(), 1.2e-2, f"cfg{tile_cfg} gelu+resid")
    got32 = ops.linear(x, w, bias=bias, out_dtype=torch.float32, tile_cfg=tile_cfg)
    _close(got32, y, 2e-3, f"cfg{tile_cfg} f32 out")
    perm = torch.randperm(M, device=dev).to(torch.int32)
    perm[::5] = -1
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    ops.linear(x, w, bias=bias, resid=resid, row_map=perm, out=out, tile_cfg=tile_cfg)
    ref = torch.zeros((M, N), dtype=torch.float32, device=dev)
    keep = perm >= 0
    ref[perm[keep].long()] = y[keep] + resid.float()[perm[keep].long()]
    _close(out, ref, 1.2e-2, f"cfg{tile_cfg} row_map")
    if N % 32 == 0:
        F_ = N // 2
        wg, wu = w[:F_], w[F_:]
        wi = torch.stack([wg.reshape(F_ // 16, 16, K), wu.reshape(F_ // 16, 16, K)], dim=1).reshape(N, K).contiguous()
        got = ops.linear(x, wi, swiglu=True, tile_cfg=tile_cfg)
        _close(got, F.silu(x.float() @ wg.float().T) * (x.float() @ wu.float().T), 1.2e-2, f"cfg{tile_cfg} swiglu")


@pytest.mark.parametrize("M,N,K", [(2808, 4096, 4096), (2808, 4096, 11008), (1500, 1280, 1280), (2808, 22016, 4096)])
def test_gemm_bf16_192_row_tile(dev, M, N, K):
    """Row counts that fill the chip's rounds badly in 256-row tiles (the fine-tune step's 2808-row products: 176 tiles of 256^2
    on 256 CUs) take the 192 x 256 form of the 8-wave tile (240 tiles) by themselves: the auto choice equals the forced one bit
    for bit, agrees with the fp32 reference like the other tiles, through the residual / SwiGLU / gather epilogues and over
    the many-tiles-per-workgroup persistent path (22016 columns: 1290 tiles on 256 workgroups)."""
    ops = _ops()
    x = _rand((M, K), dev, torch.bfloat16, 71)
    w = _rand((N, K), dev, torch.bfloat16, 72, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 73)
    resid = _rand((M, N), dev, torch.bfloat16, 74)
    y = x.float() @ w.float().T + bias
    got = ops.linear(x, w, bias=bias, resid=resid)
    _close(got, y + resid.float(), 1.2e-2, "192-row tile, auto")
    assert torch.equal(got, ops.linear(x, w, bias=bias, resid=resid, tile_cfg=3))
    _close(got, ops.linear(x, w, bias=bias, resid=resid, tile_cfg=2).float(), 1e-2, "192-row vs 256-row tile")
    got32 = ops.linear(x, w, bias=bias, out_dtype=torch.float32, tile_cfg=3)
    _close(got32, y, 2e-3, "192-row tile f32 out")
    _close(got32, ops.linear(x, w, bias=bias, out_dtype=torch.float32, tile_cfg=2), 1e-5, "192-row vs 256-row tile, f32")
    if N % 32 == 0 and N <= 8192:
        F_ = N // 2
        wi = torch.stack([w[:F_].reshape(F_ // 16, 16, K), w[F_:].reshape(F_ // 16, 16, K)], dim=1).reshape(N, K).contiguous()
        ys = x.float() @ w.float().T
        _close(ops.linear(x, wi, swiglu=True, tile_cfg=3), F.silu(ys[:, :F_]) * ys[:, F_:], 1.5e-2, "192-row tile swiglu")
    a_map = torch.randint(0, M, (M,), device=dev).to(torch.int32)
    _close(ops.linear(x, w, bias=bias, a_map=a_map), x.float()[a_map.long()] @ w.float().T + bias, 1.2e-2, "192-row tile gather")


def test_gemm_stream_cap_is_scheduling_only_and_per_stream(dev):
    """haff_gemm_stream_cap: fewer workgroups per launch of the persistent tile ON ONE STREAM (the CUs a caller leaves to another
    stream) compute the same tiles — bit-identical outputs for every accepted value, on whole and ragged tile grids, with a fused
    epilogue; values that are not a multiple of 8 in 8..256 change nothing; the call returns the stream's previous setting; a cap
    set on one stream is not seen by launches (or queries) on another."""
    ops = _ops()
    other = torch.cuda.Stream(device=dev)
    assert ops.gemm_stream_cap(256) == 256
    try:
        for (M, N, K, tile) in ((16384, 1280, 1280, 0), (9000, 2560, 1280, 0), (2808, 4096, 4096, 3)):
            x = _rand((M, K), dev, torch.bfloat16, 91)
            w = _rand((N, K), dev, torch.bfloat16, 92, K ** -0.5)
            bias = _rand((N,), dev, torch.float32, 93)
            resid = _rand((M, N), dev, torch.bfloat16, 94)
            ref = ops.linear(x, w, bias=bias, resid=resid, tile_cfg=tile)
            for cap in (8, 96, 128, 160, 192, 216, 224, 248):
                old = ops.gemm_stream_cap(cap)
                assert ops.gemm_stream_cap(cap) == cap, (old, cap)
                assert ops.gemm_stream_cap(0, stream=other) == 256          # the other stream keeps its own (default) setting
                out = ops.linear(x, w, bias=bias, resid=resid, tile_cfg=tile)
                assert torch.equal(out, ref), f"cap {cap} changed the product {M}x{N}x{K}"
            ops.gemm_stream_cap(224)
            for bad in (0, 7, 100, 260, -8):
                assert ops.gemm_stream_cap(bad) == 224     # a query: the setting stays
            # two capped streams at once: each keeps its own value, launches on both give the same bits
            assert ops.gemm_stream_cap(64, stream=other) == 256
            torch.cuda.current_stream().synchronize()
            other.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(other):
                assert ops.gemm_stream_cap(0) == 64
                out2 = ops.linear(x, w, bias=bias, resid=resid, tile_cfg=tile)
            assert ops.gemm_stream_cap(0) == 224
            other.synchronize()
            assert torch.equal(out2, ref)
            assert ops.gemm_stream_cap(256, stream=other) == 64
            ops.gemm_stream_cap(256)
    finally:
        ops.gemm_stream_cap(256)
        ops.gemm_stream_cap(256, stream=other)


@pytest.mark.parametrize("M,N,K", [(2808, 4096, 4096), (1500, 1280, 1280), (401, 512, 128)])
def test_gemm_bf16_192_row_tile_row_maps(dev, M, N, K):
    """the 192 x 256 tile is picked by the launcher for any launch without a folded norm, i.e. also with an output row
    map (the non-DMA map path: both DMA stagings are 256 rows wide and compiled out for this tile), the half-filled second
    orow slot of its 96-row wave tile, dropped rows (-1) and a ragged last M-tile (M % 192 != 0), with and without a residual,
    with the activations, and with the A-side gather. Each against the 128 x 128 tile (bit for bit: same k order) and fp32.
    (RoPE and the row statistics have entry points of their own that launch the 256-row tile only.)"""
    ops = _ops()
    x = _rand((M, K), dev, torch.bfloat16, 81)
    w = _rand((N, K), dev, torch.bfloat16, 82, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 83)
    g = torch.Generator().manual_seed(84)
    rows_out = M + 37
    perm = torch.randperm(rows_out, generator=g)[:M]
    drop = torch.rand((M,), generator=g) < 0.07
    rmap = torch.where(drop, torch.full((M,), -1), perm).to(torch.int32).to(dev)
    resid = _rand((rows_out, N), dev, torch.bfloat16, 85)
    y = x.float() @ w.float().T + bias
    keep = (~drop).to(dev)
    for act in (0, 1, 3):
        for use_res in (False, True):
            outs = []
            for cfg_id in (3, 1):
                out = torch.full((rows_out, N), 5.0, dtype=torch.bfloat16, device=dev)
                ops.linear(x, w, bias=bias, act=act, resid=resid if use_res else None, row_map=rmap, out=out, tile_cfg=cfg_id)
                outs.append(out)
            assert torch.equal(outs[0], outs[1]), f"192-row tile with a row map differs from the 128 x 128 tile (act {act}, resid {use_res})"
            exp = torch.full((rows_out, N), 5.0, device=dev)
            exp[rmap[keep].long()] = ACT_REF[act](y)[keep] + (resid.float()[rmap[keep].long()] if use_res else 0.0)
            _close(outs[0], exp, 1.2e-2, f"192-row tile, row map, act {act}, resid {use_res}")
    out32 = torch.full((rows_out, N), 5.0, dtype=torch.float32, device=dev)
    ops.linear(x, w, bias=bias, row_map=rmap, out=out32, tile_cfg=3)
    exp = torch.full((rows_out, N), 5.0, device=dev)
    exp[rmap[keep].long()] = y[keep]
    _close(out32, exp, 2e-3, "192-row tile, row map, f32 out")
    a_map = torch.randint(0, M, (M,), device=dev).to(torch.int32)
    got = ops.linear(x, w, bias=bias, a_map=a_map, row_map=rmap, out=torch.full((rows_out, N), 5.0, dtype=torch.bfloat16, device=dev))
    exp = torch.full((rows_out, N), 5.0, device=dev)
    exp[rmap[keep].long()] = (x.float()[a_map.long()] @ w.float().T + bias)[keep]
    _close(got, exp, 1.2e-2, "gather + row map (auto tile)")


@pytest.mark.parametrize("M", [1, 3, 8, 16, 17, 31, 32, 33, 50, 64])
@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 11008), (5120, 13824), (1003, 256), (320, 11008), (64, 128), (998, 384)])
def test_gemm_skinny(dev, M, N, K):
    """M <= 64 takes the weight-streaming kernel (one workgroup per 16 weight rows, K split over 4 waves, 1 / 2 / 4
    activation tiles of 16 rows): every epilogue feature, ragged M and N (N = 998: the packed 8-byte store falls back
    to element stores on the last column group and on odd row pitches), gather and scatter maps."""
    ops = _ops()
    x = _rand((M + 5, K), dev, torch.bfloat16, 15)
    w = _rand((N, K), dev, torch.bfloat16, 16, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 17)
    resid = _rand((M, N), dev, torch.bfloat16, 18)
    y = x[:M].float() @ w.float().T + bias
    _close(ops.linear(x[:M], w, bias=bias, act=1, resid=resid), F.gelu(y) + resid.float(), 1.2e-2, "skinny gelu+resid")
    _close(ops.linear(x[:M], w, out_dtype=torch.float32), x[:M].float() @ w.float().T, 2e-3, "skinny f32 out")
    a_map = torch.randint(0, M + 5, (M,), device=dev).to(torch.int32)
    _close(ops.linear(x, w, bias=bias, a_map=a_map), x.float()[a_map.long()] @ w.float().T + bias, 1.2e-2, "skinny gather")
    perm = torch.randperm(M + 3, device=dev)[:M].to(torch.int32)
    if M > 1:
        perm[0] = -1
    out = torch.zeros((M + 3, N), dtype=torch.bfloat16, device=dev)
    ops.linear(x[:M], w, bias=bias, row_map=perm, out=out)
    ref = torch.zeros((M + 3, N), dtype=torch.float32, device=dev)
    keep = perm >= 0
    ref[perm[keep].long()] = y[keep]
    _close(out, ref, 1.2e-2, "skinny row_map")
    if N % 32 == 0:
        F_ = N // 2
        wg, wu = w[:F_], w[F_:]
        wi = torch.stack([wg.reshape(F_ // 16, 16, K), wu.reshape(F_ // 16, 16, K)], dim=1).reshape(N, K).contiguous()
        got = ops.linear(x[:M], wi, swiglu=True)
        _close(got, F.silu(x[:M].float() @ wg.float().T) * (x[:M].float() @ wu.float().T), 1.2e-2, "skinny swiglu")


@pytest.mark.parametrize("M,N,K", [(700, 512, 1280), (5, 1003, 256), (16, 320, 1280), (40, 320, 1280), (300, 384, 128), (4096, 2560, 1280)])
@pytest.mark.parametrize("rms", [False, True])
def test_gemm_folded_norm(dev, M, N, K, rms):
    """haff_row_stats + haff_gemm_bf16_ln == LayerNorm / RMSNorm followed by the Linear (every tile + the skinny path),
    with activation, residual and an output row map; SwiGLU with RMSNorm as Llama's gate/up uses it."""
    ops = _ops()
    x = (_rand((M, K), dev, torch.float32, 19, 2.0) + 0.5).to(torch.bfloat16)
    w = _rand((N, K), dev, torch.bfloat16, 20, K ** -0.5)
    gamma = _rand((K,), dev, torch.float32, 21, 0.2) + 1.0
    beta = None if rms else _rand((K,), dev, torch.float32, 22, 0.3)
    bias = None if rms else _rand((N,), dev, torch.float32, 23)
    resid = _rand((M, N), dev, torch.bfloat16, 24)
    eps = 1e-5
    xf = x.float()
    if rms:
        xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * gamma
        st_ref = torch.stack([torch.zeros(M, device=dev), torch.rsqrt(xf.pow(2).mean(-1) + eps)], 1)
    else:
        xn = F.layer_norm(xf, (K,), gamma, beta, eps)
        st_ref = torch.stack([xf.mean(-1), torch.rsqrt(xf.var(-1, unbiased=False) + eps)], 1)
    stats = ops.row_stats(x, eps, rms=rms)
    _close(stats, st_ref, 1e-5, "row_stats")
    wf, colsum, bf = ops.fold_norm(w, gamma, beta, bias)
    y = xn @ w.float().T + (bias if bias is not None else 0.0)
    for cfg in ((0,) if M <= 64 else (0, 1, 2)):
        if cfg == 2 and K % 64:
            continue
        got = ops.linear(x, wf, bias=bf, act=1, resid=resid, ln_stats=stats, ln_colsum=None if rms else colsum, tile_cfg=cfg)
        _close(got, F.gelu(y) + resid.float(), 2e-2, f"folded norm cfg{cfg} gelu+resid")
    perm = torch.randperm(M + 3, device=dev)[:M].to(torch.int32)
    out = torch.zeros((M + 3, N), dtype=torch.bfloat16, device=dev)
    ops.linear(x, wf, bias=bf, row_map=perm, out=out, ln_stats=stats, ln_colsum=None if rms else colsum)
    ref = torch.zeros((M + 3, N), dtype=torch.float32, device=dev)
    ref[perm.long()] = y
    _close(out, ref, 2e-2, "folded norm row_map")
    if rms and N % 32 == 0:
        F_ = N // 2
        wg, wu = w[:F_], w[F_:]
        wi = torch.stack([wg.reshape(F_ // 16, 16, K), wu.reshape(F_ // 16, 16, K)], dim=1).reshape(N, K).contiguous()
        wif, _, _ = ops.fold_norm(wi, gamma)
        got = ops.linear(x, wif, swiglu=True, ln_stats=stats)
        _close(got, F.silu(xn @ wg.float().T) * (xn @ wu.float().T), 2e-2, "folded rmsnorm swiglu")


@pytest.mark.parametrize("K", [128, 1280])
def test_gemm_lean_register_epilogue_variants(dev, K):
    """The 8-wave tile's lean register epilogue (round 4; csrc/gemm_bf16.hip `lean_epilogue`) in each of its instances, on whole
    256 x 256 tiles: plain, bias, bias + exact GELU, folded LayerNorm (+ GELU), folded RMSNorm, an output row map with dropped
    rows (the windowed q|k|v scatter of image_encoder.py:179-183), bf16 residual in place (proj / lin2, :186-193), residual +
    row statistics. Against fp32 torch on the same bf16 operands AND against the 128 x 128 tile (generic epilogue): the two
    tiles add the same bf16 products, in another order — equal to fp32 rounding, i.e. at most one bf16 ulp of the output."""
    ops = _ops()
    import haff.ops as hops
    M, N = 4096, 2560   # (the folded-norm entry picks its tile by shape: 160 tiles of 256 x 256 beat 640 of 128 x 128)
    x = (_rand((M, K), dev, torch.float32, 301, 1.5) + 0.4).to(torch.bfloat16)
    w = _rand((N, K), dev, torch.bfloat16, 302, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 303)
    gamma = _rand((K,), dev, torch.float32, 304, 0.2) + 1.0
    beta = _rand((K,), dev, torch.float32, 305, 0.3)
    resid = _rand((M, N), dev, torch.bfloat16, 306, 2.0)
    xf, wf32 = x.float(), w.float()
    y = xf @ wf32.T

    def both(ref, what, **kw):
        big = ops.linear(x, kw.pop("w", w), tile_cfg=2, **kw)
        _close(big, ref, 1.2e-2, what + " vs fp32")
        return big
    both(y, "plain")
    b = both(y + bias, "bias", bias=bias)
    small = ops.linear(x, w, bias=bias, tile_cfg=1)
    _close(b, small.float(), 2.0 ** -7, "bias: 256 tile vs 128 tile")
    g = both(F.gelu(y + bias), "bias + gelu", bias=bias, act=1)
    _close(g, ops.linear(x, w, bias=bias, act=1, tile_cfg=1).float(), 2.0 ** -7, "gelu: 256 tile vs 128 tile")
    # large arguments: the clamp of the polynomial's range (|x| > 3 sqrt 2) must give x and 0
    wbig = (w.float() * 6.0).to(torch.bfloat16)
    both(F.gelu(xf @ wbig.float().T + bias), "gelu, large arguments", w=wbig, bias=bias, act=1)
    r = both(y + bias + resid.float(), "bias + residual", bias=bias, resid=resid)
    _close(r, ops.linear(x, w, bias=bias, resid=resid, tile_cfg=1).float(), 2.0 ** -7, "residual: 256 tile vs 128 tile")
    inplace = resid.clone()
    ops.linear(x, w, bias=bias, resid=inplace, out=inplace, tile_cfg=2)
    assert torch.equal(inplace, r), "in-place residual differs"
    # folded norms
    for rms in (False, True):
        eps = 1e-6
        if rms:
            xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * gamma
        else:
            xn = F.layer_norm(xf, (K,), gamma, beta, eps)
        stats = ops.row_stats(x, eps, rms=rms)
        wfold, colsum, bf = ops.fold_norm(w, gamma, None if rms else beta, bias)
        yn = xn @ wf32.T + bias
        for act, ref in ((0, yn), (1, F.gelu(yn))):
            got = ops.linear(x, wfold, bias=bf, act=act, ln_stats=stats, ln_colsum=None if rms else colsum, tile_cfg=2)
            _close(got, ref, 2e-2, f"folded {'rms' if rms else 'layer'}norm act{act}")
        # scatter with dropped rows
        gperm = torch.Generator(device="cpu").manual_seed(307)
        perm = torch.randperm(M + 200, generator=gperm)[:M].to(torch.int32)
        perm[::37] = -1
        perm = perm.to(dev)
        out = torch.full((M + 200, N), 7.0, dtype=torch.bfloat16, device=dev)
        ops.linear(x, wfold, bias=bf, row_map=perm, out=out, ln_stats=stats, ln_colsum=None if rms else colsum, tile_cfg=2)
        ref = torch.full((M + 200, N), 7.0, dtype=torch.float32, device=dev)
        keep = perm >= 0
        ref[perm[keep].long()] = yn[keep]
        _close(out, ref, 2e-2, "folded norm + row map with dropped rows")
        out2 = torch.full((M + 200, N), 7.0, dtype=torch.bfloat16, device=dev)
        ops.linear(x, wfold, bias=bf, act=1, row_map=perm, out=out2, ln_stats=stats, ln_colsum=None if rms else colsum, tile_cfg=2)
        ref[perm[keep].long()] = F.gelu(yn)[keep]
        _close(out2, ref, 2e-2, "folded norm + gelu + row map")
    # residual + statistics of the stored rows
    if K % 64 == 0:
        lib = hops.load_library()
        out = resid.clone()
        part = torch.empty((M, N // 64, 2), dtype=torch.float32, device=dev)
        stats = torch.empty((M, 2), dtype=torch.float32, device=dev)
        rc = lib.haff_gemm_bf16_rowstats(x.data_ptr(), x.stride(0), None, 0, w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0),
                                         bias.data_ptr(), out.data_ptr(), out.stride(0), M, N, K, part.data_ptr(), None)
        assert rc == 0
        assert lib.haff_row_stats_finalize(part.data_ptr(), stats.data_ptr(), M, N // 64, N, 1e-6, None) == 0
        assert torch.equal(out, r), "rowstats product differs from the residual product"
        want = ops.row_stats(out, 1e-6)
        assert (stats[:, 0] - want[:, 0]).abs().max().item() <= 2e-3 * want[:, 0].abs().max().item() + 1e-4
        assert ((stats[:, 1] - want[:, 1]).abs() / want[:, 1]).max().item() <= 2e-3


def test_gemm_bf16_identity_asymmetric(dev):
    """A = I with an asymmetric W catches a swapped C layout (cdna guide §3)."""
    ops = _ops()
    K = 128
    x = torch.eye(K, dtype=torch.bfloat16, device=dev)
    w = (torch.arange(256 * K, device=dev).reshape(256, K) % 251).to(torch.bfloat16)
    got = ops.linear(x, w, out_dtype=torch.float32)
    assert torch.equal(got, w.float().T.contiguous())


def test_gemm_bf16_rowmap_and_strided(dev):
    ops = _ops()
    M, N, K = 500, 256, 192
    xfull = _rand((M, K + 64), dev, torch.bfloat16, 5)
    x = xfull[:, 32:32 + K]  # strided view, 64-byte aligned start
    w = _rand((N, K), dev, torch.bfloat16, 6, K ** -0.5)
    perm = torch.randperm(M, device=dev).to(torch.int32)
    perm[::7] = -1
    resid = _rand((M, N), dev, torch.bfloat16, 7)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    ops.linear(x, w, resid=resid, row_map=perm, out=out)
    ref = torch.zeros((M, N), dtype=torch.float32, device=dev)
    y = x.float() @ w.float().T
    keep = perm >= 0
    ref[perm[keep].long()] = y[keep] + resid.float()[perm[keep].long()]
    _close(out, ref, 1.2e-2, "gemm row_map")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_swiglu(dev, dtype):
    ops = _ops()
    M, F_, K = 200, 192, 128
    x = _rand((M, K), dev, dtype, 8)
    wg = _rand((F_, K), dev, dtype, 9, K ** -0.5)
    wu = _rand((F_, K), dev, dtype, 10, K ** -0.5)
    wi = torch.stack([wg.view(F_ // 16, 16, K), wu.view(F_ // 16, 16, K)], dim=1).reshape(2 * F_, K).contiguous()
    got = ops.linear(x, wi, swiglu=True)
    ref = F.silu(x.float() @ wg.float().T) * (x.float() @ wu.float().T)
    _close(got, ref, 1.2e-2 if dtype == torch.bfloat16 else 2e-5, "swiglu")


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (130, 70, 36), (300, 1003, 256), (1, 256, 256), (6, 32, 2048), (4100, 128, 260),
                                   (513, 262, 1284), (8192, 256, 256)])
def test_gemm_f32(dev, M, N, K):
    ops = _ops()
    x = _rand((M, K), dev, torch.float32, 11)
    w = _rand((N, K), dev, torch.float32, 12, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 13)
    resid = _rand((M, N), dev, torch.float32, 14)
    got = ops.linear(x, w, bias=bias, act=1, resid=resid)
    ref = F.gelu(x.double() @ w.double().T + bias.double()) + resid.double()
    _close(got, ref, 2e-5, "gemm_f32")


@pytest.mark.parametrize("M", [1, 3, 7, 8, 12, 16])
@pytest.mark.parametrize("N,K", [(4096, 4096), (256, 4096), (256, 256), (2048, 256), (256, 2048), (1003, 260), (32, 256), (5, 8)])
def test_gemm_f32_few_rows(dev, M, N, K):
    """The weight-streaming fp32 kernel behind haff_gemm_f32 for M <= 16 (decoder tail at a handful of prompts): every
    epilogue feature, ragged N, K not a multiple of the 256-float chunk, scattered / dropped output rows; repeatable."""
    ops = _ops()
    x = _rand((M, K), dev, torch.float32, 31)
    w = _rand((N, K), dev, torch.float32, 32, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 33)
    resid = _rand((M + 3, N), dev, torch.float32, 34)
    y = x.double() @ w.double().T + bias.double()
    got = ops.linear(x, w, bias=bias, act=1, resid=resid[:M])
    _close(got, F.gelu(y) + resid[:M].double(), 2e-5, "gemm_f32 few rows")
    assert torch.equal(got, ops.linear(x, w, bias=bias, act=1, resid=resid[:M]))
    _close(ops.linear(x, w, act=3), F.relu(x.double() @ w.double().T), 2e-5, "gemm_f32 few rows relu")
    perm = torch.randperm(M + 3, device=dev)[:M].to(torch.int32)
    if M > 2:
        perm[1] = -1
    out = torch.full((M + 3, N), 7.0, device=dev)
    ops.linear(x, w, bias=bias, resid=resid, row_map=perm, out=out)
    ref = torch.full((M + 3, N), 7.0, device=dev, dtype=torch.float64)
    keep = perm >= 0
    ref[perm[keep].long()] = y[keep] + resid.double()[perm[keep].long()]
    _close(out, ref, 2e-5, "gemm_f32 few rows row_map")


def test_gemm_f32_rowmap_views_and_tails(dev):
    """f32-input MFMA GEMM: scattered output rows (negative = dropped), strided A / C views whose rows are not 16-B
    aligned (scalar epilogue path), K tail of the 16-deep tile, N tail of the 4-column lane group."""
    ops = _ops()
    M, N, K = 333, 203, 72
    xb = _rand((M, K + 4), dev, torch.float32, 21)
    x = xb[:, 4:]
    w = _rand((N, K), dev, torch.float32, 22, K ** -0.5)
    bias = _rand((N + 1,), dev, torch.float32, 23)[1:]          # 4-byte aligned only
    g = torch.Generator().manual_seed(24)
    perm = torch.randperm(M + 40, generator=g)[:M].to(torch.int32)
    perm[::7] = -1
    row_map = perm.to(dev)
    out_b = torch.full((M + 40, N + 3), 7.0, device=dev)
    out = out_b[:, 3:]                                           # rows start 12 B past a 16-B boundary
    ops.linear(x, w, bias=bias, act=3, row_map=row_map, out=out)
    ref = F.relu(x.double() @ w.double().T + bias.double())
    exp = torch.full((M + 40, N), 7.0, device=dev, dtype=torch.float64)
    keep = row_map >= 0
    exp[row_map[keep].long()] = ref[keep]
    _close(out, exp, 2e-5, "gemm_f32 row_map/views")
    assert torch.equal(out_b[:, :3], torch.full((M + 40, 3), 7.0, device=dev))


def _attn_ref(q, k, v, scale, causal=False, q_pos0=0, relh=None, relw=None, S=0):
    q, k, v = q.double(), k.double(), v.double()
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    B, H, Nq, Nk = s.shape
    if relh is not None:
        kk = torch.arange(Nk, device=q.device)
        bias = relh.double()[:, :, kk // S] + relw.double()[:, :, kk % S]
        s = s + bias.view(B, H, Nq, Nk)
    if causal:
        qi = torch.arange(Nq, device=q.device)[:, None]
        kj = torch.arange(Nk, device=q.device)[None, :]
        s = s.masked_fill(kj > qi + q_pos0, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhqk,bhkd->bhqd", p, v)
    return o.permute(0, 2, 1, 3).reshape(B, Nq, H * q.shape[-1])


ATTN_CASES = [
    # B, H, Nq, Nk, d, causal, S
    (2, 3, 196, 196, 80, False, 14),
    (1, 2, 49, 49, 32, False, 7),
    (1, 2, 4096, 4096, 80, False, 64),
    (2, 4, 257, 257, 64, False, 0),
    (2, 4, 291, 291, 128, True, 0),
    (3, 4, 1, 295, 128, False, 0),
    (2, 3, 1, 1, 128, False, 0),     # decode kernel: single key
    (2, 3, 1, 7, 128, False, 0),     # decode kernel: ragged last group of 4 keys
    (5, 7, 1, 300, 128, True, 0),    # decode through the causal flag (q_pos0 = Nk-1: every key visible)
    (1, 32, 1, 33, 128, False, 0),   # decode kernel, keys split over the 4 waves of a workgroup: the last wave sees none
    (40, 32, 1, 50, 128, False, 0),  # decode kernel, B*H > 1024: one wave per (batch, head)
    (2, 8, 6, 4096, 16, False, 0),
    (2, 8, 4096, 6, 16, False, 0),
    (2, 8, 6, 6, 32, False, 0),
    (1, 4, 40, 40, 16, True, 0),
]


@pytest.mark.parametrize("case", ATTN_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_attention(dev, case, dtype):
    ops = _ops()
    B, H, Nq, Nk, d, causal, S = case
    if dtype == torch.float32 and Nk > 4096:
        pytest.skip("f32 path bounds Nk by LDS")
    # token-major fused layout like the product: [B, N, 3, H, d]
    if Nq == Nk:
        qkv = _rand((B, Nq, 3, H, d), dev, dtype, 20)
        q = qkv[:, :, 0].permute(0, 2, 1, 3)
        k = qkv[:, :, 1].permute(0, 2, 1, 3)
        v = qkv[:, :, 2].permute(0, 2, 1, 3)
    else:
        q = _rand((B, Nq, H, d), dev, dtype, 21).permute(0, 2, 1, 3)
        k = _rand((B, Nk, H, d), dev, dtype, 22).permute(0, 2, 1, 3)
        v = _rand((B, Nk, H, d), dev, dtype, 23).permute(0, 2, 1, 3)
    scale = d ** -0.5
    relh = relw = None
    if S:
        relh = _rand((B * H, Nq, S), dev, torch.float32, 24)
        relw = _rand((B * H, Nq, S), dev, torch.float32, 25)
    q_pos0 = Nk - Nq
    got = ops.attention(q, k, v, scale, causal=causal, q_pos0=q_pos0, relh=relh, relw=relw, S=S)
    ref = _attn_ref(q, k, v, scale, causal, q_pos0, relh, relw, S)
    _close(got, ref, 2e-2 if dtype == torch.bfloat16 else 2e-5, f"attention {case} {dtype}")


@pytest.mark.parametrize("B,H", [(3, 4), (1, 32), (40, 32)])
def test_decode_attention_rope_fused_is_bit_identical(dev, B, H):
    """haff_decode_attention_rope_rows_bf16 (RoPE of q and the new k, KV-cache append and the attention in one launch) ==
    haff_rope_cache_rows followed by haff_attention_decode_rows_bf16, bit for bit: outputs AND the caches, ragged positions
    (a row at position 0 included), both wave layouts of the decode kernel (B*H <= 1024 and above)."""
    ops = _ops()
    d, Tmax = 128, 70
    g = torch.Generator(device=dev).manual_seed(B * 100 + H)
    qkv = torch.randn((B, 3 * H * d), generator=g, device=dev).to(torch.bfloat16)
    kc = torch.randn((B, Tmax, H * d), generator=g, device=dev).to(torch.bfloat16)
    vc = torch.randn((B, Tmax, H * d), generator=g, device=dev).to(torch.bfloat16)
    pos = torch.randint(0, Tmax, (B,), generator=g, device=dev).to(torch.int32)
    pos[0] = 0
    pos[-1] = Tmax - 1
    nk = pos + 1
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    ang = torch.arange(Tmax, dtype=torch.float32)[:, None] * inv[None, :]
    cs = torch.cat([ang.cos(), ang.sin()], 1).contiguous().to(dev)
    # two-kernel path
    q1, k1, v1 = qkv.clone(), kc.clone(), vc.clone()
    ops.rope_cache_rows(q1, k1, v1, cs, B, 1, H, H, d, pos)
    ref = ops.attention_decode_rows(q1.view(B, 1, 3, H, d)[:, :, 0].permute(0, 2, 1, 3), k1.view(B, Tmax, H, d).permute(0, 2, 1, 3),
                                    v1.view(B, Tmax, H, d).permute(0, 2, 1, 3), d ** -0.5, nk)
    # fused
    k2, v2 = kc.clone(), vc.clone()
    got = ops.decode_attention_rope(qkv.clone(), k2, v2, cs, H, d, d ** -0.5, nk)
    assert torch.equal(got, ref)
    assert torch.equal(k2, k1) and torch.equal(v2, v1)
    # and the pair agrees with the formula (fp64)
    qf = q1.view(B, 3, H, d)[:, 0].double()
    for b in (0, B - 1):
        n = int(nk[b])
        kk, vv = k1[b, :n].view(n, H, d).double(), v1[b, :n].view(n, H, d).double()
        sc = torch.einsum("hd,nhd->hn", qf[b], kk) * d ** -0.5
        o = torch.einsum("hn,nhd->hd", torch.softmax(sc, -1), vv).reshape(-1)
        _close(got[b, 0], o, 2e-2, "fused decode attention vs formula")


def test_attention_online_softmax_rescale(dev):
    """Force the running max to jump in a late KV tile (cdna guide rule 26)."""
    ops = _ops()
    B, H, N, d = 1, 2, 512, 64
    q = _rand((B, N, H, d), dev, torch.bfloat16, 30).permute(0, 2, 1, 3).contiguous()
    k = _rand((B, N, H, d), dev, torch.bfloat16, 31).permute(0, 2, 1, 3).contiguous()
    v = _rand((B, N, H, d), dev, torch.bfloat16, 32).permute(0, 2, 1, 3).contiguous()
    k[:, :, 300] = q[:, :, 17] * 4.0  # spike for query 17 at key 300 (tile 4)
    got = ops.attention(q, k, v, 0.5)
    ref = _attn_ref(q, k, v, 0.5)
    _close(got, ref, 2e-2, "attention rescale spike")


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 16, 4096, 4096), (1, 3, 4096, 4096), (2, 4, 256, 128), (1, 8, 512, 4096), (3, 1, 768, 192)])
def test_global_attention_pingpong_kernel(dev, B, H, Nq, Nk):
    """attn_global_pp_kernel (d = 80, S = 64, 256-query workgroups, whole tiles, k|v fused rows): both grid decodes
    (B*H a multiple of 8 or not), Nq != Nk, the shortest K/V pipelines (2 and 3 tiles), a late score spike that forces the lazy
    rescale in BOTH wave groups, and rel-pos terms large enough to matter. Against fp64 on the same bf16 inputs."""
    ops = _ops()
    d, S = 80, 64
    if Nq == Nk:
        qkv = _rand((B, Nq, 3, H, d), dev, torch.bfloat16, 70)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    else:
        q = _rand((B, Nq, H, d), dev, torch.bfloat16, 71).permute(0, 2, 1, 3)
        kv = _rand((B, Nk, 2, H, d), dev, torch.bfloat16, 72)
        k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)
    # spikes: query 5 (group 0's first wave) and query 200 (group 1) meet a key that dominates late in the sequence. x5: the
    # score lands ~2^60 above the first tile's reference — past the kernel's lazy-reference bound, so the rare arm (cross-lane
    # maximum, accumulator rescale, exponentials redone) runs; x3 stays below it: probabilities far above 1 without a rescale
    k[:, :, Nk - 70] = q[:, :, 5] * 5.0
    k[:, :, Nk - 3] = q[:, :, 200] * 3.0
    relh = _rand((B * H, Nq, S), dev, torch.float32, 73, 2.0)
    relw = _rand((B * H, Nq, S), dev, torch.float32, 74, 2.0)
    got = ops.attention(q, k, v, d ** -0.5, relh=relh, relw=relw, S=S)
    ref = _attn_ref(q, k, v, d ** -0.5, False, 0, relh, relw, S)
    assert torch.isfinite(got.float()).all()
    _close(got, ref, 2e-2, f"global attention ping-pong B={B} H={H} Nq={Nq} Nk={Nk}")
    # poisoned LDS neighbours must not leak in: the same launch after a kernel that left NaN patterns in LDS-sized scratch is
    # covered by running twice around other work and demanding identical bits
    ops.layernorm(_rand((512, 1280), dev, torch.bfloat16, 75), torch.ones(1280, device=dev), torch.zeros(1280, device=dev), 1e-6)
    assert torch.equal(got, ops.attention(q, k, v, d ** -0.5, relh=relh, relw=relw, S=S))


@pytest.mark.parametrize("B,T,H,K,pos0", [(4, 291, 8, 1024, 0), (3, 350, 4, 512, 7), (2, 256, 2, 256, 0)])
def test_qkv_rope_gemm(dev, B, T, H, K, pos0):
    """haff_gemm_bf16_qkv_rope (q|k|v projection + rotate-half RoPE + KV-cache append in one epilogue, weights row-permuted
    inside 256-row tiles) against haff_gemm_bf16 + haff_rope_cache: v rows bit-identical (no arithmetic behind the product),
    rotated q / k equal up to ONE bf16 rounding (the fused form rotates the fp32 accumulators, the pair rounds the product
    first), and both against fp64. Ragged last M-tile (B*T not a multiple of 256), pos0 > 0, cache rows outside the new
    positions untouched."""
    ops = _ops()
    d, Tmax = 128, T + pos0 + 5
    hd = H * d
    x = _rand((B * T, K), dev, torch.bfloat16, 100)
    w = _rand((3 * hd, K), dev, torch.bfloat16, 101, K ** -0.5)
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    ang = torch.arange(Tmax, dtype=torch.float32)[:, None] * inv[None, :]
    cs = torch.cat([ang.cos(), ang.sin()], 1).contiguous().to(dev)
    fill = torch.full((B, Tmax, hd), 7.0, dtype=torch.bfloat16, device=dev)
    # the two-kernel path
    qkv = ops.linear(x, w)
    k1, v1 = fill.clone(), fill.clone()
    ops.rope_cache(qkv, k1, v1, cs, B, T, H, H, d, pos0)
    q1 = qkv[:, :hd]
    # fused
    k2, v2 = fill.clone(), fill.clone()
    q2 = ops.qkv_rope(x, ops.rope_permute_rows(w), k2, v2, cs, B, T, H, d, pos0)
    assert torch.equal(v2, v1), "v rows / untouched cache rows differ"
    new = slice(pos0, pos0 + T)
    assert torch.equal(k2[:, :pos0], k1[:, :pos0]) and torch.equal(k2[:, pos0 + T:], k1[:, pos0 + T:])
    # fp64 reference of the rotation on the fp64 product
    y = x.double() @ w.double().t()
    y = y.view(B, T, 3, H, d)
    c = cs[pos0:pos0 + T, :64].double()[None, :, None, :]
    s_ = cs[pos0:pos0 + T, 64:].double()[None, :, None, :]

    def rot(u):
        u1, u2 = u[..., :64], u[..., 64:]
        return torch.cat([u1 * c - u2 * s_, u2 * c + u1 * s_], -1)
    qr, kr = rot(y[:, :, 0]).reshape(B * T, hd), rot(y[:, :, 1]).reshape(B, T, hd)
    _close(q2, qr, 1e-2, "fused q vs fp64")
    _close(k2[:, new], kr, 1e-2, "fused k vs fp64")
    _close(q2, q1, 2.0 ** -6, "fused q vs two-kernel path")
    _close(k2[:, new], k1[:, new], 2.0 ** -6, "fused k vs two-kernel path")
    # and it is at least as close to fp64 as the pair (one rounding less)
    assert (q2.double() - qr).abs().mean().item() <= (q1.double() - qr).abs().mean().item() * 1.05
    k3, v3 = fill.clone(), fill.clone()
    q3 = ops.qkv_rope(x, ops.rope_permute_rows(w), k3, v3, cs, B, T, H, d, pos0)
    assert torch.equal(q3, q2) and torch.equal(k3, k2) and torch.equal(v3, v2)


@pytest.mark.parametrize("M,N,K,inplace", [(4136, 4096, 128, False), (16448, 1024, 256, True), (4136, 256, 128, False)])
def test_linear_row_tail_split(dev, M, N, K, inplace):
    """linear() sends a <= 64-row tail that would cost the persistent 256 x 256 tile an extra round out as its own launch (CLIP at
    64 frames: 16448 rows): the two-launch form equals the one-launch form (whole row tiles bit for bit on the same tile, the
    tail rows to bf16 rounding: the weight-streaming kernel sums K in another order), bias + activation + in-place residual
    included; shapes where the tail adds no round stay one launch."""
    ops = _ops()
    import haff.ops as hops
    x = _rand((M, K), dev, torch.bfloat16, 70)
    w = _rand((N, K), dev, torch.bfloat16, 71, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 72)
    resid = _rand((M, N), dev, torch.bfloat16, 73)
    assert hops.SPLIT_ROW_TAIL
    o_split = resid.clone() if inplace else None
    o_split = ops.linear(x, w, bias=bias, act=ops.ACT_QUICK_GELU, resid=o_split if inplace else resid, out=o_split)
    hops.SPLIT_ROW_TAIL = False
    try:
        o_one = ops.linear(x, w, bias=bias, act=ops.ACT_QUICK_GELU, resid=resid)
    finally:
        hops.SPLIT_ROW_TAIL = True
    m0 = M & ~255
    assert torch.equal(o_split[:m0], o_one[:m0])
    _close(o_split[m0:], o_one[m0:], 2.0 ** -7, "tail rows")
    ref = F.linear(x.float(), w.float(), bias)
    ref = ref * torch.sigmoid(1.702 * ref) + resid.float()
    _close(o_split, ref, 2e-2, "two-launch linear vs fp32")


@pytest.mark.parametrize("M,N,K,gather", [(2048, 1280, 1280, False), (1024, 1280, 5120, False), (1536, 1280, 1280, True)])
def test_linear_rowstats(dev, M, N, K, gather):
    """haff_gemm_bf16_rowstats: (1) the product + bias + residual is bit-identical to haff_gemm_bf16 on the 256 x 256 tile
    (in place over the residual, with and without the A-side gather); (2) the {mean, rstd} it hands on equal the row statistics
    of the stored rows (haff_row_stats of the bf16 output: the producer sums its fp32 values before rounding, so agreement is to
    a few 1e-4 relative) and the fp64 definition; (3) repeat launches are bit-identical (slots are summed in order)."""
    ops = _ops()
    import haff.ops as hops
    assert hops.linear_rowstats_supported(131072, N, K, torch.bfloat16) and not hops.linear_rowstats_supported(4096, N, K, torch.bfloat16)
    assert hops.linear_rowstats_supported(8192, N, K, torch.bfloat16)
    x = _rand((M + 300 if gather else M, K), dev, torch.bfloat16, 90)
    w = _rand((N, K), dev, torch.bfloat16, 91, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 92)
    resid = _rand((M, N), dev, torch.bfloat16, 93, 3.0) + 2.0
    a_map = None
    if gather:
        g = torch.Generator(device="cpu").manual_seed(94)
        a_map = torch.randperm(M + 300, generator=g)[:M].to(torch.int32).to(dev)
    ref = ops.linear(x, w, bias=bias, resid=resid, a_map=a_map, tile_cfg=0 if gather else 2)
    lib = hops.load_library()
    out = resid.clone()
    part = torch.empty((M, N // 64, 2), dtype=torch.float32, device=dev)
    stats = torch.empty((M, 2), dtype=torch.float32, device=dev)

    def run():
        out.copy_(resid)
        rc = lib.haff_gemm_bf16_rowstats(x.data_ptr(), x.stride(0), a_map.data_ptr() if gather else None, x.shape[0], w.data_ptr(),
                                         w.stride(0), out.data_ptr(), out.stride(0), bias.data_ptr(), out.data_ptr(), out.stride(0),
                                         M, N, K, part.data_ptr(), None)
        assert rc == 0
        assert lib.haff_row_stats_finalize(part.data_ptr(), stats.data_ptr(), M, N // 64, N, 1e-6, None) == 0
        torch.cuda.synchronize()
        return out.clone(), stats.clone()
    o1, s1 = run()
    if gather:   # (the gather entry picks its tile by shape: same sums, possibly another order)
        _close(o1, ref, 2.0 ** -7, "rowstats product vs haff_gemm_bf16_gather")
    else:
        assert torch.equal(o1, ref), "product differs from haff_gemm_bf16 on the same tile"
    want = ops.row_stats(o1, 1e-6)
    assert (s1[:, 0] - want[:, 0]).abs().max().item() <= 2e-3 * want[:, 0].abs().max().item() + 1e-4
    assert ((s1[:, 1] - want[:, 1]).abs() / want[:, 1]).max().item() <= 2e-3
    o64 = o1.double()
    mean = o64.mean(1)
    rstd = (o64.var(1, unbiased=False) + 1e-6).rsqrt()
    assert (s1[:, 0].double() - mean).abs().max().item() <= 2e-3 * mean.abs().max().item() + 1e-4
    assert ((s1[:, 1].double() - rstd).abs() / rstd).max().item() <= 2e-3
    o2, s2 = run()
    assert torch.equal(o1, o2) and torch.equal(s1, s2)


@pytest.mark.parametrize("M,N,K,gather", [(2048, 1280, 1280, False), (1024, 1280, 5120, False), (1536, 1280, 1280, True)])
def test_linear_rowstats32_fp32_residual_stream(dev, M, N, K, gather):
    """haff_gemm_bf16_rowstats32 (round 6, the fused fp32 residual stream): x32 += A.W^T + bias in place in fp32 — within fp32
    accumulation noise of the fp64 sum, i.e. NOT rounded to bf16; out16 is exactly the bf16 rounding of the stored fp32 rows; the
    {mean, rstd} handed on are those of the fp32 rows; the product part equals haff_gemm_bf16's accumulators (same tile, same K
    loop: x32_new - x32_old == the bf16-free product to fp32 rounding of the add); repeat launches are bit-identical."""
    ops = _ops()
    x = _rand((M + 300 if gather else M, K), dev, torch.bfloat16, 90)
    w = _rand((N, K), dev, torch.bfloat16, 91, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 92)
    x32_0 = _rand((M, N), dev, torch.float32, 93, 3.0) + 2.0
    a_map = None
    if gather:
        g = torch.Generator(device="cpu").manual_seed(94)
        a_map = torch.randperm(M + 300, generator=g)[:M].to(torch.int32).to(dev)
    xa = x if a_map is None else x[a_map.long()]
    exact = x32_0.double() + xa.double() @ w.double().T + bias.double()

    def run():
        x32 = x32_0.clone()
        out16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        st = ops.linear_rowstats32(x, w, bias, x32, out16, 1e-6, a_map=a_map)
        torch.cuda.synchronize()
        return x32, out16, st
    x32, out16, st = run()
    err = (x32.double() - exact).abs().max().item()
    assert err <= 2e-5 * exact.abs().max().item(), err          # fp32 accumulation over K, far below a bf16 ulp (4e-3 relative)
    assert torch.equal(out16, x32.to(torch.bfloat16)), "the bf16 copy is not the RNE rounding of the stored fp32 stream"
    mean = exact.mean(1)
    rstd = (exact.var(1, unbiased=False) + 1e-6).rsqrt()
    assert (st[:, 0].double() - mean).abs().max().item() <= 1e-4 * mean.abs().max().item() + 1e-5
    assert ((st[:, 1].double() - rstd).abs() / rstd).max().item() <= 2e-4
    # the same product through haff_gemm_bf16 with an fp32 residual and fp32 output (the unfused form's epilogue)
    ref32 = ops.linear(x, w, bias=bias, resid=x32_0.clone(), a_map=a_map, out_dtype=torch.float32)
    assert (ref32 - x32).abs().max().item() <= 2e-5 * exact.abs().max().item()
    x32b, out16b, stb = run()
    assert torch.equal(x32, x32b) and torch.equal(out16, out16b) and torch.equal(st, stb)
    # contract: shapes that are not whole 256 x 256 tiles are refused, nothing is written
    import haff.ops as hops
    lib = hops.load_library()
    part = torch.empty((M, N // 64, 2), dtype=torch.float32, device=dev)
    rc = lib.haff_gemm_bf16_rowstats32(x.data_ptr(), x.stride(0), None, 0, w.data_ptr(), w.stride(0), x32.data_ptr(), x32.stride(0),
                                       out16.data_ptr(), out16.stride(0), bias.data_ptr(), M - 16, N, K, part.data_ptr(), None)
    assert rc == -2


@pytest.mark.parametrize("ratio", [30.0, 100.0])
def test_linear_rowstats_with_a_large_row_mean(dev, ratio):
    """the producer's statistics are {sum, sum of squares} of fp32 values, so the variance is a difference of two
    numbers of size mean^2 — rows whose mean is `ratio` standard deviations away from zero lose ~5e-7 * ratio^2 of it. The bound
    held here: rstd within 3e-6 * ratio^2 (relative) of the two-pass haff_row_stats on the stored rows, mean within 1e-5."""
    ops = _ops()
    M, N, K = 2048, 1280, 1280
    x = _rand((M, K), dev, torch.bfloat16, 190)
    w = _rand((N, K), dev, torch.bfloat16, 191, K ** -0.5)
    resid = (_rand((M, N), dev, torch.float32, 192) + ratio).to(torch.bfloat16)     # product ~ N(0, 1), residual ~ N(ratio, 1): std ~ 1.4
    out, stats = ops.linear_rowstats(x, w, None, resid, 1e-6)
    # reference: the statistics of the EXACT rows (fp64 product + residual) — what the producer sums before it rounds to bf16;
    # the stored rows themselves are only good to a bf16 ulp of `ratio` (0.125 .. 0.5), their mean to ~4e-3
    exact = x.double() @ w.double().T + resid.double()
    mean = exact.mean(1)
    rstd = (exact.var(1, unbiased=False) + 1e-6).rsqrt()
    assert (stats[:, 0].double() - mean).abs().max().item() <= 1e-5 * ratio
    rel = ((stats[:, 1].double() - rstd).abs() / rstd).max().item()
    print(f"mean / std = {ratio:g}: rstd relative error {rel:.2e}")
    assert rel <= 3e-6 * ratio * ratio + 1e-5, rel
    assert (out.double() - exact).abs().max().item() <= 2.0 ** -8 * exact.abs().max().item()


@pytest.mark.parametrize("B,H", [(1, 16), (2, 3)])
def test_global_attention_fused_relpos(dev, B, H):
    """haff_global_attention_bf16 (rel_h / rel_w computed in the kernel's prologue from the bf16 parameter tables) against
    (a) fp64 attention with the decomposed bias of image_encoder.py:354-392 on the same bf16 inputs, and (b) the two-kernel path
    (haff_relpos_tables_bf16 + haff_attention_bf16): same bf16 products, fp32 sums in another order -> equal to fp32 rounding of
    the bias, i.e. well inside one bf16 ulp of the output. Table values large enough that a wrong row / flipped index shows."""
    ops = _ops()
    S, d = 64, 80
    N = S * S
    qkv = _rand((B, N, 3, H, d), dev, torch.bfloat16, 80)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    th = _rand((2 * S - 1, d), dev, torch.float32, 81, 0.25)
    tw = _rand((2 * S - 1, d), dev, torch.float32, 82, 0.25)
    assert ops.global_attention_supported(q, k, v, S)
    got = ops.global_attention(q, k, v, d ** -0.5, th, tw, S)
    # (b) two-kernel path
    rh, rw = ops.relpos_tables(q, th, tw, S)
    two = ops.attention(q, k, v, d ** -0.5, relh=rh, relw=rw, S=S)
    # (a) fp64 with the tables as the kernels see them (bf16-rounded), bias from the definition
    thb, twb = th.to(torch.bfloat16).double(), tw.to(torch.bfloat16).double()
    idx = torch.arange(S, device=dev)[:, None] - torch.arange(S, device=dev)[None, :] + (S - 1)     # [q coord][k coord]
    q6 = q.double().reshape(B, H, S, S, d)
    relh = torch.einsum("bhyxc,ykc->bhyxk", q6, thb[idx]).reshape(B * H, N, S)
    relw = torch.einsum("bhyxc,xkc->bhyxk", q6, twb[idx]).reshape(B * H, N, S)
    ref = _attn_ref(q, k, v, d ** -0.5, False, 0, relh, relw, S)
    assert torch.isfinite(got.float()).all()
    _close(got, ref, 2e-2, f"fused global attention B={B} H={H} vs fp64")
    _close(rh.double(), relh, 1e-4, "rel_h tables vs definition")
    assert (got.float() - two.float()).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item(), "fused vs two-kernel path"
    assert torch.equal(got, ops.global_attention(q, k, v, d ** -0.5, th, tw, S)), "repeat launch differs"


@pytest.mark.parametrize("S,d", [(14, 80), (7, 32), (64, 80), (20, 80)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_relpos_tables(dev, dtype, S, d):
    ops = _ops()
    B, H = 2, 3
    N = S * S
    qkv = _rand((B, N, 3, H, d), dev, dtype, 40)
    q = qkv[:, :, 0].permute(0, 2, 1, 3)
    th = _rand((2 * S - 1, d), dev, torch.float32, 41)
    tw = _rand((2 * S - 1, d), dev, torch.float32, 42)
    if dtype == torch.bfloat16:  # the bf16 path multiplies bf16 tables on the MFMA units
        th, tw = th.to(torch.bfloat16).float(), tw.to(torch.bfloat16).float()
    relh, relw = ops.relpos_tables(q, th, tw, S)
    idx = torch.arange(S, device=dev)[:, None] - torch.arange(S, device=dev)[None, :] + (S - 1)
    Rh, Rw = th[idx], tw[idx]  # [S(q), S(k), d]
    rq = q.float().reshape(B * H, S, S, d)
    ref_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh).reshape(B * H, N, S)
    ref_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw).reshape(B * H, N, S)
    _close(relh, ref_h, 1e-5, "relh")
    _close(relw, ref_w, 1e-5, "relw")


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1000, 1280, 1280), (70, 96, 64)])
def test_linear_gather_rows(dev, M, N, K):
    """haff_gemm_bf16_gather: logical row m reads x[a_map[m]]; combined with bias, residual and an output row map."""
    ops = _ops()
    R = M + 57
    x = _rand((R, K), dev, torch.bfloat16, 70)
    w = _rand((N, K), dev, torch.bfloat16, 71, K ** -0.5)
    b = _rand((N,), dev, torch.float32, 72)
    g = torch.Generator(device="cpu").manual_seed(73)
    a_map = torch.randint(0, R, (M,), generator=g).to(torch.int32).to(dev)
    resid = _rand((M, N), dev, torch.bfloat16, 74)
    got = ops.linear(x, w, bias=b, resid=resid, a_map=a_map)
    ref = x.float()[a_map.long()] @ w.float().t() + b + resid.float()
    _close(got, ref, 2e-2, "gather linear")
    # in-place residual, as the SAM proj uses it
    xr = resid.clone()
    ops.linear(x, w, bias=b, resid=xr, a_map=a_map, out=xr)
    _close(xr, ref, 2e-2, "gather linear in place")


def test_window_attention_pad_token(dev):
    """Padded windows: rows of padded tokens are never written (filled with NaN here); the kernel must take the pad
    token row instead. 2 images of 20x20 tokens -> 2x2 windows of 14x14, windows on the right/bottom edge padded."""
    ops = _ops()
    S, d, H, grid, nimg = 14, 80, 2, 20, 2
    N, wps = S * S, 2
    n_win = nimg * wps * wps
    qkv = _rand((n_win * N + 1, 3, H, d), dev, torch.bfloat16, 48, 1.5)
    pad_tok = qkv[-1].clone()
    is_pad = torch.zeros((n_win, S, S), dtype=torch.bool, device=dev)
    for w in range(n_win):
        wy, wx = (w % 4) // 2, (w % 4) % 2
        is_pad[w, max(0, grid - wy * S):, :] = True
        is_pad[w, :, max(0, grid - wx * S):] = True
    full = qkv[:-1].view(n_win, N, 3, H, d).clone()
    full[is_pad.view(n_win, N)] = pad_tok          # what window_partition's zero pad + qkv bias would hold
    holes = qkv.clone()
    holes[:-1].view(n_win, N, 3, H, d)[is_pad.view(n_win, N)] = float("nan")
    th = _rand((2 * S - 1, d), dev, torch.float32, 46, 0.5).to(torch.bfloat16).float()
    tw = _rand((2 * S - 1, d), dev, torch.float32, 47, 0.5).to(torch.bfloat16).float()

    def views(buf):
        q5 = buf.view(n_win, N, 3, H, d)
        return (q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    q, k, v = views(holes[:-1])
    got = ops.window_attention(q, k, v, d ** -0.5, th, tw, S, grid=grid, pad_token=n_win * N)
    qf, kf, vf = views(full)
    ref = ops.window_attention(qf, kf, vf, d ** -0.5, th, tw, S)
    real = ~is_pad.view(n_win, N)
    assert torch.isfinite(got[real]).all()
    assert torch.equal(got[real], ref[real])


@pytest.mark.parametrize("n_win", [8, 5, 48])
def test_window_attention_fused(dev, n_win):
    """haff_window_attention_bf16 (rel-pos computed in the kernel) vs the reference formula
    (image_encoder.py:235-260,354-392) on the ViT-H window geometry; both grid->workgroup mappings (n_win % 8)."""
    ops = _ops()
    S, d, H = 14, 80, 3
    N = S * S
    qkv = _rand((n_win, N, 3, H, d), dev, torch.bfloat16, 45, 1.5)
    q = qkv[:, :, 0].permute(0, 2, 1, 3)
    k = qkv[:, :, 1].permute(0, 2, 1, 3)
    v = qkv[:, :, 2].permute(0, 2, 1, 3)
    th = _rand((2 * S - 1, d), dev, torch.float32, 46, 0.5).to(torch.bfloat16).float()
    tw = _rand((2 * S - 1, d), dev, torch.float32, 47, 0.5).to(torch.bfloat16).float()
    assert ops.window_attention_supported(q, S)
    scale = d ** -0.5
    got = ops.window_attention(q, k, v, scale, th, tw, S)
    idx = torch.arange(S, device=dev)[:, None] - torch.arange(S, device=dev)[None, :] + (S - 1)
    rq = q.float().reshape(n_win * H, S, S, d)
    relh = torch.einsum("bhwc,hkc->bhwk", rq, th[idx]).reshape(n_win * H, N, S)
    relw = torch.einsum("bhwc,wkc->bhwk", rq, tw[idx]).reshape(n_win * H, N, S)
    ref = _attn_ref(q, k, v, scale, False, 0, relh, relw, S)
    _close(got, ref, 2e-2, f"fused window attention n_win={n_win}")
    # and against the generic pair it replaces
    rh, rw = ops.relpos_tables(q, th, tw, S)
    old = ops.attention(q, k, v, scale, relh=rh, relw=rw, S=S)
    _close(got, old.float(), 2e-2, "fused vs generic window attention")


@pytest.mark.parametrize("C", [64, 256, 1024, 1280, 2048, 4096, 5120, 6144])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_norms(dev, C, dtype):
    ops = _ops()
    R = 77
    x = _rand((R, C), dev, dtype, 50, 2.0) + 0.3
    w = _rand((C,), dev, torch.float32, 51) + 1.0
    b = _rand((C,), dev, torch.float32, 52)
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    got = ops.layernorm(x, w, b, 1e-6)
    _close(got, F.layer_norm(x.float(), (C,), w, b, 1e-6), tol, "layernorm")
    xf = x.float()
    ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * w
    _close(ops.rmsnorm(x, w, 1e-5), ref, tol, "rmsnorm")
    m = torch.randint(0, R, (100,), device=dev).to(torch.int32)
    m[::5] = -1
    got = ops.layernorm(x, w, b, 1e-6, in_map=m)
    ref = F.layer_norm(xf, (C,), w, b, 1e-6)[m.clamp(min=0).long()]
    ref[m < 0] = 0
    _close(got, ref, tol, "layernorm gather")
    assert (got[m < 0] == 0).all()


@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("M,K,H,d", [(512, 128, 16, 80), (1024, 1280, 16, 80), (256, 64, 4, 64)])
def test_linear_heads_scatter(dev, M, K, H, d, fold):
    """haff_gemm_bf16_heads: the product's columns part * H * d + h * d + c of row m land at planes[part][w][h][t][c] with
    row_map[m] = w * H * n_tok + t — bit-identical to the token-major product (same kernel, other store addresses), dropped rows
    (-1) and unaddressed slots untouched."""
    ops = _ops()
    ntok = 196
    nwin = (M + ntok - 1) // ntok + 1
    N = 3 * H * d
    x = _rand((M, K), dev, torch.bfloat16, 70)
    w = _rand((N, K), dev, torch.bfloat16, 71, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 72)
    g = torch.Generator().manual_seed(73)
    slots = torch.randperm(nwin * ntok, generator=g)[:M]
    keep = torch.rand((M,), generator=g) > 0.1
    rmap = torch.where(keep, (slots // ntok) * (H * ntok) + slots % ntok, torch.full((M,), -1)).to(torch.int32).to(dev)
    kw = {}
    if fold:
        kw = dict(ln_stats=torch.stack([_rand((M,), dev, torch.float32, 74), _rand((M,), dev, torch.float32, 75).abs() + 0.5], 1).contiguous(),
                  ln_colsum=_rand((N,), dev, torch.float32, 76))
    assert ops.linear_heads_supported(M, N, K, d, H, torch.bfloat16)
    planes = torch.full((3, nwin + 1, H, ntok, d), 7.0, dtype=torch.bfloat16, device=dev)
    ops.linear_heads(x, w, bias, rmap, planes, d, H, (nwin + 1) * H * ntok * d, ntok * d, **kw)
    ref = ops.linear(x, w, bias=bias, tile_cfg=0 if fold else 2, **kw).view(M, 3, H, d)
    exp = torch.full_like(planes, 7.0)
    mk = keep.to(dev)
    wi, ti = (slots // ntok).to(dev)[mk], (slots % ntok).to(dev)[mk]
    for part in range(3):
        exp[part, wi, :, ti] = ref[mk, part]
    assert torch.equal(planes, exp)


@pytest.mark.parametrize("C,R", [(1280, 77), (1280, 8203), (4096, 5), (4096, 600), (5120, 3)])
def test_norms_f32_rows_to_bf16(dev, C, R):
    """dtype 2 of haff_layernorm / haff_rmsnorm: an fp32 residual stream normalised in fp32 and rounded to bf16 ONCE — bit-equal
    to the fp32 kernel's result rounded to bf16 (both row kernels: one wave per row, one workgroup per row)."""
    ops = _ops()
    x = _rand((R, C), dev, torch.float32, 56, 2.0) + 0.3
    w = _rand((C,), dev, torch.float32, 57) + 1.0
    b = _rand((C,), dev, torch.float32, 58)
    got = ops.layernorm(x, w, b, 1e-6, out_dtype=torch.bfloat16)
    assert got.dtype == torch.bfloat16
    assert torch.equal(got, ops.layernorm(x, w, b, 1e-6).to(torch.bfloat16))
    _close(got, F.layer_norm(x, (C,), w, b, 1e-6), 1e-2, "layernorm f32 -> bf16")
    got = ops.rmsnorm(x, w, 1e-5, out_dtype=torch.bfloat16)
    assert got.dtype == torch.bfloat16 and torch.equal(got, ops.rmsnorm(x, w, 1e-5).to(torch.bfloat16))
    m = torch.randint(0, R, (R + 9,), device=dev).to(torch.int32)
    m[::5] = -1
    got = ops.layernorm(x, w, b, 1e-6, in_map=m, out_dtype=torch.bfloat16)
    assert torch.equal(got, ops.layernorm(x, w, b, 1e-6, in_map=m).to(torch.bfloat16)) and (got[m < 0] == 0).all()


@pytest.mark.parametrize("C", [256, 1024, 1280])
def test_norms_many_rows(dev, C):
    """>= 8192 rows take the 4-rows-per-wave kernel: ragged row count, gather map with dropped rows."""
    ops = _ops()
    R = 8203
    x = _rand((R, C), dev, torch.bfloat16, 53, 2.0) + 0.3
    w = _rand((C,), dev, torch.float32, 54) + 1.0
    b = _rand((C,), dev, torch.float32, 55)
    xf = x.float()
    _close(ops.layernorm(x, w, b, 1e-6), F.layer_norm(xf, (C,), w, b, 1e-6), 1e-2, "layernorm many rows")
    ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * w
    _close(ops.rmsnorm(x, w, 1e-5), ref, 1e-2, "rmsnorm many rows")
    m = torch.randint(0, R, (9001,), device=dev).to(torch.int32)
    m[::7] = -1
    got = ops.layernorm(x, w, b, 1e-6, in_map=m)
    ref = F.layer_norm(xf, (C,), w, b, 1e-6)[m.clamp(min=0).long()]
    ref[m < 0] = 0
    _close(got, ref, 1e-2, "layernorm gather many rows")
    assert (got[m < 0] == 0).all()


def test_patchify_and_im2col(dev):
    ops = _ops()
    B, P, g = 2, 16, 5
    x = _rand((B, 3, g * P, g * P), dev, torch.float32, 60)
    w = _rand((32, 3, P, P), dev, torch.float32, 61, 0.05)
    rows = ops.patchify_nchw(x, P, g, g, 3 * P * P, torch.float32)
    got = (rows @ w.reshape(32, -1).T).view(B, g, g, 32)
    ref = F.conv2d(x, w, stride=P).permute(0, 2, 3, 1)
    _close(got, ref, 1e-5, "patchify conv")
    # CLIP-style 14x14 patches with K padded to a multiple of 8
    xc = _rand((B, 3, 28, 28), dev, torch.bfloat16, 62)
    rows = ops.patchify_nchw(xc, 14, 2, 2, 592, torch.bfloat16)
    ref = F.unfold(xc.float(), 14, stride=14).transpose(1, 2).reshape(B * 4, 588)
    assert torch.equal(rows[:, :588].float(), ref) and (rows[:, 588:] == 0).all()
    # uint8 NHWC ingest with SAM normalisation and zero pad
    fr = torch.randint(0, 256, (B, 70, 60, 3), dtype=torch.uint8, device=dev)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    rows = ops.patchify_u8(fr, P, g, g, 3 * P * P, mean, std, torch.float32)
    xn = (fr.float().permute(0, 3, 1, 2) - torch.tensor(mean, device=dev).view(1, 3, 1, 1)) / torch.tensor(std, device=dev).view(1, 3, 1, 1)
    xn = F.pad(xn, (0, g * P - 60, 0, g * P - 70))
    ref = ops.patchify_nchw(xn, P, g, g, 3 * P * P, torch.float32)
    _close(rows, ref, 1e-6, "patchify_u8")
    # 3x3 im2col, channels-last
    xi = _rand((B, 6, 7, 16), dev, torch.float32, 63)
    wi = _rand((8, 16, 3, 3), dev, torch.float32, 64, 0.1)
    cols = ops.im2col3x3(xi)
    got = (cols @ wi.permute(0, 2, 3, 1).reshape(8, -1).T).view(B, 6, 7, 8)
    ref = F.conv2d(xi.permute(0, 3, 1, 2), wi, padding=1).permute(0, 2, 3, 1)
    _close(got, ref, 1e-5, "im2col3x3 conv")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_embed_splice_rope_argmax(dev, dtype):
    ops = _ops()
    B, L, n_img, Hd, V = 3, 12, 16, 64, 50
    ids = torch.randint(0, V, (B, L), device=dev)
    pos = torch.tensor([2, 2, 5], dtype=torch.int32, device=dev)
    for b in range(B):
        ids[b, pos[b]] = -200
    emb = _rand((V, Hd), dev, dtype, 70)
    img = _rand((B, n_img, Hd), dev, dtype, 71)
    got = ops.embed_splice(ids, pos, emb, img)
    for b in range(B):
        p = int(pos[b])
        ref = torch.cat([emb[ids[b, :p]], img[b], emb[ids[b, p + 1:]]], 0)
        assert torch.equal(got[b], ref)
    # rope
    Bq, Tq, Hq, d, pos0, Tmax = 2, 5, 4, 32, 3, 16
    qkv = _rand((Bq * Tq, 3 * Hq * d), dev, dtype, 72)
    orig = qkv.clone().float().view(Bq, Tq, 3, Hq, d)
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, device=dev).float() / d))
    ang = torch.arange(Tmax, device=dev).float()[:, None] * inv[None, :]
    cs = torch.cat([ang.cos(), ang.sin()], 1).contiguous()
    kc = torch.zeros((Bq, Tmax, Hq * d), dtype=dtype, device=dev)
    vc = torch.zeros_like(kc)
    ops.rope_cache(qkv, kc, vc, cs, Bq, Tq, Hq, Hq, d, pos0)
    cos = torch.cat([ang.cos(), ang.cos()], 1)[pos0:pos0 + Tq].view(1, Tq, 1, d)
    sin = torch.cat([ang.sin(), ang.sin()], 1)[pos0:pos0 + Tq].view(1, Tq, 1, d)

    def rot(x):
        return torch.cat([-x[..., d // 2:], x[..., :d // 2]], -1)
    qr = orig[:, :, 0] * cos + rot(orig[:, :, 0]) * sin
    kr = orig[:, :, 1] * cos + rot(orig[:, :, 1]) * sin
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-6
    out = qkv.float().view(Bq, Tq, 3, Hq, d)
    _close(out[:, :, 0], qr, tol, "rope q")
    _close(out[:, :, 1], kr, tol, "rope k")
    _close(kc.view(Bq, Tmax, Hq, d)[:, pos0:pos0 + Tq], kr, tol, "k cache")
    assert torch.equal(vc.view(Bq, Tmax, Hq, d)[:, pos0:pos0 + Tq].float(), orig[:, :, 2])
    assert (kc[:, :pos0] == 0).all() and (kc[:, pos0 + Tq:] == 0).all()
    # argmax (first index on ties)
    lg = _rand((5, 32003), dev, torch.float32, 73)
    lg[2, 100] = lg[2, 31999] = 50.0
    got = ops.argmax_rows(lg)
    assert torch.equal(got, lg.argmax(-1)) or int(got[2]) == 100
    assert int(got[2]) == 100


def test_add_softmax(dev):
    ops = _ops()
    a = _rand((4096 * 2, 256), dev, torch.bfloat16, 80)
    pe = _rand((4096, 256), dev, torch.bfloat16, 81)
    _close(ops.add_bcast(a, pe), a.float() + pe.float().repeat(2, 1), 1e-2, "add_bcast")
    x = _rand((7, 4), dev, torch.float32, 82)
    _close(ops.softmax_rows(x), torch.softmax(x, -1), 1e-6, "softmax")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_upscale_mask_and_resize(dev, dtype):
    ops = _ops()
    n, h, w = 2, 14, 14
    src = _rand((n, 256, h, w), dev, torch.float32, 90)
    ct1 = torch.nn.ConvTranspose2d(256, 64, 2, 2).to(dev).float()
    ct2 = torch.nn.ConvTranspose2d(64, 32, 2, 2).to(dev).float()
    ln_w = _rand((64,), dev, torch.float32, 91) + 1.0
    ln_b = _rand((64,), dev, torch.float32, 92)
    hyper = _rand((n, 32), dev, torch.float32, 93)
    with torch.no_grad():
        u = ct1(src)
        mu = u.mean(1, keepdim=True)
        var = (u - mu).pow(2).mean(1, keepdim=True)
        u = (u - mu) / torch.sqrt(var + 1e-6) * ln_w[:, None, None] + ln_b[:, None, None]
        u = F.gelu(ct2(F.gelu(u)))
        ref = torch.einsum("nc,nchw->nhw", hyper, u)
        # product path: first ConvT as GEMM with columns (dy,dx,co)
        w1 = ct1.weight.permute(2, 3, 1, 0).reshape(256, 256).contiguous()  # [(dy,dx,co), ci]
        b1 = ct1.bias.repeat(4).contiguous()
        x = src.permute(0, 2, 3, 1).reshape(n * h * w, 256).contiguous().to(dtype)
        up1 = ops.linear(x, w1.to(dtype), bias=b1)
        w2 = ct2.weight.permute(0, 2, 3, 1).reshape(64, 128).contiguous()  # [co][(dy2,dx2,c2)]
        got = ops.upscale_mask(up1, ln_w, ln_b, w2, ct2.bias.contiguous(), hyper, n, h, w)
    _close(got, ref, 3e-2 if dtype == torch.bfloat16 else 1e-4, "upscale_mask")
    m = _rand((3, 56, 56), dev, torch.float32, 94)
    up = ops.resize_bilinear(m, (56, 56), (224, 224))
    _close(up, F.interpolate(m[:, None], (224, 224), mode="bilinear", align_corners=False)[:, 0], 1e-6, "resize x4")
    crop = ops.resize_bilinear(up, (224, 168), (120, 90))
    ref = F.interpolate(up[:, None, :224, :168], (120, 90), mode="bilinear", align_corners=False)[:, 0]
    _close(crop, ref, 1e-6, "resize crop")
    th = ops.threshold_masks(crop, 0.0)
    assert torch.equal(th, ((crop > 0).to(torch.uint8) * 255))


@pytest.mark.parametrize("tile_cfg", [1, 2])
@pytest.mark.parametrize("M", [592, 4400])
def test_gemm_stable_beside_second_stream(dev, tile_cfg, M):
    """Regression for the LDS staging races of the tile loops: a GEMM on structured operands (K-tile kt of
    A holds kt + 1, W is all ones: the fp32 result is exact; a fragment that still holds K-tile kt-2 lowers an output by a
    multiple of 16, one that already holds K-tile kt+2 raises it) launched repeatedly while (LayerNorm, qkv GEMM) pairs
    run on another HIP stream must return the same bits every time. Round 1's 128x128 loop failed 5-10 % of its launches
    here (a write-after-read race: tools/vmcnt_forensics.py). M = 4400: 288 tiles of 256^2 on 256 workgroups, so the
    persistent ring loop crosses a tile boundary with its requests in flight."""
    ops = _ops()
    k = torch.arange(4096, device=dev)
    a = (1 + k // 64).to(torch.bfloat16)[None, :].expand(M, 4096).contiguous()
    w = torch.ones(4096, 4096, dtype=torch.bfloat16, device=dev)
    x = _rand((9800, 1280), dev, torch.bfloat16, 31)
    wq = _rand((3840, 1280), dev, torch.bfloat16, 32, 1280 ** -0.5)
    lw, lb = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
    ref = ops.linear(a, w, tile_cfg=tile_cfg, out_dtype=torch.float32).clone()
    assert float(ref.min()) == float(ref.max()) == 64.0 * (64 * 65 // 2)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(60):
            ops.linear(ops.layernorm(x, lw, lb, 1e-6), wq)
    n_rep = 200 if M < 1000 else 60
    outs = [ops.linear(a, w, tile_cfg=tile_cfg, out_dtype=torch.float32) for _ in range(n_rep)]
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(o, ref)) for o in outs)
    assert bad == 0, f"{bad}/{n_rep} launches differ with a second stream active"


@pytest.mark.parametrize("M,N,K", [(4400, 4100, 128), (4352, 4096, 192), (8192, 2304, 320), (70000, 1280, 1280), (2100, 33000, 256)])
def test_gemm_ring_loop_across_tiles(dev, M, N, K):
    """The persistent ping-pong ring loop of the 8-wave tile (tile_cfg 2): more than 256 tiles, so workgroups walk 2+ tiles
    and the K-tile stream runs across the tile boundary (K-tile 0 of the next tile is requested from inside the K loop);
    short K loops (2, 3, 5 K-tiles: every prologue / tail case of the request schedule), a gathered A operand, a row map
    and a residual. Against the fp32 product and bit-for-bit against the 128x128 tile, whose loop shares nothing with it."""
    ops = _ops()
    x = _rand((M, K), dev, torch.bfloat16, 21)
    w = _rand((N, K), dev, torch.bfloat16, 22, K ** -0.5)
    bias = _rand((N,), dev, torch.float32, 23)
    y = x.float() @ w.float().T + bias
    got = ops.linear(x, w, bias=bias, out_dtype=torch.float32, tile_cfg=2)
    _close(got, y, 2e-3, "ring loop f32 out")
    assert torch.equal(got, ops.linear(x, w, bias=bias, out_dtype=torch.float32, tile_cfg=1)), "ring loop vs 128x128 tile"
    if M * N < 100_000_000:
        resid = _rand((M, N), dev, torch.bfloat16, 24)
        amap = torch.randperm(M, device=dev).to(torch.int32)
        rmap = torch.randperm(M, device=dev).to(torch.int32)
        out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
        ops.linear(x, w, bias=bias, resid=resid, a_map=amap, row_map=rmap, out=out, tile_cfg=2)
        ref = torch.zeros((M, N), dtype=torch.float32, device=dev)
        ref[rmap.long()] = y[amap.long()] + resid.float()[rmap.long()]
        _close(out, ref, 1.2e-2, "ring loop gather + row map + residual")


def test_kernels_stable_beside_second_stream(dev):
    """Every hot kernel, launched repeatedly while (LayerNorm, qkv GEMM) pairs run on another HIP stream, must return the
    same bits as alone on the GPU. Memory contention changes the order in which a kernel's loads complete; a kernel
    that depended on that order (the 128x128 GEMM tile once did: DESIGN.md 10a) fails here."""
    ops = _ops()
    x = _rand((9800, 1280), dev, torch.bfloat16, 51)
    wq = _rand((3840, 1280), dev, torch.bfloat16, 52, 1280 ** -0.5)
    lw, lb = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
    qkv_w = _rand((50 * 196 + 1, 3840), dev, torch.bfloat16, 53, 0.5)
    qkv_g = _rand((2 * 4096, 3840), dev, torch.bfloat16, 54, 0.5)
    t14 = (_rand((27, 80), dev, torch.float32, 55, 0.1), _rand((27, 80), dev, torch.float32, 56, 0.1))
    t64 = (_rand((127, 80), dev, torch.float32, 57, 0.1), _rand((127, 80), dev, torch.float32, 58, 0.1))
    qp = _rand((2, 291, 3, 32, 128), dev, torch.bfloat16, 59, 0.5)
    kc = _rand((8, 300, 32, 128), dev, torch.bfloat16, 60, 0.5)
    xs, ws = _rand((64, 4096), dev, torch.bfloat16, 61), _rand((4096, 4096), dev, torch.bfloat16, 62, 4096 ** -0.5)
    xb, wb = _rand((4096, 1280), dev, torch.bfloat16, 63), _rand((5120, 1280), dev, torch.bfloat16, 64, 1280 ** -0.5)
    bias = _rand((5120,), dev, torch.float32, 65)

    def win():
        v = qkv_w[:50 * 196].view(50, 196, 3, 16, 80).permute(2, 0, 3, 1, 4)
        return ops.window_attention(v[0], v[1], v[2], 80 ** -0.5, t14[0], t14[1], 14)

    def glob():
        v = qkv_g.view(2, 4096, 3, 16, 80).permute(2, 0, 3, 1, 4)
        rh, rw = ops.relpos_tables(v[0], t64[0], t64[1], 64)
        return ops.attention(v[0], v[1], v[2], 80 ** -0.5, relh=rh, relw=rw, S=64)

    def prefill():
        v = qp.permute(2, 0, 3, 1, 4)
        return ops.attention(v[0], v[1], v[2], 128 ** -0.5, causal=True)

    def decode():
        q = kc[:, :1].permute(0, 2, 1, 3)
        k = kc.permute(0, 2, 1, 3)
        return ops.attention(q, k, k, 128 ** -0.5)

    def glob_fused():
        v = qkv_g.view(2, 4096, 3, 16, 80).permute(2, 0, 3, 1, 4)
        return ops.global_attention(v[0], v[1], v[2], 80 ** -0.5, t64[0], t64[1], 64)

    cases = {"window attention": win, "global attention + rel-pos tables": glob, "fused global attention": glob_fused,
             "causal prefill attention": prefill,
             "decode attention": decode, "weight-streaming GEMM M=64": lambda: ops.linear(xs, ws),
             "weight-streaming GEMM M=8": lambda: ops.linear(xs[:8], ws, resid=xs[:8]),
             "256x256 GEMM + GELU": lambda: ops.linear(xb, wb, bias=bias, act=1),
             "layernorm": lambda: ops.layernorm(xb, lw, lb, 1e-6), "rmsnorm": lambda: ops.rmsnorm(xs, torch.ones(4096, device=dev), 1e-5)}
    side = torch.cuda.Stream(dev)
    for name, f in cases.items():
        ref = f().clone()
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(30):
                ops.linear(ops.layernorm(x, lw, lb, 1e-6), wq)
        outs = [f() for _ in range(40)]
        torch.cuda.synchronize()
        bad = sum(int(not torch.equal(o, ref)) for o in outs)
        assert bad == 0, f"{name}: {bad}/40 launches differ with a second stream active"


def test_decode_book_kernel(dev):
    """haff_decode_book against the host loop it replaces (generate()'s per-token bookkeeping): forced vs greedy token,
    finished rows emit pad, EOS sets the flag, out_ids / hidden rows land at each row's own position, positions advance."""
    ops = _ops()
    B, H, tmax, pad, eos = 5, 64, 24, 0, 2
    i64, i32 = torch.int64, torch.int32
    lens = torch.tensor([4, 7, 5, 6, 3], dtype=i64, device=dev)
    t_rows = (lens + 9).to(i32)
    for use_forced in (0, 1):
        st = {"forced": torch.randint(3, 50, (B, tmax), dtype=i64, device=dev), "use_forced": torch.full((1,), use_forced, dtype=i32, device=dev),
              "steps": torch.zeros((B,), dtype=i32, device=dev), "finished": torch.zeros((B,), dtype=torch.uint8, device=dev),
              "out_ids": torch.full((B, tmax), pad, dtype=i64, device=dev), "lens": lens, "t_rows": t_rows,
              "tok": torch.zeros((B,), dtype=i64, device=dev), "pos": torch.zeros((B,), dtype=i32, device=dev),
              "nk": torch.zeros((B,), dtype=i32, device=dev), "hidden": torch.zeros((B, tmax, H), dtype=torch.bfloat16, device=dev)}
        st["forced"][1, 2] = eos                      # row 1 ends at its third token when forced
        ref_out = st["out_ids"].clone().cpu()
        ref_hid = torch.zeros((B, tmax, H))
        fin = [False] * B
        g = torch.Generator().manual_seed(7 + use_forced)
        for s_ in range(6):
            nxt = torch.randint(3, 50, (B,), generator=g)
            if s_ == 3:
                nxt[3] = eos                          # row 3 ends at its fourth token when greedy
            h1 = torch.randn((B, H), generator=g).to(torch.bfloat16)
            ops.decode_book(nxt.to(dev), st, h1.to(dev) if s_ else None, pad, eos)
            for b in range(B):
                t = int(st["forced"][b, s_]) if use_forced else int(nxt[b])
                if fin[b]:
                    t = pad
                ref_out[b, int(lens[b]) + s_] = t
                fin[b] = fin[b] or t == eos
                if s_ >= 1:
                    ref_hid[b, int(t_rows[b]) + s_ - 1] = h1[b].float()
                assert int(st["tok"][b]) == t and int(st["pos"][b]) == int(t_rows[b]) + s_ and int(st["nk"][b]) == int(t_rows[b]) + s_ + 1
            assert st["finished"].cpu().bool().tolist() == fin and st["steps"].cpu().tolist() == [s_ + 1] * B
        assert torch.equal(st["out_ids"].cpu(), ref_out) and torch.equal(st["hidden"].float().cpu(), ref_hid)
        assert any(fin)

