"""Tensor-level wrappers over the C-ABI of libhaff_hip.so.

torch is used here only for device memory and the current HIP stream; every arithmetic op below is a
hand-written HIP kernel. All functions require CUDA(HIP) tensors and raise if the library is missing.
dtype policy: torch.bfloat16 = throughput mode (bf16 MFMA, fp32 accumulate), torch.float32 = parity mode.
"""
import ctypes

import torch

from .lib import check, load_library

ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_RELU, ACT_SILU = 0, 1, 2, 3, 4


def _dt(t):
    if t.dtype == torch.bfloat16:
        return 0
    if t.dtype == torch.float32:
        return 1
    raise TypeError(f"unsupported dtype {t.dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _req(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live in HBM (cuda tensor); the hot path has no CPU fallback")


_WORKSPACES = {}


def _workspace(device, nbytes):
    """Scratch HBM for kernels that take a caller-provided workspace: one buffer per (device, stream), so launches on
    different HIP streams never share one.
    Under hipGraph capture the key is the CAPTURE stream: the buffer is then allocated from the graph's pool and its address is
    baked into the graph. Branches of one capture that fork onto other streams get their own buffers (their own keys), but two
    different graphs may end up holding the same buffer — which is why LisaMI355 replays its graphs one after another on one
    stream (lisa.py: `_graph_pool`, "replays never overlap") and never replays one beside an eager launch that uses the
    workspace of the same stream key. A caller that wants concurrent replays must capture them with distinct workspaces."""
    key = (device.index, _stream())
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WORKSPACES[key] = torch.empty((nbytes,), dtype=torch.uint8, device=device)
    return buf


def row_stats(x, eps, rms=False):
    """Per-row {mean, rstd} (rms: {0, rsqrt(mean(x^2)+eps)}) as fp32 [rows, 2] — the normalisation itself is applied
    by linear(..., ln_stats=, ln_colsum=)."""
    lib = load_library()
    _req(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    stats = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    rc = lib.haff_row_stats(x.data_ptr(), x.stride(0), stats.data_ptr(), x.shape[0], x.shape[1], float(eps),
                            1 if rms else 0, _dt(x), _stream())
    check(rc, "haff_row_stats")
    return stats


def linear_rowstats_supported(M, N, K, dtype, min_tiles=160):
    """Shapes whose residual product can emit the LayerNorm statistics of its output rows (haff_gemm_bf16_rowstats): whole
    256 x 256 tiles, and (min_tiles) enough of them that the 8-wave tile is what linear() would launch anyway."""
    return dtype == torch.bfloat16 and M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and (M // 256) * (N // 256) >= min_tiles


def rowstats_gemm(x, w, bias, resid, out, a_map=None):
    """The product of linear_rowstats alone (one haff_gemm_bf16_rowstats launch): returns the partial sums fp32 [M, N/64, 2]."""
    lib = load_library()
    M = a_map.numel() if a_map is not None else x.shape[0]
    N, K = w.shape
    part = torch.empty((M, N // 64, 2), dtype=torch.float32, device=x.device)
    rc = lib.haff_gemm_bf16_rowstats(x.data_ptr(), x.stride(0), _p(a_map), x.shape[0], w.data_ptr(), w.stride(0), out.data_ptr(),
                                     out.stride(0), _p(bias), resid.data_ptr(), resid.stride(0), M, N, K, part.data_ptr(), _stream())
    check(rc, "haff_gemm_bf16_rowstats")
    return part


def linear_rowstats(x, w, bias, resid, eps, out=None, a_map=None):
    """out = x @ w.T + bias + resid (bf16; out may be resid) AND the {mean, rstd} of every OUTPUT row as fp32 [M, 2] — the
    ln_stats of the next linear(..., ln_stats=): the producer's epilogue sums its own results, nobody reads the rows again."""
    lib = load_library()
    _req(x, "x")
    M, K = x.shape
    if a_map is not None:
        assert a_map.dtype == torch.int32
        M = a_map.numel()
    N = w.shape[0]
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K
    assert linear_rowstats_supported(M, N, K, x.dtype, 0) and resid is not None and resid.dtype == torch.bfloat16
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    part = rowstats_gemm(x, w, bias, resid, out, a_map)
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    check(lib.haff_row_stats_finalize(part.data_ptr(), stats.data_ptr(), M, N // 64, N, float(eps), _stream()),
          "haff_row_stats_finalize")
    return out, stats


def rowstats32_gemm(x, w, bias, x32, out16, a_map=None):
    """The product of linear_rowstats32 alone (one haff_gemm_bf16_rowstats32 launch): returns the partial sums fp32 [M, N/64, 2]."""
    lib = load_library()
    M = a_map.numel() if a_map is not None else x.shape[0]
    N, K = w.shape
    part = torch.empty((M, N // 64, 2), dtype=torch.float32, device=x.device)
    rc = lib.haff_gemm_bf16_rowstats32(x.data_ptr(), x.stride(0), _p(a_map), x.shape[0], w.data_ptr(), w.stride(0), x32.data_ptr(),
                                       x32.stride(0), out16.data_ptr(), out16.stride(0), _p(bias), M, N, K, part.data_ptr(), _stream())
    check(rc, "haff_gemm_bf16_rowstats32")
    return part


def linear_rowstats32(x, w, bias, x32, out16, eps, a_map=None):
    """x32 += x @ w.T + bias IN PLACE on the fp32 residual stream, out16 = bf16(x32) (the next product's operand), and the
    {mean, rstd} of every row of the new x32 as fp32 [M, 2] (haff_gemm_bf16_rowstats32 + haff_row_stats_finalize)."""
    lib = load_library()
    _req(x, "x")
    M = a_map.numel() if a_map is not None else x.shape[0]
    N, K = w.shape
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.stride(1) == 1 and w.stride(1) == 1 and x.shape[1] == K
    assert linear_rowstats_supported(M, N, K, x.dtype, 0) and x32.dtype == torch.float32 and out16.dtype == torch.bfloat16
    assert x32.shape == (M, N) and out16.shape == (M, N) and x32.stride(1) == 1 and out16.stride(1) == 1 and bias is not None
    part = rowstats32_gemm(x, w, bias, x32, out16, a_map)
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    check(lib.haff_row_stats_finalize(part.data_ptr(), stats.data_ptr(), M, N // 64, N, float(eps), _stream()),
          "haff_row_stats_finalize")
    return stats


def rope_permute_rows(w):
    """The row order haff_gemm_bf16_qkv_rope wants for the fused q|k|v weights [3*H*128, K]: inside every 256-row tile, natural
    row wn*64 + t*16 + i takes logical row (wn>>1)*128 + (t>>1)*64 + (wn&1)*32 + (t&1)*16 + i."""
    N = w.shape[0]
    assert N % 256 == 0
    j = torch.arange(256)
    wn, t, i = j // 64, (j // 16) % 4, j % 16
    logical = (wn // 2) * 128 + (t // 2) * 64 + (wn % 2) * 32 + (t % 2) * 16 + i
    idx = (torch.arange(0, N, 256)[:, None] + logical[None, :]).reshape(-1).to(w.device)
    return w.index_select(0, idx).contiguous()


def qkv_rope_supported(M, H, d, K, dtype, min_rows=1024):
    """Prefill-sized batches whose q|k|v projection can carry RoPE and the cache append (haff_gemm_bf16_qkv_rope): (min_rows)
    enough rows that the 8-wave tile is what linear() would launch anyway."""
    return dtype == torch.bfloat16 and d == 128 and (H * d) % 256 == 0 and K % 64 == 0 and min_rows <= M < (1 << 22)


def qkv_rope(x, w_perm, kcache, vcache, cos_sin, B, T, H, d, pos0):
    """Rotated q [B*T, H*d] of x @ w.T; the rotated k and v rows land in kcache / vcache [B, Tmax, H*d] at pos0 .. pos0+T-1."""
    lib = load_library()
    _req(x, "x")
    M, K = x.shape
    assert M == B * T and w_perm.shape == (3 * H * d, K) and x.stride(1) == 1 and w_perm.stride(1) == 1
    assert x.dtype == torch.bfloat16 and w_perm.dtype == torch.bfloat16 and kcache.dtype == torch.bfloat16
    assert kcache.is_contiguous() and vcache.is_contiguous() and kcache.shape == vcache.shape and kcache.shape[2] == H * d
    assert cos_sin.dtype == torch.float32 and cos_sin.is_contiguous() and cos_sin.shape[0] >= kcache.shape[1] and cos_sin.shape[1] == d
    q = torch.empty((M, H * d), dtype=x.dtype, device=x.device)
    rc = lib.haff_gemm_bf16_qkv_rope(x.data_ptr(), x.stride(0), w_perm.data_ptr(), w_perm.stride(0), q.data_ptr(), q.stride(0),
                                     kcache.data_ptr(), vcache.data_ptr(), cos_sin.data_ptr(), B, T, kcache.shape[1], int(pos0),
                                     H, d, K, _stream())
    check(rc, "haff_gemm_bf16_qkv_rope")
    return q


def fold_norm(w, gamma, beta=None, bias=None):
    """Fold y = norm(x) * gamma + beta followed by y @ w.T + bias into the weights: returns (w_bf16 = bf16(w * gamma),
    colsum fp32 [N] of the ROUNDED folded weights, bias' = bias + w @ beta)."""
    wf = (w.float() * gamma.float()[None, :]).to(torch.bfloat16).contiguous()
    colsum = wf.float().sum(1).contiguous()
    b = None
    if beta is not None or bias is not None:
        b = torch.zeros((w.shape[0],), dtype=torch.float32, device=w.device)
        if bias is not None:
            b += bias.float()
        if beta is not None:
            b += w.float() @ beta.float()
        b = b.contiguous()
    return wf, colsum, b


SPLIT_ROW_TAIL = True   # linear(): a <= 64-row tail that would cost the 8-wave tile an extra round goes out as its own launch


def linear(x, w, bias=None, act=ACT_NONE, resid=None, row_map=None, out=None, out_rows=None, out_dtype=None,
           swiglu=False, tile_cfg=0, a_map=None, ln_stats=None, ln_colsum=None):
    """y = epi(x @ w.T): x [M,K] (row stride free, unit inner stride), w [N,K], bias fp32 [N] or None.

    epilogue order: +bias -> act -> +resid (resid indexed like out). row_map (int32 [M]) redirects output
    (and residual) rows, negative entries are dropped. swiglu: w rows interleaved [gate x16 | up x16], N -> N/2.
    """
    lib = load_library()
    _req(x, "x")
    assert x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    if a_map is not None:  # gather: logical row m reads x[a_map[m]] (bf16 only)
        assert a_map.dtype == torch.int32 and x.dtype == torch.bfloat16
        M = a_map.numel()
    N = w.shape[0]
    assert w.shape[1] == K, (x.shape, w.shape)
    n_out = N // 2 if swiglu else N
    if out_dtype is None:
        out_dtype = x.dtype
    if out is None:
        out = torch.empty((out_rows if out_rows is not None else M, n_out), dtype=out_dtype, device=x.device)
    assert out.stride(1) == 1 and out.shape[1] == n_out
    if resid is not None:
        assert resid.dtype == out.dtype and resid.stride(1) == 1
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    if row_map is not None:
        assert row_map.dtype == torch.int32 and row_map.numel() == M
    if (SPLIT_ROW_TAIL and x.dtype == torch.bfloat16 and M > 4096 and 0 < (M & 255) <= 64 and row_map is None and a_map is None
            and ln_stats is None and not swiglu and not tile_cfg and out_rows is None and N >= 256):
        # A short row tail that costs the persistent 256 x 256 tile a whole extra round (CLIP at 64 frames: 16448 rows = 64.25
        # row tiles; N = 1024 gives 260 tiles on 256 CUs, the last 4 alone in a second round: 758 TFLOP/s): the whole row
        # tiles and the <= 64 tail rows go out as two launches (the tail is a weight-streaming product)
        cols = (N + 255) // 256
        if -(-((M + 255) // 256 * cols) // 256) > -(-((M // 256) * cols) // 256):
            m0 = M & ~255
            _LINEAR(x[:m0], w, bias, act, None if resid is None else resid[:m0], out=out[:m0])
            _LINEAR(x[m0:], w, bias, act, None if resid is None else resid[m0:], out=out[m0:])
            return out
    if ln_stats is not None:
        assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a_map is None
        assert ln_stats.dtype == torch.float32 and ln_stats.shape == (M, 2) and ln_stats.is_contiguous()
        if ln_colsum is not None:
            assert ln_colsum.dtype == torch.float32 and ln_colsum.numel() == N
        rc = lib.haff_gemm_bf16_ln(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(),
                                   out.stride(0), _p(bias), _p(resid), 0 if resid is None else resid.stride(0),
                                   _p(row_map), ln_stats.data_ptr(), _p(ln_colsum), M, N, K, act,
                                   1 if out.dtype == torch.float32 else 0, 1 if swiglu else 0, _stream())
    elif a_map is not None:
        assert w.dtype == torch.bfloat16
        rc = lib.haff_gemm_bf16_gather(x.data_ptr(), x.stride(0), a_map.data_ptr(), x.shape[0], w.data_ptr(),
                                       w.stride(0), out.data_ptr(), out.stride(0), _p(bias), _p(resid),
                                       0 if resid is None else resid.stride(0), _p(row_map), M, N, K, act,
                                       1 if out.dtype == torch.float32 else 0, 1 if swiglu else 0, _stream())
    elif x.dtype == torch.bfloat16 and 32 < M <= 1024 and not tile_cfg:
        # few output tiles (decode at batch 33..64 on narrow weights, prefill / CLIP at one frame): the library may split K
        # over workgroups through a per-stream fp32 workspace (partials summed in a fixed order)
        assert w.dtype == torch.bfloat16
        ws = _workspace(x.device, 64 << 20)
        rc = lib.haff_gemm_bf16_ws(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0),
                                   _p(bias), _p(resid), 0 if resid is None else resid.stride(0), _p(row_map), M, N, K, act,
                                   1 if out.dtype == torch.float32 else 0, 1 if swiglu else 0, ws.data_ptr(), ws.numel(), _stream())
    elif x.dtype == torch.bfloat16:
        assert w.dtype == torch.bfloat16
        rc = lib.haff_gemm_bf16_cfg(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(),
                                    out.stride(0), _p(bias), _p(resid), 0 if resid is None else resid.stride(0),
                                    _p(row_map), M, N, K, act, 1 if out.dtype == torch.float32 else 0,
                                    1 if swiglu else 0, tile_cfg, _stream())
    else:
        assert w.dtype == torch.float32 and out.dtype == torch.float32
        rc = lib.haff_gemm_f32(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0),
                               _p(bias), _p(resid), 0 if resid is None else resid.stride(0), _p(row_map),
                               M, N, K, act, 1 if swiglu else 0, _stream())
    check(rc, "haff_gemm")
    return out


def linear_heads_supported(M, N, K, d, heads, dtype):
    """Shapes haff_gemm_bf16_heads serves: whole 256 x 256 tiles of the 8-wave kernel, N = parts * heads * d."""
    return (dtype == torch.bfloat16 and M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and d % 8 == 0 and N % (heads * d) == 0
            and N // (heads * d) <= 3 and heads * d < (1 << 16) and M * K * 2 < (1 << 32) and N * K * 2 < (1 << 32))


def gemm_stream_cap(cap, stream=None):
    """Workgroups per launch of the persistent GEMM tile for launches enqueued on `stream` (a torch stream; default: the current
    one) from now on (haff_gemm_stream_cap; 256 = every CU). Returns the stream's previous setting. Scheduling only: outputs do
    not depend on it, launches on other streams are unaffected."""
    s = _stream() if stream is None else stream.cuda_stream
    rc = int(load_library().haff_gemm_stream_cap(s, int(cap)))
    if rc < 0:
        check(rc, "haff_gemm_stream_cap")
    return rc


def linear_heads(x, w, bias, row_map, out, d, heads, part_stride, head_stride, ln_stats=None, ln_colsum=None):
    """x @ w.T (+ folded norm) scattered HEAD-MAJOR into `out` (haff_gemm_bf16_heads): product column part * heads * d + h * d + c of
    row m goes to out.flat[part * part_stride + h * head_stride + row_map[m] * d + c]."""
    lib = load_library()
    _req(x, "x")
    M, K = x.shape
    N = w.shape[0]
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and out.is_contiguous()
    assert x.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K and row_map.dtype == torch.int32 and row_map.numel() == M
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == N)
    if ln_stats is not None:
        assert ln_stats.dtype == torch.float32 and ln_stats.shape == (M, 2) and ln_stats.is_contiguous()
    parts = N // (heads * d)
    assert out.numel() >= (parts - 1) * part_stride + heads * head_stride   # (the row map's range is the caller's contract)
    rc = lib.haff_gemm_bf16_heads(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), _p(bias), row_map.data_ptr(),
                                  _p(ln_stats), _p(ln_colsum), M, N, K, d, heads, part_stride, head_stride, _stream())
    check(rc, "haff_gemm_bf16_heads")
    return out


_LINEAR = linear   # the function itself: linear()'s own two-launch form must not go through a wrapper installed on ops.linear (bench.py's meter)


def linear_rms(x, w, resid=None, out=None, swiglu=False, ssq_in=None, ssq_out=None, eps=0.0):
    """Decode-sized bf16 product (M <= 16) carrying RMSNorm statistics between products (haff_gemm_bf16_rms):
    ssq_in fp32 [parts,16]: rows are scaled by rsqrt(sum(ssq_in[:, m]) / K + eps) (w = gamma-folded weights, x = raw
    residual stream); ssq_out fp32 [>= N/16, 16]: receives this product's per-workgroup sums of squares of the bf16 output."""
    lib = load_library()
    _req(x, "x")
    M, K = x.shape
    N = w.shape[0]
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    assert out.dtype == torch.bfloat16 and out.stride(1) == 1 and out.shape == (M, n_out)
    if resid is not None:
        assert resid.dtype == out.dtype and resid.stride(1) == 1
    if ssq_in is not None:
        assert ssq_in.dtype == torch.float32 and ssq_in.is_contiguous() and ssq_in.shape[1] == 16
    if ssq_out is not None:
        assert ssq_out.dtype == torch.float32 and ssq_out.is_contiguous() and ssq_out.shape[1] == 16 and ssq_out.shape[0] * 16 >= N
    rc = lib.haff_gemm_bf16_rms(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), None,
                                _p(resid), 0 if resid is None else resid.stride(0), M, N, K, ACT_NONE, 0, 1 if swiglu else 0,
                                _p(ssq_in), 0 if ssq_in is None else ssq_in.shape[0], float(eps), _p(ssq_out), None, _stream())
    check(rc, "haff_gemm_bf16_rms")
    return out


class ChainLayer(ctypes.Structure):
    """haff_chain_layer of include/haff_hip.h: the six device pointers of one Llama layer on the chained decode step."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("wqkv", "wo", "wgu", "wd", "kcache", "vcache")]


def decode_chain_supported(M, hidden, ffn, heads, n_layers):
    return int(load_library().haff_decode_chain_supported(int(M), int(hidden), int(ffn), int(heads), int(n_layers))) > 0


def decode_chain_sync_words(n_layers, hidden):
    return int(load_library().haff_decode_chain_sync_words(int(n_layers), int(hidden)))


def decode_chain_table(layers):
    """layers: one (wqkv, wo, wgu, wd, kcache, vcache) tuple of bf16 tensors per layer -> the host table haff_decode_chain_bf16 takes."""
    for row in layers:
        for t in row:
            _req(t, "chain operand")
            assert t.dtype == torch.bfloat16 and t.is_contiguous()
    return (ChainLayer * len(layers))(*[ChainLayer(*[t.data_ptr() for t in row]) for row in layers])


def decode_chain(table, n_layers, x, qkv, att, g, ssq_a, ssq_b, ws, stats0, eps, cos_sin, nk_rows, heads, tmax, scale, sync,
                 per_stage_launches=False):
    """One KV-cached decode step of the whole Llama stack at <= 8 rows as ONE launch (haff_decode_chain_bf16): x [M,H] bf16 is
    the residual stream (in place); qkv [M,3H], att [M,H], g [M,F], ssq_a / ssq_b f32 [H/16,16]: scratch; stats0 f32 [M,2];
    ws f32 [H/16, 2, 16, 16]: scratch; sync uint32 (int32 tensor) [decode_chain_sync_words(n_layers, H)], zeroed once at allocation. per_stage_launches: the same kernel, one launch per
    (layer, stage) — identical arithmetic without the chaining (tests, A/B)."""
    lib = load_library()
    _req(x, "x")
    M, H = x.shape
    F = g.shape[1]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and qkv.is_contiguous() and att.is_contiguous() and g.is_contiguous()
    assert qkv.shape == (M, 3 * H) and att.shape == (M, H) and g.shape[0] == M
    assert ssq_a.dtype == torch.float32 and ssq_a.shape == (H // 16, 16) and ssq_b.shape == (H // 16, 16)
    assert stats0.dtype == torch.float32 and stats0.shape == (M, 2) and stats0.is_contiguous()
    assert nk_rows.dtype == torch.int32 and nk_rows.numel() == M and cos_sin.dtype == torch.float32 and cos_sin.shape[1] == 128
    assert sync.dtype == torch.int32 and sync.numel() >= decode_chain_sync_words(n_layers, H)
    assert ws.dtype == torch.float32 and ws.is_contiguous() and ws.numel() >= (H // 16) * 2 * 256
    rc = lib.haff_decode_chain_bf16(table, int(n_layers), M, H, F, int(heads), x.data_ptr(), qkv.data_ptr(), att.data_ptr(),
                                    g.data_ptr(), ssq_a.data_ptr(), ssq_b.data_ptr(), ws.data_ptr(), stats0.data_ptr(), float(eps),
                                    cos_sin.data_ptr(), nk_rows.data_ptr(), int(tmax), float(scale), sync.data_ptr(),
                                    1 if per_stage_launches else 0, _stream())
    check(rc, "haff_decode_chain_bf16")
    return x


def decode_chain_status(sync, n_layers):
    """True when every bounded wait of the chained launches so far was satisfied (synchronises the current stream)."""
    rc = int(load_library().haff_decode_chain_status(sync.data_ptr(), int(n_layers), _stream()))
    if rc < 0:
        check(rc, "haff_decode_chain_status")
    return rc == 0


def attention(q, k, v, scale, causal=False, q_pos0=0, relh=None, relw=None, S=0, out=None):
    """q [B,H,Nq,d], k/v [B,H,Nk,d] strided views (unit stride on d). Returns out [B,Nq,H*d] (token-major)."""
    lib = load_library()
    _req(q, "q")
    B, H, Nq, d = q.shape
    Nk = k.shape[2]
    assert q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    if out is None:
        out = torch.empty((B, Nq, H * d), dtype=q.dtype, device=q.device)
    o4 = out.view(B, Nq, H, d).permute(0, 2, 1, 3)
    args = [q.data_ptr(), q.stride(0), q.stride(1), q.stride(2),
            k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
            v.data_ptr(), v.stride(0), v.stride(1), v.stride(2),
            out.data_ptr(), o4.stride(0), o4.stride(1), o4.stride(2),
            B, H, Nq, Nk, d, float(scale), 1 if causal else 0, int(q_pos0), _p(relh), _p(relw), int(S), _stream()]
    if relh is not None:
        assert relh.dtype == torch.float32 and relh.is_contiguous() and relw.is_contiguous()
    if q.dtype == torch.bfloat16:
        rc = lib.haff_attention_bf16(*args)
    else:
        rc = lib.haff_attention_f32(*args)
    check(rc, "haff_attention")
    return out


_BF16_TABLES = {}


def _bf16_table(t):
    key = (t.data_ptr(), tuple(t.shape))
    hit = _BF16_TABLES.get(key)
    if hit is None or hit[0] is not t:
        _BF16_TABLES[key] = (t, t.to(torch.bfloat16).contiguous())
    return _BF16_TABLES[key][1]


def relpos_tables(q, tab_h, tab_w, S):
    """q [B,H,N,d] view with N == S*S; tab_* fp32 [2S-1,d]. Returns relh, relw fp32 [B*H,N,S]."""
    lib = load_library()
    B, H, N, d = q.shape
    assert N == S * S and tab_h.shape == (2 * S - 1, d) and tab_h.dtype == torch.float32
    relh = torch.empty((B * H, N, S), dtype=torch.float32, device=q.device)
    relw = torch.empty_like(relh)
    if q.dtype == torch.bfloat16:
        # throughput mode: MFMA kernel on bf16 tables (the reference's bf16 checkpoint stores them in bf16 too)
        th, tw = _bf16_table(tab_h), _bf16_table(tab_w)
        rc = lib.haff_relpos_tables_bf16(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), th.data_ptr(),
                                         tw.data_ptr(), relh.data_ptr(), relw.data_ptr(), B, H, S, d, _stream())
        check(rc, "haff_relpos_tables_bf16")
        return relh, relw
    rc = lib.haff_relpos_tables(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), tab_h.data_ptr(),
                                tab_w.data_ptr(), relh.data_ptr(), relw.data_ptr(), B, H, S, d, _dt(q), _stream())
    check(rc, "haff_relpos_tables")
    return relh, relw


def window_attention_supported(q, S):
    """Geometry served by the fused window kernel (haff_window_attention_bf16)."""
    return q.dtype == torch.bfloat16 and S == 14 and q.shape[3] == 80 and q.shape[2] == S * S


def window_attention(q, k, v, scale, tab_h, tab_w, S, out=None, grid=0, pad_token=0):
    """Fused SAM window attention + decomposed rel-pos. q/k/v [n_windows,H,S*S,d] strided views, tab_* fp32 [2S-1,d]
    (rounded to bf16 once and cached, as the reference's bf16 checkpoint stores them). Returns [n_windows,S*S,H*d].
    grid > 0: windows tile grid x grid token images and the padded window tokens were never written; their q/k/v
    are taken from token row `pad_token` of the views (the caller stores the qkv bias there)."""
    lib = load_library()
    _req(q, "q")
    B, H, N, d = q.shape
    assert window_attention_supported(q, S) and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    th, tw = _bf16_table(tab_h), _bf16_table(tab_w)
    if out is None:
        out = torch.empty((B, N, H * d), dtype=q.dtype, device=q.device)
    o4 = out.view(B, N, H, d).permute(0, 2, 1, 3)
    rc = lib.haff_window_attention_bf16(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2),
                                        k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
                                        v.data_ptr(), v.stride(0), v.stride(1), v.stride(2),
                                        out.data_ptr(), o4.stride(0), o4.stride(1), o4.stride(2),
                                        B, H, S, d, float(scale), th.data_ptr(), tw.data_ptr(), int(grid), int(grid),
                                        int(pad_token), _stream())
    check(rc, "haff_window_attention_bf16")
    return out


def global_attention_supported(q, k, v, S):
    """Geometry served by the fused global kernel (haff_global_attention_bf16): ViT-H global blocks, k|v in one row layout."""
    return (q.dtype == torch.bfloat16 and S == 64 and q.shape[3] == 80 and q.shape[2] == S * S and k.shape[2] == S * S
            and k.stride() == v.stride() and v.data_ptr() >= k.data_ptr()
            and (v.data_ptr() - k.data_ptr()) + k.shape[2] * k.stride(2) * 2 < (1 << 31))


def global_attention(q, k, v, scale, tab_h, tab_w, S, out=None):
    """Fused SAM global attention + decomposed rel-pos (no rel-pos tables in HBM). q/k/v [B,H,S*S,d] strided views, tab_* fp32
    [2S-1,d] (rounded to bf16 once and cached, as the reference's bf16 checkpoint stores them). Returns [B,S*S,H*d]."""
    lib = load_library()
    _req(q, "q")
    B, H, N, d = q.shape
    assert global_attention_supported(q, k, v, S) and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    th, tw = _bf16_table(tab_h), _bf16_table(tab_w)
    if out is None:
        out = torch.empty((B, N, H * d), dtype=q.dtype, device=q.device)
    o4 = out.view(B, N, H, d).permute(0, 2, 1, 3)
    rc = lib.haff_global_attention_bf16(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2),
                                        k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
                                        v.data_ptr(), v.stride(0), v.stride(1), v.stride(2),
                                        out.data_ptr(), o4.stride(0), o4.stride(1), o4.stride(2),
                                        B, H, S, d, float(scale), th.data_ptr(), tw.data_ptr(), _stream())
    check(rc, "haff_global_attention_bf16")
    return out


def _norm_dt(x, out):
    """dtype code of the norm kernels: 0 bf16, 1 f32, 2 = f32 rows in, bf16 rows out (fp32 residual stream -> bf16 product)."""
    if x.dtype == torch.float32 and out.dtype == torch.bfloat16:
        return 2
    assert out.dtype == x.dtype, (x.dtype, out.dtype)
    return _dt(x)


def layernorm(x, w, b, eps, in_map=None, out=None, out_dtype=None):
    """x [R,C]; optional gather map (int32 [R_out], <0 -> zero row). out_dtype=torch.bfloat16 on fp32 rows: one rounding, of the
    normalised row."""
    lib = load_library()
    _req(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    rows = x.shape[0] if in_map is None else in_map.numel()
    C = x.shape[1]
    if out is None:
        out = torch.empty((rows, C), dtype=out_dtype or x.dtype, device=x.device)
    rc = lib.haff_layernorm(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), w.data_ptr(), b.data_ptr(),
                            _p(in_map), rows, C, float(eps), _norm_dt(x, out), _stream())
    check(rc, "haff_layernorm")
    return out


def rmsnorm(x, w, eps, out=None, out_dtype=None):
    lib = load_library()
    _req(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    rc = lib.haff_rmsnorm(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), w.data_ptr(), x.shape[0],
                          x.shape[1], float(eps), _norm_dt(x, out), _stream())
    check(rc, "haff_rmsnorm")
    return out


def patchify_nchw(x, P, gh, gw, Kp, out_dtype):
    lib = load_library()
    _req(x, "x")
    x = x.contiguous()
    B, Cin, Hin, Win = x.shape
    out = torch.empty((B * gh * gw, Kp), dtype=out_dtype, device=x.device)
    rc = lib.haff_patchify_nchw(x.data_ptr(), out.data_ptr(), B, Cin, Hin, Win, P, gh, gw, Kp, _dt(x), _dt(out),
                                _stream())
    check(rc, "haff_patchify_nchw")
    return out


def patchify_u8(frames, P, gh, gw, Kp, mean3, std3, out_dtype):
    """frames uint8 NHWC [B,Hf,Wf,3]; mean3/std3 host sequences of 3 floats (0..255 scale)."""
    import ctypes
    lib = load_library()
    _req(frames, "frames")
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.shape[3] == 3
    B, Hf, Wf, _ = frames.shape
    out = torch.empty((B * gh * gw, Kp), dtype=out_dtype, device=frames.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean3])
    s = (ctypes.c_float * 3)(*[float(v) for v in std3])
    rc = lib.haff_patchify_u8(frames.data_ptr(), out.data_ptr(), B, Hf, Wf, P, gh, gw, Kp,
                              ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), _dt(out), _stream())
    check(rc, "haff_patchify_u8")
    return out


def im2col3x3(x):
    """x [B,H,W,C] channels-last contiguous -> [B*H*W, 9*C]."""
    lib = load_library()
    _req(x, "x")
    assert x.is_contiguous()
    B, H, W, C = x.shape
    out = torch.empty((B * H * W, 9 * C), dtype=x.dtype, device=x.device)
    rc = lib.haff_im2col3x3(x.data_ptr(), out.data_ptr(), B, H, W, C, _dt(x), _stream())
    check(rc, "haff_im2col3x3")
    return out


def embed_splice(ids, img_pos, embed, img):
    """ids int64 [B,L] (sentinel < 0 at img_pos[b]); embed [V,Hd]; img [B,n_img,Hd] -> [B, L+n_img-1, Hd]."""
    lib = load_library()
    _req(ids, "ids")
    B, L = ids.shape
    n_img, Hd = img.shape[1], img.shape[2]
    assert ids.dtype == torch.int64 and ids.is_contiguous() and img.is_contiguous() and embed.is_contiguous()
    assert img_pos.dtype == torch.int32
    out = torch.empty((B, L + n_img - 1, Hd), dtype=embed.dtype, device=embed.device)
    rc = lib.haff_embed_splice(ids.data_ptr(), img_pos.data_ptr(), embed.data_ptr(), img.data_ptr(), out.data_ptr(),
                               B, L, n_img, Hd, _dt(embed), _stream())
    check(rc, "haff_embed_splice")
    return out


def rope_cache(qkv, kcache, vcache, cos_sin, B, Tq, Hq, Hkv, d, pos0):
    """qkv [B*Tq, (Hq+2Hkv)*d] rotated in place; k,v appended to caches [B,Tmax,Hkv*d]."""
    lib = load_library()
    _req(qkv, "qkv")
    Tmax = kcache.shape[1]
    assert kcache.is_contiguous() and vcache.is_contiguous() and cos_sin.dtype == torch.float32
    rc = lib.haff_rope_cache(qkv.data_ptr(), qkv.stride(0), kcache.data_ptr(), vcache.data_ptr(), cos_sin.data_ptr(),
                             B, Tq, Hq, Hkv, d, pos0, Tmax, _dt(qkv), _stream())
    check(rc, "haff_rope_cache")


def rope_cache_rows(qkv, kcache, vcache, cos_sin, B, Tq, Hq, Hkv, d, pos0_rows):
    """rope_cache with a per-row start position (int32 [B] on the device): ragged right-padded batches."""
    lib = load_library()
    _req(qkv, "qkv")
    Tmax = kcache.shape[1]
    assert kcache.is_contiguous() and vcache.is_contiguous() and cos_sin.dtype == torch.float32
    assert pos0_rows.dtype == torch.int32 and pos0_rows.is_cuda and pos0_rows.numel() == B
    rc = lib.haff_rope_cache_rows(qkv.data_ptr(), qkv.stride(0), kcache.data_ptr(), vcache.data_ptr(), cos_sin.data_ptr(),
                                  B, Tq, Hq, Hkv, d, pos0_rows.data_ptr(), Tmax, _dt(qkv), _stream())
    check(rc, "haff_rope_cache_rows")


def attention_decode_rows(q, k, v, scale, nk_rows, out=None):
    """One query per (batch, head) against ragged KV caches: q [B,H,1,d] view, k/v [B,H,Nk,d] views of the whole cache,
    nk_rows int32 [B] on the device (keys visible per batch entry). Returns [B,1,H*d]."""
    lib = load_library()
    _req(q, "q")
    B, H, Nq, d = q.shape
    assert Nq == 1 and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    assert nk_rows.dtype == torch.int32 and nk_rows.is_cuda and nk_rows.numel() == B
    if out is None:
        out = torch.empty((B, 1, H * d), dtype=q.dtype, device=q.device)
    fn = lib.haff_attention_decode_rows_bf16 if q.dtype == torch.bfloat16 else lib.haff_attention_decode_rows_f32
    rc = fn(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
            v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), out.data_ptr(), H * d, d, B, H, k.shape[2], d, float(scale),
            nk_rows.data_ptr(), _stream())
    check(rc, "haff_attention_decode_rows")
    return out


def decode_attention_rope(qkv, kcache, vcache, cos_sin, H, d, scale, nk_rows):
    """bf16 decode position with RoPE + cache append fused: qkv [B, 3*H*d] raw -> [B, 1, H*d]; nk_rows int32 [B] = new pos + 1."""
    lib = load_library()
    _req(qkv, "qkv")
    B = qkv.shape[0]
    assert qkv.dtype == torch.bfloat16 and qkv.stride(1) == 1 and kcache.is_contiguous() and vcache.is_contiguous()
    assert nk_rows.dtype == torch.int32 and nk_rows.is_cuda and nk_rows.numel() == B and cos_sin.dtype == torch.float32
    out = torch.empty((B, 1, H * d), dtype=qkv.dtype, device=qkv.device)
    rc = lib.haff_decode_attention_rope_rows_bf16(qkv.data_ptr(), qkv.stride(0), kcache.data_ptr(), vcache.data_ptr(),
                                                  cos_sin.data_ptr(), out.data_ptr(), B, H, d, kcache.shape[1], float(scale),
                                                  nk_rows.data_ptr(), _stream())
    check(rc, "haff_decode_attention_rope_rows_bf16")
    return out


def argmax_rows(logits):
    lib = load_library()
    _req(logits, "logits")
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    out = torch.empty((logits.shape[0],), dtype=torch.int64, device=logits.device)
    rc = lib.haff_argmax_rows(logits.data_ptr(), logits.stride(0), out.data_ptr(), logits.shape[0], logits.shape[1],
                              _stream())
    check(rc, "haff_argmax_rows")
    return out


def decode_book(nxt_raw, st, h1, pad, eos):
    """Device-side bookkeeping of one greedy-decode token (haff_decode_book). st: the persistent per-(batch, capacity) state
    dict of LisaMI355._persistent_cache (forced, use_forced, steps, finished, out_ids, lens, t_rows, tok, pos, nk, hidden)."""
    lib = load_library()
    _req(nxt_raw, "nxt_raw")
    B = nxt_raw.shape[0]
    hid = st["hidden"]
    assert nxt_raw.dtype == torch.int64 and st["out_ids"].dtype == torch.int64 and st["forced"].dtype == torch.int64
    assert hid.is_contiguous() and (h1 is None or (h1.is_contiguous() and h1.numel() == B * hid.shape[2] and h1.dtype == hid.dtype))
    rc = lib.haff_decode_book(nxt_raw.data_ptr(), st["forced"].data_ptr(), st["forced"].stride(0), st["use_forced"].data_ptr(),
                              st["steps"].data_ptr(), st["finished"].data_ptr(), st["out_ids"].data_ptr(), st["out_ids"].stride(0),
                              st["lens"].data_ptr(), st["t_rows"].data_ptr(), st["tok"].data_ptr(), st["pos"].data_ptr(),
                              st["nk"].data_ptr(), _p(h1), hid.data_ptr(), hid.stride(0) * hid.element_size(),
                              hid.shape[2] * hid.element_size(), int(pad), int(eos), B, _stream())
    check(rc, "haff_decode_book")


def add_bcast(a, b, mod=None, out=None):
    """out[r] = a[r] + b[r % mod]; a [R,C], b [mod,C]."""
    lib = load_library()
    _req(a, "a")
    assert a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
    R, C = a.shape
    mod = b.shape[0] if mod is None else mod
    if out is None:
        out = torch.empty_like(a)
    rc = lib.haff_add_bcast(a.data_ptr(), b.data_ptr(), out.data_ptr(), R, C, mod, _dt(a), _stream())
    check(rc, "haff_add_bcast")
    return out


def softmax_rows(x):
    lib = load_library()
    _req(x, "x")
    assert x.is_contiguous() and x.dim() == 2
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    rc = lib.haff_softmax_rows(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _dt(x), _stream())
    check(rc, "haff_softmax_rows")
    return out


def upscale_mask(up1, ln_w, ln_b, w2, b2, hyper, n_prompts, h, w, eps=1e-6):
    lib = load_library()
    _req(up1, "up1")
    assert up1.is_contiguous() and up1.shape == (n_prompts * h * w, 256)
    assert hyper.dtype == torch.float32 and hyper.is_contiguous() and hyper.shape == (n_prompts, 32)
    out = torch.empty((n_prompts, 4 * h, 4 * w), dtype=torch.float32, device=up1.device)
    rc = lib.haff_upscale_mask(up1.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                               hyper.data_ptr(), out.data_ptr(), n_prompts, h, w, float(eps), _dt(up1), _stream())
    check(rc, "haff_upscale_mask")
    return out


def resize_bilinear(x, crop_hw, out_hw):
    """x fp32 [N,Hs,Ws]; resample its top-left crop to out_hw (F.interpolate bilinear, align_corners=False)."""
    lib = load_library()
    _req(x, "x")
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    N, Hs, Ws = x.shape
    out = torch.empty((N, out_hw[0], out_hw[1]), dtype=torch.float32, device=x.device)
    rc = lib.haff_resize_bilinear(x.data_ptr(), out.data_ptr(), N, Hs, Ws, int(crop_hw[0]), int(crop_hw[1]),
                                  int(out_hw[0]), int(out_hw[1]), _stream())
    check(rc, "haff_resize_bilinear")
    return out


def threshold_masks(x, logit_th=0.0):
    lib = load_library()
    _req(x, "x")
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    rc = lib.haff_threshold_masks(x.data_ptr(), out.data_ptr(), x.numel(), float(logit_th), _stream())
    check(rc, "haff_threshold_masks")
    return out


def gate_threshold_masks(x, logit_ths, on_value=255, taxonomy=None, blank_class=-1):
    """x fp32 [...] mask logits of ONE prompt -> uint8 [n_th, ...]: (argmax(taxonomy) != blank_class and x > th) ? on_value : 0."""
    import ctypes
    lib = load_library()
    _req(x, "x")
    assert x.dtype == torch.float32 and x.is_contiguous()
    n_th = len(logit_ths)
    total = x.numel()
    stride = (total + 3) // 4 * 4
    buf = torch.empty((n_th, stride), dtype=torch.uint8, device=x.device)
    ths = (ctypes.c_float * n_th)(*[float(v) for v in logit_ths])
    if taxonomy is not None:
        assert taxonomy.dtype == torch.float32 and taxonomy.is_cuda and taxonomy.numel() == 4 and taxonomy.is_contiguous()
    rc = lib.haff_gate_threshold_masks(x.data_ptr(), buf.data_ptr(), total, stride, ctypes.cast(ths, ctypes.c_void_p), n_th,
                                       int(on_value), _p(taxonomy), int(blank_class), _stream())
    check(rc, "haff_gate_threshold_masks")
    return buf[:, :total].reshape((n_th,) + tuple(x.shape))


def resample_u8(frames, out_hw, axis, bounds, coeffs):
    """One axis of Pillow's antialiased resampling on uint8 NHWC frames [B,H,W,3]; bounds/coeffs = device int32 tables."""
    lib = load_library()
    _req(frames, "frames")
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.dim() == 4 and frames.shape[3] == 3
    assert bounds.dtype == torch.int32 and coeffs.dtype == torch.int32 and bounds.is_cuda and coeffs.is_cuda
    B, H, W, _ = frames.shape
    oh, ow = out_hw
    n_out = ow if axis == 0 else oh
    assert bounds.shape == (n_out, 2) and coeffs.shape[0] == n_out and bounds.is_contiguous() and coeffs.is_contiguous()
    out = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=frames.device)
    rc = lib.haff_resample_u8(frames.data_ptr(), out.data_ptr(), B, H, W, oh, ow, axis, bounds.data_ptr(), coeffs.data_ptr(),
                              coeffs.shape[1], _stream())
    check(rc, "haff_resample_u8")
    return out


def clip_normalize_u8(frames, top, left, size, lut, out_dtype):
    """uint8 NHWC [B,H,W,3] window (top, left, size, size) -> [B,3,size,size] through the f32 [3,256] device LUT."""
    lib = load_library()
    _req(frames, "frames")
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.shape[3] == 3
    assert lut.dtype == torch.float32 and lut.is_cuda and lut.shape == (3, 256) and lut.is_contiguous()
    B, H, W, _ = frames.shape
    out = torch.empty((B, 3, size, size), dtype=out_dtype, device=frames.device)
    rc = lib.haff_clip_normalize_u8(frames.data_ptr(), out.data_ptr(), B, H, W, int(top), int(left), int(size), lut.data_ptr(),
                                    _dt(out), _stream())
    check(rc, "haff_clip_normalize_u8")
    return out
