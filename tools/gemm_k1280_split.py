#!/usr/bin/env python3
"""Where does a launch of the four ViT-H block products go (image_encoder.py:177-193, 223-258; common.py:13-26)? Each product is
timed through its REAL entry point and epilogue at the bench shape (32 frames: 131072 token rows) from experiment builds of
gemm_bf16.hip (tools/build_gemm_variant.sh <name> -DHAFF_EXP_...):

    qkv   131072 x 3840 x 1280  haff_gemm_bf16_ln, folded LayerNorm, rows scattered into the window-major layout (row_map)
    proj  131072 x 1280 x 1280  haff_gemm_bf16_rowstats, A rows gathered back from the window layout, residual, statistics
    lin1  131072 x 5120 x 1280  haff_gemm_bf16_ln, folded LayerNorm, GELU
    lin2  131072 x 1280 x 5120  haff_gemm_bf16_rowstats, residual, statistics
    (+ the global blocks' qkv / proj: no maps)

usage: gemm_k1280_split.py base nostore noepi nodma ...   (timing-only builds compute wrong results; `base` is checked)"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp, cl, ci = ctypes.c_void_p, ctypes.c_long, ctypes.c_int


def load(name):
    lib = ctypes.CDLL(os.path.join(HERE, "2handedafforder_amd", "lib", f"libhaff_gemm_{name}.so"))
    lib.haff_gemm_bf16_ln.argtypes = [vp, cl, vp, cl, vp, cl, vp, vp, cl, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.haff_gemm_bf16_ln.restype = ci
    lib.haff_gemm_bf16_rowstats.argtypes = [vp, cl, vp, cl, vp, cl, vp, cl, vp, vp, cl, ci, ci, ci, vp, vp]
    lib.haff_gemm_bf16_rowstats.restype = ci
    return lib


def window_map(B, g=64, w=14):
    nw = (g + w - 1) // w
    b = torch.arange(B).view(B, 1, 1)
    y = torch.arange(g).view(1, g, 1)
    x = torch.arange(g).view(1, 1, g)
    dest = ((b * nw * nw + (y // w) * nw + (x // w)) * (w * w) + (y % w) * w + (x % w))
    return dest.reshape(-1).to(torch.int32), B * nw * nw * w * w


def main():
    names = sys.argv[1:] or ["base"]
    libs = {n: load(n) for n in names}
    dev = torch.device("cuda:0")
    B, C = 32, 1280
    M = B * 4096
    inv, n_win_rows = window_map(B)
    inv = inv.to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape, s=1.0):
        return (torch.randn(shape, generator=g) * s).to(dev)
    x = rnd(M, C).to(torch.bfloat16)
    stats = torch.stack([rnd(M) * 0.1, 1 + 0.1 * rnd(M).abs()], 1).contiguous()
    cases = []
    for name, N, K, kind, use_map in (("qkv  (windowed: ln fold, scatter)", 3 * C, C, "ln", True), ("qkv  (global: ln fold)", 3 * C, C, "ln", False),
                                      ("proj (windowed: gather, resid, stats)", C, C, "rs", True), ("proj (global: resid, stats)", C, C, "rs", False),
                                      ("lin1 (ln fold, GELU)", 4 * C, C, "gelu", False), ("lin2 (resid, stats)", C, 4 * C, "rs", False)):
        cases.append((name, N, K, kind, use_map))
    for name, N, K, kind, use_map in cases:
        w = rnd(N, K, s=K ** -0.5).to(torch.bfloat16)
        bias = rnd(N)
        if kind in ("ln", "gelu"):
            a = x
            colsum = w.float().sum(1).contiguous()
            out_rows = n_win_rows if use_map else M
            out = torch.empty((out_rows, N), dtype=torch.bfloat16, device=dev)

            def call(lib):
                return lib.haff_gemm_bf16_ln(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, bias.data_ptr(), None, 0,
                                             inv.data_ptr() if use_map else None, stats.data_ptr(), colsum.data_ptr(), M, N, K,
                                             1 if kind == "gelu" else 0, 0, 0, None)   # act code 1 = HAFF_ACT_GELU
        else:
            a_rows = n_win_rows if use_map else M
            a = rnd(a_rows, K).to(torch.bfloat16) if (use_map or K != C) else x
            out = rnd(M, N).to(torch.bfloat16)    # residual stream, updated in place
            so = torch.empty((M, N // 64, 2), dtype=torch.float32, device=dev)

            def call(lib):
                return lib.haff_gemm_bf16_rowstats(a.data_ptr(), K, inv.data_ptr() if use_map else None, a_rows, w.data_ptr(), K,
                                                   out.data_ptr(), N, bias.data_ptr(), out.data_ptr(), N, M, N, K, so.data_ptr(), None)
        res = {n: [] for n in names}
        for r in range(4):
            for n in names:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    rc = call(libs[n])
                    assert rc == 0, (name, n, rc)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    res[n].append(e0.elapsed_time(e1) / 3 * 1e3)
        fl = 2.0 * M * N * K
        med = {n: sorted(v)[1] for n, v in res.items()}
        line = f"{name:40s} {M}x{N}x{K} | " + " | ".join(f"{n}: {med[n]:7.1f} us {fl / med[n] / 1e6:5.0f}" for n in names)
        if "base" in med:
            b = med["base"]
            extra = []
            if "noepi" in med:
                extra.append(f"epilogue {100 * (b - med['noepi']) / b:4.1f} %")
            if "nostore" in med:
                extra.append(f"its stores {100 * (b - med['nostore']) / b:4.1f} %")
            if "nodma" in med:
                extra.append(f"operand requests {100 * (b - med['nodma']) / b:4.1f} %")
            tiles = (M // 256) * (N // 256)
            extra.append(f"{tiles} tiles = {tiles / 256:.2f} rounds, {b / (tiles / 256):.1f} us per round")
            line += " | " + ", ".join(extra)
        print(line, flush=True)
        del w, out


if __name__ == "__main__":
    main()
