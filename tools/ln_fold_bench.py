#!/usr/bin/env python3
"""Cost of carrying LayerNorm into the consumer GEMM (haff_gemm_bf16_ln) on the SAM shapes: norm kernel + plain product
vs row_stats + folded product vs folded product alone (statistics given)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

dev = torch.device("cuda:0")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M, N, K, act in ((131072, 3840, 1280, 0), (131072, 5120, 1280, 1)):
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
    g, b = 1 + 0.1 * torch.randn((K,), device=dev), 0.1 * torch.randn((K,), device=dev)
    bias = torch.randn((N,), device=dev)
    wf, cs, bf = ops.fold_norm(w, g, b, bias)
    st = ops.row_stats(x, 1e-6)
    t_norm = timed(lambda: ops.layernorm(x, g, b, 1e-6))
    t_plain = timed(lambda: ops.linear(x, w, bias=bias, act=act))
    t_stats = timed(lambda: ops.row_stats(x, 1e-6))
    t_fold = timed(lambda: ops.linear(x, wf, bias=bf, act=act, ln_stats=st, ln_colsum=cs))
    print(f"{M}x{N}x{K} act{act}: norm {t_norm:.1f} us, plain product {t_plain:.1f}, row_stats {t_stats:.1f}, folded product {t_fold:.1f} "
          f"(fold costs {t_fold - t_plain:+.1f}); norm+plain {t_norm + t_plain:.1f} vs stats+folded {t_stats + t_fold:.1f} vs folded alone {t_fold:.1f}")
