#!/usr/bin/env python3
"""The weight-stationary tile (65..320 rows) on the one-frame Llama shapes: weights with their natural row stride (a power of two
at K = 4096) against rows padded by 64 elements, and the time of the reduce pass alone.  usage: mid_gemm_probe.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 288


def timed(fn, n=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3


for name, N, K, sw in (("qkv", 12288, 4096, False), ("o", 4096, 4096, False), ("gate|up", 22016, 4096, True), ("down", 4096, 11008, False)):
    for pad in (0, 64):
        n_w = max(2, int(600e6 // (N * (K + pad) * 2)))
        ws = [torch.randn((N, K + pad), device=dev).to(torch.bfloat16)[:, :K] for _ in range(n_w)]
        xpad = int(os.environ.get("XPAD", "0"))
        x = torch.randn((M, K + xpad), device=dev).to(torch.bfloat16)[:, :K]
        it = [0]

        def auto():
            it[0] += 1
            return ops.linear(x, ws[it[0] % len(ws)], swiglu=sw)

        def unsplit():
            it[0] += 1
            return ops.linear(x, ws[it[0] % len(ws)], swiglu=sw, tile_cfg=1)
        ta, tu = timed(auto), timed(unsplit)
        print(f"{name:8s} {M} x {N} x {K}  weight row stride {K + pad}: auto {ta:7.1f} us  {N * K * 2 / ta / 1e6:5.2f} TB/s  {2.0 * M * N * K / ta / 1e6:6.0f} TF/s | 128^2 unsplit {tu:7.1f} us", flush=True)
