#!/usr/bin/env python3
"""One-frame (batch 1) GEMM shapes: Llama-7B prefill at 288 positions and CLIP-L at 257 tokens; ops.linear (auto: split-K
where the library chooses it) vs the unsplit 128x128 tile (tile_cfg=1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

dev = torch.device("cuda:0")
SHAPES = [("llama qkv", 288, 12288, 4096), ("llama o", 288, 4096, 4096), ("llama down", 288, 4096, 11008),
          ("clip qkv", 257, 3072, 1024), ("clip out", 257, 1024, 1024), ("clip fc1", 257, 4096, 1024), ("clip fc2", 257, 1024, 4096),
          ("sam qkv b1", 4096, 3840, 1280), ("sam lin2 b1", 4096, 1280, 5120)]


def timed(fn, n=40):
    """n calls captured in one hipGraph (no host launch cost between the kernels), replayed 5 times."""
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


for name, M, N, K in SHAPES:
    # several weight copies so each call streams its weights from HBM like the layers of a model do
    ws = [torch.randn((N, K), device=dev).to(torch.bfloat16) for _ in range(max(2, int(600e6 // (N * K * 2))))]
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    it = [0]

    def auto():
        it[0] += 1
        return ops.linear(x, ws[it[0] % len(ws)])

    def unsplit():
        it[0] += 1
        return ops.linear(x, ws[it[0] % len(ws)], tile_cfg=1)
    ta, tu = timed(auto), timed(unsplit)
    print(f"{name:12s} {M:5d} x {N:5d} x {K:5d} | auto {ta:7.1f} us  {2.0 * M * N * K / ta / 1e6:7.0f} TF/s {N * K * 2 / ta / 1e6:5.2f} TB/s | 128^2 unsplit {tu:7.1f} us")
