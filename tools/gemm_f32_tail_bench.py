#!/usr/bin/env python3
"""The fp32 decoder tail's products (64 prompts: image-side 262144-row projections, token-side few-row ones) through
ops.linear on fp32 tensors: time per launch, TFLOP/s, GB/s of operands + output."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
shapes = [(262144, 128, 256), (262144, 256, 128), (262144, 256, 256), (448, 256, 256), (448, 2048, 256), (448, 256, 2048), (448, 128, 256),
          (64, 256, 256), (16384, 256, 256), (4096, 256, 256)]
for M, N, K in shapes:
    x = torch.randn((M, K), device=dev, generator=g)
    w = torch.randn((N, K), device=dev, generator=g) * K ** -0.5
    b = torch.randn((N,), device=dev, generator=g)
    out = torch.empty((M, N), device=dev)
    for _ in range(3):
        ops.linear(x, w, bias=b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.linear(x, w, bias=b, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    ref = x @ w.T + b
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    print("%7d x %5d x %5d  %8.1f us  %6.1f TFLOP/s  %6.0f GB/s  rel err %.1e" % (M, N, K, us, 2.0 * M * N * K / us / 1e6, 4.0 * (M * K + N * K + M * N) / us / 1e3, err))
