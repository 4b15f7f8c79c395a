#!/usr/bin/env python3
"""Split the fixed GEMM cost: K=64 launches with/without activation, with all rows dropped (no global stores)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

dev = torch.device("cuda:0")
M, N = 65536, 5120
for K in (64, 1280):
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    drop = torch.full((M,), -1, dtype=torch.int32, device=dev)
    for cfg in (1, 2):
        res = []
        for name, kw in (("plain", {}), ("bias", dict(bias=bias)), ("gelu", dict(bias=bias, act=1)), ("nostore", dict(bias=bias, row_map=drop)),
                         ("f32out", dict(bias=bias, out_dtype=torch.float32))):
            o = out if name != "f32out" else torch.empty((M, N), dtype=torch.float32, device=dev)
            for _ in range(2):
                ops.linear(x, w, out=o, tile_cfg=cfg, **{k: v for k, v in kw.items() if k != "out_dtype"})
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(x, w, out=o, tile_cfg=cfg, **{k: v for k, v in kw.items() if k != "out_dtype"})
            e1.record()
            torch.cuda.synchronize()
            res.append(f"{name}={e0.elapsed_time(e1) / 5 * 1e3:.0f}us")
        print(f"K{K} cfg{cfg}: " + " ".join(res), flush=True)
