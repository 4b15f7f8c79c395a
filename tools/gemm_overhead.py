#!/usr/bin/env python3
"""Fixed per-tile cost of the GEMM kernel: time vs K at constant M, N (intercept = prologue + epilogue)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

dev = torch.device("cuda:0")
for (M, N, kind) in ((65536, 5120, "gelu"), (78400, 3840, "bias"), (65536, 1280, "resid")):
    for cfg in (1, 2):
        line = f"M{M} N{N} {kind} cfg{cfg}: "
        for K in (64, 128, 256, 640, 1280, 2560):
            x = torch.randn((M, K), device=dev).to(torch.bfloat16)
            w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
            bias = torch.randn((N,), device=dev)
            out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            resid = torch.randn((M, N), device=dev).to(torch.bfloat16) if kind == "resid" else None
            act = 1 if kind == "gelu" else 0
            for _ in range(2):
                ops.linear(x, w, bias=bias, act=act, resid=resid, out=out, tile_cfg=cfg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(x, w, bias=bias, act=act, resid=resid, out=out, tile_cfg=cfg)
            e1.record()
            torch.cuda.synchronize()
            line += f"K{K}={e0.elapsed_time(e1) / 5 * 1e3:.0f}us "
        print(line, flush=True)
