#!/usr/bin/env python3
"""One GEMM shape, a few launches — the target of rocprofv3 --pmc passes.  usage: gemm_one.py M N K [tile_cfg] [iters] [kind]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

M, N, K = (int(v) for v in sys.argv[1:4])
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
kind = sys.argv[6] if len(sys.argv) > 6 else "bias"
dev = torch.device("cuda:0")
x = torch.randn((M, K), device=dev).to(torch.bfloat16)
w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
bias = torch.randn((N,), device=dev) if kind in ("bias", "gelu") else None
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
for _ in range(iters):
    ops.linear(x, w, bias=bias, act=1 if kind == "gelu" else 0, out=out, tile_cfg=cfg)
torch.cuda.synchronize()
print("done", M, N, K, cfg)
