#!/usr/bin/env python3
"""The fused window attention alone at bench size, for rocprofv3 --pmc passes. usage: window_one.py [frames] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
S, d, H = 14, 80, 16
n_win, N = frames * 25, S * S
qkv = torch.randn((n_win * N, 3 * H * d), device=dev).to(torch.bfloat16)
q5 = qkv.view(n_win, N, 3, H, d)
q, k, v = (q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
th = torch.randn((2 * S - 1, d), device=dev) * 0.3
tw = torch.randn((2 * S - 1, d), device=dev) * 0.3
out = torch.empty((n_win, N, H * d), dtype=torch.bfloat16, device=dev)
for _ in range(iters):
    ops.window_attention(q, k, v, d ** -0.5, th, tw, S, out=out)
torch.cuda.synchronize()
print("done")
