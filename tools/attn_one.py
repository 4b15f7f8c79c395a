#!/usr/bin/env python3
"""One attention shape for rocprofv3 --pmc passes. usage: attn_one.py B H N d S [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

B, H, N, d, S = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda:0")
qkv = torch.randn((B, N, 3, H, d), device=dev).to(torch.bfloat16)
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
relh = relw = None
if S:
    th = torch.randn((2 * S - 1, d), device=dev)
    tw = torch.randn((2 * S - 1, d), device=dev)
    relh, relw = ops.relpos_tables(q, th, tw, S)
for _ in range(iters):
    ops.attention(q, k, v, d ** -0.5, relh=relh, relw=relw, S=S)
torch.cuda.synchronize()
print("done")
