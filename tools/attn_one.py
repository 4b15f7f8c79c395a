#!/usr/bin/env python3
"""The SAM global-attention launch of the bench (32 frames x 16 heads, 4096 x 4096, d = 80, decomposed rel-pos) a few times —
the target of rocprofv3 --pmc passes.  usage: [FUSED=1] attn_one.py [frames] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
H, N, d, S = 16, 4096, 80, 64
qkv = (torch.randn((B, N, 3, H, d), device=dev) * 0.5).to(torch.bfloat16)
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
relh = torch.randn((B * H, N, S), device=dev) * 0.1
relw = torch.randn((B * H, N, S), device=dev) * 0.1
th, tw = torch.randn((2 * S - 1, d), device=dev) * 0.1, torch.randn((2 * S - 1, d), device=dev) * 0.1
for _ in range(iters):
    if os.environ.get("FUSED"):     # haff_global_attention_bf16: rel-pos from the parameter tables inside the kernel
        ops.global_attention(q, k, v, d ** -0.5, th, tw, S)
    else:
        ops.attention(q, k, v, d ** -0.5, relh=relh, relw=relw, S=S)
torch.cuda.synchronize()
print("done")
