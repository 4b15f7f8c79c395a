#!/usr/bin/env python3
"""A few plain launches of ops.linear on one shape (default: the one-frame Llama q|k|v product) — the target of rocprofv3 --pmc
passes on the weight-stationary tile.  usage: mid_one.py [M N K iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (288, 12288, 4096)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda:0")
ws = [torch.randn((N, K), device=dev).to(torch.bfloat16) for _ in range(iters)]
x = torch.randn((M, K), device=dev).to(torch.bfloat16)
for w in ws:
    ops.linear(x, w)
torch.cuda.synchronize()
print("ok")
