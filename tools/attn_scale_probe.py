#!/usr/bin/env python3
"""Does the ping-pong global attention slow down when the scores have a wide range (its lazy softmax reference moves when a score
lands 40 log2 units above it)? Times haff_global_attention_bf16 on q, k scaled by s: logits scale with s^2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops
dev = torch.device("cuda:0")
B, H, N, d, S = 32, 16, 4096, 80, 64
th, tw = torch.randn((2 * S - 1, d), device=dev) * 0.1, torch.randn((2 * S - 1, d), device=dev) * 0.1
for s in (0.5, 1.0, 2.0, 3.0, 4.0, 6.0):
    qkv = (torch.randn((B, N, 3, H, d), device=dev) * s).to(torch.bfloat16)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    for _ in range(2):
        ops.global_attention(q, k, v, d ** -0.5, th, tw, S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        o = ops.global_attention(q, k, v, d ** -0.5, th, tw, S)
    e1.record(); torch.cuda.synchronize()
    sc = (q[0, 0, :64].float() @ k[0, 0].float().t()) * d ** -0.5 * 1.4427
    print(f"input scale {s}: {e0.elapsed_time(e1) / 4 * 1e3:8.1f} us   row (max - first-tile max) log2 units: {(sc.max(1).values - sc[:, :64].max(1).values).max().item():6.1f}   finite {bool(torch.isfinite(o.float()).all())}", flush=True)
