#!/usr/bin/env python3
"""Throughput of haff_attention_bf16 / haff_relpos_tables on the shapes the 2Haff path launches (random data)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

CASES = [  # name, B, H, Nq, Nk, d, causal, S
    ("sam window  B8", 8 * 25, 16, 196, 196, 80, False, 14),
    ("sam global  B8", 8, 16, 4096, 4096, 80, False, 64),
    ("sam global  B32", 32, 16, 4096, 4096, 80, False, 64),
    ("clip        B64", 64, 16, 257, 257, 64, False, 0),
    ("llama pre   B64", 64, 32, 291, 291, 128, True, 0),
    ("llama dec   B64", 64, 32, 1, 298, 128, False, 0),
    ("dec t2i     P64", 64, 8, 6, 4096, 16, False, 0),
    ("dec i2t     P64", 64, 8, 4096, 6, 16, False, 0),
    ("dec t2i f32 P64", 64, 8, 6, 4096, 16, False, -1),
    ("dec i2t f32 P64", 64, 8, 4096, 6, 16, False, -1),
    ("dec self f32 P64", 64, 8, 6, 6, 32, False, -1),
]


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    for name, B, H, Nq, Nk, d, causal, S in CASES:
        dt = torch.bfloat16
        if S < 0:
            dt, S = torch.float32, 0
        if Nq == Nk:
            qkv = torch.randn((B, Nq, 3, H, d), device=dev).to(dt)
            q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        else:
            q = torch.randn((B, Nq, H, d), device=dev).to(dt).permute(0, 2, 1, 3)
            k = torch.randn((B, Nk, H, d), device=dev).to(dt).permute(0, 2, 1, 3)
            v = torch.randn((B, Nk, H, d), device=dev).to(dt).permute(0, 2, 1, 3)
        relh = relw = None
        line = f"{name:16s}"
        if S:
            th = torch.randn((2 * S - 1, d), device=dev)
            tw = torch.randn((2 * S - 1, d), device=dev)
            t_rel = timeit(lambda: ops.relpos_tables(q, th, tw, S))
            relh, relw = ops.relpos_tables(q, th, tw, S)
            line += f" relpos {t_rel:8.1f} us |"
        t = timeit(lambda: ops.attention(q, k, v, d ** -0.5, causal=causal, q_pos0=Nk - Nq, relh=relh, relw=relw, S=S))
        fl = 4.0 * B * H * Nq * Nk * d * (0.5 if causal else 1.0)
        line += f" attn {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s"
        if S == 64 and ops.global_attention_supported(q, k, v, S):
            tf = timeit(lambda: ops.global_attention(q, k, v, d ** -0.5, th, tw, S))
            line += f" | fused rel-pos + attn {tf:8.1f} us  {fl / tf / 1e6:7.1f} TF/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
