#!/usr/bin/env python3
"""Calibration only: what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the same shapes, to know how far
the hand-written kernel is from what the silicon sustains. Never used by the product path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

SHAPES = [(78400, 3840, 1280), (65536, 5120, 1280), (65536, 1280, 5120), (18624, 12288, 4096), (18624, 4096, 11008),
          (8192, 8192, 8192), (64, 12288, 4096), (16448, 4096, 1024)]


def t_us(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3 * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    for M, N, K in SHAPES:
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * N * K
        a = t_us(lambda: ops.linear(x, w, out=out))
        b = t_us(lambda: torch.matmul(x, w.t(), out=out))
        print(f"{M:6d} {N:6d} {K:6d} | haff {a:9.1f} us {fl / a / 1e6:6.0f} TF/s | vendor {b:9.1f} us {fl / b / 1e6:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
