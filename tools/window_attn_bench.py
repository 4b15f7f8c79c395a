#!/usr/bin/env python3
"""SAM ViT-H window attention at bench size (16 frames x 25 windows, 16 heads, 14x14 tokens, d=80):
fused kernel (haff_window_attention_bf16) vs the generic pair it replaces (rel-pos tables + flash attention).
Reports time and the HBM roofline fraction (algorithmic bytes: qkv read once + out written once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops


def t_us(fn, n=5, inner=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    frames = int(os.environ.get("FRAMES", "16"))
    S, d, H = 14, 80, 16
    n_win, N = frames * 25, S * S
    qkv = (torch.randn((n_win * N, 3 * H * d), device=dev) * 1.0).to(torch.bfloat16)
    q5 = qkv.view(n_win, N, 3, H, d)
    q, k, v = (q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    th = torch.randn((2 * S - 1, d), device=dev) * 0.3
    tw = torch.randn((2 * S - 1, d), device=dev) * 0.3
    scale = d ** -0.5
    out = torch.empty((n_win, N, H * d), dtype=torch.bfloat16, device=dev)

    def fused():
        ops.window_attention(q, k, v, scale, th, tw, S, out=out)

    def generic():
        rh, rw = ops.relpos_tables(q, th, tw, S)
        ops.attention(q, k, v, scale, relh=rh, relw=rw, S=S, out=out)

    # round 5: the same items with q, k, v as HEAD-MAJOR planes [3][window][head][token][d] (what haff_gemm_bf16_heads writes): an
    # item's K and V are one contiguous 31 KB block each instead of 196 pieces of 160 B at a 7680-B stride
    planes = torch.empty((3, n_win + 1, H, N, d), dtype=torch.bfloat16, device=dev)
    for i in range(3):
        planes[i, :n_win] = q5[:, :, i].permute(0, 2, 1, 3)
    out_hm = torch.empty_like(out)

    def fused_hm():
        ops.window_attention(planes[0, :n_win], planes[1, :n_win], planes[2, :n_win], scale, th, tw, S, out=out_hm)

    c = t_us(fused_hm)
    fused()
    torch.cuda.synchronize()
    same = torch.equal(out, out_hm)
    a, b = t_us(fused), t_us(generic)
    print(f"windows {n_win} heads {H}: head-major planes {c:8.1f} us ({(qkv.numel() * 2 + out.numel() * 2) / c / 1e6:6.2f} TB/s algorithmic), "
          f"bit-identical to token-major: {same}", flush=True)
    byts = qkv.numel() * 2 + out.numel() * 2
    fl = 4.0 * N * N * d * H * n_win
    print(f"windows {n_win} heads {H}: fused {a:8.1f} us ({byts / a / 1e6:6.2f} TB/s algorithmic, {fl / a / 1e6:6.0f} TF/s) | "
          f"generic pair {b:8.1f} us | speed-up {b / a:4.2f}x", flush=True)


if __name__ == "__main__":
    main()
