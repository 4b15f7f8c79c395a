#!/usr/bin/env python3
"""Times haff_window_attention_bf16 at bench size (FRAMES x 25 windows x 16 heads, S = 14, d = 80, token-major q|k|v views) from
experiment builds of window_attention.hip (tools/build_window_variant.sh), one variant per process so that a rocprofv3 --pmc pass sees
one kernel build. usage: VARIANT=base|nokread|novread|noscr|nodma [FRAMES=32] python tools/window_variant.py
The variant is checked against the product library's output (max abs difference printed; ablations are EXPECTED to differ)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import haff  # noqa
from haff import ops


def main():
    dev = torch.device("cuda:0")
    frames, name = int(os.environ.get("FRAMES", "32")), os.environ.get("VARIANT", "base")
    S, d, H = 14, 80, 16
    n_win, N = frames * 25, S * S
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = torch.randn((n_win * N, 3 * H * d), generator=g).to(torch.bfloat16).to(dev)
    q5 = qkv.view(n_win, N, 3, H, d)
    q, k, v = (q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    th = (torch.randn((2 * S - 1, d), generator=g) * 0.3).to(dev)
    tw = (torch.randn((2 * S - 1, d), generator=g) * 0.3).to(dev)
    ref = ops.window_attention(q, k, v, d ** -0.5, th, tw, S)
    thb, twb = th.to(torch.bfloat16).contiguous(), tw.to(torch.bfloat16).contiguous()
    out = torch.empty_like(ref)
    o4 = out.view(n_win, N, H, d).permute(0, 2, 1, 3)
    lib = ctypes.CDLL(os.path.join(ROOT, "2handedafforder_amd", "lib", f"libhaff_win_{name}.so"))
    fn = lib.haff_window_attention_bf16
    fn.restype = ctypes.c_int
    L, P, I, F = ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    fn.argtypes = [P, L, L, L, P, L, L, L, P, L, L, L, P, L, L, L, I, I, I, I, F, P, P, I, I, L, P]
    args = [q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
            v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), out.data_ptr(), o4.stride(0), o4.stride(1), o4.stride(2),
            n_win, H, S, d, d ** -0.5, thb.data_ptr(), twb.data_ptr(), 0, 0, 0, torch.cuda.current_stream().cuda_stream]

    def run():
        rc = fn(*args)
        assert rc == 0, rc
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e3
    diff = (out.float() - ref.float()).abs().max().item()
    print(f"{name:10s} {t:8.1f} us per {frames}-frame launch   max|diff| vs product {diff:.3e}", flush=True)


if __name__ == "__main__":
    main()
