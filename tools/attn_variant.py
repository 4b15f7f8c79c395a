#!/usr/bin/env python3
"""Times haff_attention_bf16 on the SAM global shape (B x 16 heads x 4096 x 4096, d = 80, rel-pos tables) from experiment
builds of attention.hip (tools/build_attn_variant.sh). usage: VARIANTS=base,nodma,... [B=32] python tools/attn_variant.py
Each variant is checked against the product library's output (max abs difference printed; ablations are EXPECTED to differ)."""
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
    B, H, N, d, S = int(os.environ.get("B", "32")), 16, 4096, 80, 64
    g = torch.Generator(device="cpu").manual_seed(3)
    qkv = torch.randn((B, N, 3, H, d), generator=g).to(torch.bfloat16).to(dev)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    relh = torch.randn((B * H, N, S), generator=g).to(dev)
    relw = torch.randn((B * H, N, S), generator=g).to(dev)
    ref = ops.attention(q, k, v, d ** -0.5, relh=relh, relw=relw, S=S)
    torch.cuda.synchronize()
    out = torch.empty_like(ref)
    o4 = out.view(B, N, H, d).permute(0, 2, 1, 3)
    for name in os.environ.get("VARIANTS", "base").split(","):
        if name == "old":     # the 4-wave kernel, through the tuning build's switch
            os.environ["HAFF_ATTN_NO_PP"] = "1"
        else:
            os.environ.pop("HAFF_ATTN_NO_PP", None)
        lib = ctypes.CDLL(os.path.join(ROOT, "2handedafforder_amd", "lib", f"libhaff_attn_{'base' if name == 'old' else name}.so"))
        fn = lib.haff_attention_bf16
        fn.restype = ctypes.c_int
        L, P, I, F = ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        fn.argtypes = [P, L, L, L, P, L, L, L, P, L, L, L, P, L, L, L, I, I, I, I, I, F, I, I, P, P, I, P]
        args = [q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
                v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), out.data_ptr(), o4.stride(0), o4.stride(1), o4.stride(2),
                B, H, N, N, d, d ** -0.5, 0, 0, relh.data_ptr(), relw.data_ptr(), S, torch.cuda.current_stream().cuda_stream]

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
        print(f"{name:16s} {t:9.1f} us  {4.0 * B * H * N * N * d / t / 1e6:7.1f} TF/s   max|diff| vs product {diff:.3e}", flush=True)


if __name__ == "__main__":
    main()
