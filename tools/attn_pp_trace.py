#!/usr/bin/env python3
"""Slot timeline of attn_global_pp_kernel (lib/libhaff_attn_trace.so: tools/build_attn_variant.sh trace -DHAFF_PP_TRACE).
Waves 0 (group 0) and 4 (group 1) of the first 256 workgroups stamp the shader clock in KV tiles 16..19:
0 MFMA slot start | 1 MFMAs issued | 2 my requests + LDS reads done | 3 past the barrier (VALU slot start) | 4 requests issued |
5 softmax done | 6 past the barrier. Prints the median cycles between consecutive stamps and per whole tile."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dev = torch.device("cuda:0")
    B, H, N, d, S = int(os.environ.get("B", "32")), 16, 4096, 80, 64
    g = torch.Generator(device="cpu").manual_seed(3)
    qkv = torch.randn((B, N, 3, H, d), generator=g).to(torch.bfloat16).to(dev)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    relh = torch.randn((B * H, N, S), generator=g).to(dev)
    relw = torch.randn((B * H, N, S), generator=g).to(dev)
    out = torch.empty((B, N, H * d), dtype=torch.bfloat16, device=dev)
    o4 = out.view(B, N, H, d).permute(0, 2, 1, 3)
    lib = ctypes.CDLL(os.path.join(ROOT, "2handedafforder_amd", "lib", "libhaff_attn_%s.so" % os.environ.get("TRACELIB", "trace")))
    fn = lib.haff_attention_bf16
    L, P, I, F = ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    fn.argtypes = [P, L, L, L, P, L, L, L, P, L, L, L, P, L, L, L, I, I, I, I, I, F, I, I, P, P, I, P]
    lib.haff_pp_trace_read.argtypes = [P, I]
    args = [q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
            v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), out.data_ptr(), o4.stride(0), o4.stride(1), o4.stride(2),
            B, H, N, N, d, d ** -0.5, 0, 0, relh.data_ptr(), relw.data_ptr(), S, torch.cuda.current_stream().cuda_stream]
    for _ in range(2):
        assert fn(*args) == 0
    torch.cuda.synchronize()
    buf = np.zeros(256 * 2 * 4 * 8, dtype=np.uint64)
    assert lib.haff_pp_trace_read(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(256, 2, 4, 8).astype(np.int64)
    names = ["mfma issue", "wait reqs+lds", "barrier A", "dma issue", "softmax", "barrier B"]
    for grp in (0, 1):
        d = np.diff(t[:, grp, :, :7], axis=-1).reshape(-1, 6)
        med = np.median(d, axis=0)
        whole = np.median(t[:, grp, 1:, 0] - t[:, grp, :-1, 0])
        print(f"group {grp}: " + " | ".join(f"{n} {m:6.0f}" for n, m in zip(names, med)) + f" || tile {whole:6.0f} cycles", flush=True)
    # offset between the groups: group 1's MFMA-slot start relative to group 0's VALU-slot start (same workgroup, same tile)
    off = np.median(t[:, 1, :, 0] - t[:, 0, :, 3])
    print(f"group 1 MFMA-slot start minus group 0 VALU-slot start: {off:.0f} cycles (0 = the two run exactly one slot apart)")


if __name__ == "__main__":
    main()
