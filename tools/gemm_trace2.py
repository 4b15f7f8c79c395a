#!/usr/bin/env python3
"""Fine timeline of the ring-loop GEMM's epilogue (lib/libhaff_gemm_trace2.so: gemm_bf16.hip built with -DHAFF_GEMM_TRACE2):
per workgroup, waves 0 (group 0) and 4 (group 1) stamp the 100 MHz clock at the end of their K loop (0), before pass 0 (1),
after each of the 8 register-epilogue passes (2..9), before / after the barrier behind the epilogue (10, 11) — of the
workgroup's next-to-last tile (stamps 10/11) and last tile (0..9). Prints medians in us relative to stamp 0 of wave 0."""
import ctypes, os, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(HERE, "2handedafforder_amd", "lib", "libhaff_gemm_%s.so" % os.environ.get("TRACELIB", "trace2")))
vp, cl, ci = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
lib.haff_gemm_bf16_cfg.argtypes = [vp, cl, vp, cl, vp, cl, vp, vp, cl, vp, ci, ci, ci, ci, ci, ci, ci, vp]
lib.haff_gemm_trace2_read.argtypes = [vp, ci]
SHAPES = [("K1280 plain", 131072, 3840, 1280, "none"), ("K1280 bias", 131072, 3840, 1280, "bias"), ("K1280 gelu", 131072, 5120, 1280, "gelu"),
          ("K1280 resid", 131072, 1280, 1280, "resid"), ("K4096 plain", 18624, 12288, 4096, "none")]
dev = torch.device("cuda:0")
for name, M, N, K, kind in SHAPES:
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    bias = torch.randn((N,), device=dev) if kind in ("bias", "gelu") else None
    resid = torch.randn((M, N), device=dev).to(torch.bfloat16) if kind == "resid" else None
    for _ in range(3):
        rc = lib.haff_gemm_bf16_cfg(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, bias.data_ptr() if bias is not None else None,
                                    resid.data_ptr() if resid is not None else None, N, None, M, N, K, 1 if kind == "gelu" else 0, 0, 0, 2, None)
        assert rc == 0
    torch.cuda.synchronize()
    buf = np.zeros(256 * 2 * 16, dtype=np.uint64)
    assert lib.haff_gemm_trace2_read(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(256, 2, 16).astype(np.int64)
    base = t[:, 0:1, 0:1]
    d = (t - base) / 100.0
    for wv in (0, 1):
        med = np.median(d[:, wv, :12], axis=0)
        print(f"{name:14s} wave {4 * wv}: kloop-end {med[0]:6.2f} | pre {med[1]:6.2f} | passes " + " ".join(f"{v:6.2f}" for v in med[2:10]) +
              f" | (prev tile) before/after barrier {med[10]:8.2f} {med[11]:8.2f}", flush=True)
