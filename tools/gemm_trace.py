#!/usr/bin/env python3
"""Phase timeline of the bf16 GEMM tile (needs lib/libhaff_gemm_trace.so: gemm_bf16.hip built with
-DHAFF_GEMM_TRACE). Prints, per shape, the median duration of: entry -> first K-tile landed, K loop, epilogue issue,
store drain; plus the spread of workgroup start times. 100 MHz wall clock -> 10 ns ticks."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(HERE, "2handedafforder_amd", "lib", "libhaff_gemm_%s.so" % os.environ.get("TRACELIB", "trace")))
vp, cl, ci = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
lib.haff_gemm_bf16_cfg.argtypes = [vp, cl, vp, cl, vp, cl, vp, vp, cl, vp, ci, ci, ci, ci, ci, ci, ci, vp]
lib.haff_gemm_bf16_cfg.restype = ci
lib.haff_gemm_trace_read.argtypes = [vp, ci]
lib.haff_gemm_trace_read.restype = ci

SHAPES = [
    ("K64 plain", 65536, 5120, 64, "none"),
    ("K64 bias", 65536, 5120, 64, "bias"),
    ("K1280 bias (sam qkv)", 78400, 3840, 1280, "bias"),
    ("K1280 gelu (sam lin1)", 65536, 5120, 1280, "gelu"),
    ("K1280 resid (sam proj)", 78400, 1280, 1280, "resid"),
    ("K5120 resid (sam lin2)", 65536, 1280, 5120, "resid"),
    ("K4096 llama qkv", 18624, 12288, 4096, "none"),
]


def main():
    dev = torch.device("cuda:0")
    for cfg in [int(c) for c in os.environ.get("CFGS", "2,1").split(",")]:
        bm = 128 if cfg == 1 else 256
        for name, M, N, K, kind in SHAPES:
            x = torch.randn((M, K), device=dev).to(torch.bfloat16)
            w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
            out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            bias = torch.randn((N,), device=dev) if kind in ("bias", "gelu") else None
            resid = torch.randn((M, N), device=dev).to(torch.bfloat16) if kind == "resid" else None
            act = 1 if kind == "gelu" else 0
            for _ in range(3):
                rc = lib.haff_gemm_bf16_cfg(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N,
                                            bias.data_ptr() if bias is not None else None,
                                            resid.data_ptr() if resid is not None else None, N, None,
                                            M, N, K, act, 0, 0, cfg, None)
                assert rc == 0
            torch.cuda.synchronize()
            tiles = ((M + bm - 1) // bm) * ((N + bm - 1) // bm)
            nb = min(tiles, 8192)
            if cfg != 1:   # persistent 8-wave tile: one record per workgroup (its LAST tile)
                nb = min(nb, int(os.environ.get("HAFF_GEMM_PERSIST", "256")) or nb)
            buf = np.zeros(nb * 8, dtype=np.uint64)
            assert lib.haff_gemm_trace_read(buf.ctypes.data, nb * 8) == 0
            t = buf.reshape(nb, 8).astype(np.int64)
            d = np.diff(t[:, :5], axis=1) / 100.0  # us
            med = np.median(d, axis=0)
            span = (t[:, 4].max() - t[:, 0].min()) / 100.0
            per_slot = min(nb, 512 if cfg == 1 else 256)
            first = t[:per_slot]
            print(f"cfg{cfg} {name:24s} tiles {tiles:5d} | load {med[0]:6.2f} kloop {med[1]:6.2f} epi-issue {med[2]:6.2f} "
                  f"drain {med[3]:6.2f} us | tile total {np.median(t[:, 4] - t[:, 0]) / 100.0:6.2f} | "
                  f"first-wave start spread {(first[:, 0].max() - first[:, 0].min()) / 100.0:6.2f} | span {span:8.1f}",
                  flush=True)


if __name__ == "__main__":
    main()
