#!/usr/bin/env python3
"""Time haff_gemm_bf16_cfg from an explicitly named build of gemm_bf16.hip (experiment variants compiled with -D flags
into 2handedafforder_amd/lib/libhaff_gemm_<name>.so). usage: [ACT=1] [CFG=2] gemm_variant.py name [name ...]
(RESID=1: residual epilogue. ACT: epilogue activation code of haff_hip.h, with a bias vector; default 0 = plain product. CFG: tile_cfg, 1 = 128x128,
2 = 256x256 8-wave (default), 3 = 256x256 4-wave)"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(131072, 3840, 1280), (131072, 1280, 1280), (131072, 5120, 1280), (131072, 1280, 5120), (18624, 4096, 4096), (18624, 4096, 11008), (16448, 1024, 4096), (16448, 4096, 1024), (65536, 1280, 5120)]


def load(name):
    lib = ctypes.CDLL(os.path.join(HERE, "2handedafforder_amd", "lib", f"libhaff_gemm_{name}.so"))
    vp, cl, ci = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    lib.haff_gemm_bf16_cfg.argtypes = [vp, cl, vp, cl, vp, cl, vp, vp, cl, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    lib.haff_gemm_bf16_cfg.restype = ci
    return lib


def main():
    names = sys.argv[1:]
    libs = {n: load(n) for n in names}
    dev = torch.device("cuda:0")
    for M, N, K in SHAPES:
        pad = int(os.environ.get("LDPAD", "0"))   # row stride of both operands = K + pad elements (L2 channel experiments)
        x = torch.randn((M, K + pad), device=dev).to(torch.bfloat16)
        w = (torch.randn((N, K + pad), device=dev) * K ** -0.5).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        act = int(os.environ.get("ACT", "0"))
        bias = torch.randn((N,), device=dev) if act else None
        resid = torch.randn((M, N), device=dev).to(torch.bfloat16) if os.environ.get("RESID") else None
        cfgs = [int(c) for c in os.environ.get("CFG", "2").split(",")]
        cols = [(n, c) for n in names for c in cfgs]
        res = {k: [] for k in cols}
        for r in range(4):
            for n, cfg_id in cols:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    rc = libs[n].haff_gemm_bf16_cfg(x.data_ptr(), K + pad, w.data_ptr(), K + pad, out.data_ptr(), N,
                                                    bias.data_ptr() if act else None,
                                                    resid.data_ptr() if resid is not None else None, N if resid is not None else 0,
                                                    None, M, N, K, act, 0, 0, cfg_id, None)
                    assert rc == 0
                e1.record()
                torch.cuda.synchronize()
                if r:
                    res[(n, cfg_id)].append(e0.elapsed_time(e1) / 3 * 1e3)
        fl = 2.0 * M * N * K
        print(f"{M:6d} {N:6d} {K:6d} | " + " | ".join(f"{n}/cfg{c}: {sorted(v)[1]:8.1f} us {fl / sorted(v)[1] / 1e6:5.0f}" for (n, c), v in res.items()), flush=True)


if __name__ == "__main__":
    main()
