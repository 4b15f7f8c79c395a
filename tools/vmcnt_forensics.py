#!/usr/bin/env python3
"""Forensics for DESIGN.md section 10a item 1 (round 3): was the round-1 "counted vmcnt(8) over LDS-DMA" failure of the
128x128 GEMM tile an out-of-order retirement of LDS-DMA, or a write-after-read race?

Two builds of the round-1 source (git c1e704c^, built in /tmp, never committed) are loaded by name from
2handedafforder_amd/lib/libhaff_gemm_<name>.so:
  old     the loop as it failed: tile kt+1's DMA issued at the top of iteration kt, s_waitcnt vmcnt(8), barrier, reads,
          MFMAs, end barrier — hipcc hoists that end barrier ABOVE the s_waitcnt lgkmcnt(0) of the last two ds_read_b128
  oldfix  the same loop, same vmcnt(8), plus one explicit s_waitcnt lgkmcnt(0) before the end barrier
The victim product has A = 1 + k // 64 (K-tile kt holds kt + 1, exact in bf16) and W = 1: a 32-deep k-step fragment that still
holds K-tile kt-2 (stale: a DMA that landed late) lowers an output by 64, one that already holds K-tile kt+2 (overwritten
early: write-after-read) raises it by 64. The round-1 stress used a period-4 pattern, which cannot tell the two apart.
usage: python tools/vmcnt_forensics.py old oldfix"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import haff  # noqa: E402
from haff import ops  # noqa: E402


def load(name):
    lib = ctypes.CDLL(os.path.join(HERE, "2handedafforder_amd", "lib", f"libhaff_gemm_{name}.so"))
    vp, cl, ci = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    lib.haff_gemm_bf16_cfg.argtypes = [vp, cl, vp, cl, vp, cl, vp, vp, cl, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    lib.haff_gemm_bf16_cfg.restype = ci
    return lib


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    M = 9800
    x = torch.randn(M, 1280, device=dev).to(torch.bfloat16)
    wqkv = (torch.randn(3840, 1280, device=dev) * 0.05).to(torch.bfloat16)
    bqkv = torch.randn(3840, device=dev) * 0.05
    lw, lb = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
    st = {}

    def ln():
        st["h"] = ops.layernorm(x, lw, lb, 1e-6)

    def qkv():
        st["qkv"] = ops.linear(st.get("h", x), wqkv, bias=bqkv)

    k = torch.arange(4096, device=dev)
    A = (1 + k // 64).to(torch.bfloat16)[None, :].expand(592, 4096).contiguous()
    W = torch.ones(4096, 4096, dtype=torch.bfloat16, device=dev)
    side = torch.cuda.Stream(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for name in sys.argv[1:]:
        lib = load(name)

        def f():
            out = torch.empty(592, 4096, dtype=torch.float32, device=dev)
            rc = lib.haff_gemm_bf16_cfg(A.data_ptr(), 4096, W.data_ptr(), 4096, out.data_ptr(), 4096, None, None, 0, None,
                                        592, 4096, 4096, 0, 1, 0, 1, stream)
            assert rc == 0
            return out

        ref = f().clone()
        torch.cuda.synchronize()
        exact = float((A.float() @ W.float().t())[0, 0])
        assert float(ref[0, 0]) == exact, (float(ref[0, 0]), exact)
        for beside in (False, True):
            wrong, deltas = 0, {}
            for rep in range(4):
                if beside:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(100):
                            ln(); qkv()
                outs = [f() for _ in range(300)]
                torch.cuda.synchronize()
                for o in outs:
                    if not torch.equal(o, ref):
                        wrong += 1
                        d = (o - ref)
                        for v in torch.unique(d[d != 0]).tolist()[:8]:
                            deltas[v] = deltas.get(v, 0) + 1
            print(f"{name:8s} 128x128 tile, {'beside (layernorm, qkv GEMM)' if beside else 'alone':30s}: wrong {wrong}/1200"
                  f"  output deltas (value: launches) {dict(sorted(deltas.items()))}", flush=True)
    print("delta -64 per fragment = stale K-tile kt-2 (late DMA); +64 = K-tile kt+2 already in the buffer (WAR)")


if __name__ == "__main__":
    main()
