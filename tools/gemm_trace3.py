#!/usr/bin/env python3
"""Round 5: where a MIDDLE tile of the persistent 8-wave GEMM spends its time around the tile boundary (lib/libhaff_gemm_trace3.so:
gemm_bf16.hip built with -DHAFF_TUNING -DHAFF_GEMM_TRACE3; add -DHAFF_EXP_NOSTORE for the store-free arm). Per workgroup, waves 0
(group 0) and 4 (group 1) stamp the 100 MHz clock at: 0 tile start, 1 tile-start requests issued, 2..7 the last K-tile (top, load
slot A closed, multiply slot A closed, before / after the wait for the NEXT tile's first K-tile, load slot B closed), 9 K loop end,
10 before pass 0, 11..18 after each epilogue pass, 21..23 / 25..27 inside passes 0 / 4 (after the norm fold, after bias +
activation, after swap / pack / stores), 19 / 20 before / after the barrier behind the epilogue. Medians over the workgroups, us,
relative to stamp 9 of wave 0.   usage: TRACELIB=trace3 python tools/gemm_trace3.py [shape ...]"""
import ctypes, os, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(HERE, "2handedafforder_amd", "lib", "libhaff_gemm_%s.so" % os.environ.get("TRACELIB", "trace3")))
vp, cl, ci = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
lib.haff_gemm_bf16_ln.argtypes = [vp, cl, vp, cl, vp, cl, vp, vp, cl, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
lib.haff_gemm_bf16_rowstats.argtypes = [vp, cl, vp, cl, vp, cl, vp, cl, vp, vp, cl, ci, ci, ci, vp, vp]
lib.haff_gemm_bf16_cfg.argtypes = [vp, cl, vp, cl, vp, cl, vp, vp, cl, vp, ci, ci, ci, ci, ci, ci, ci, vp]
lib.haff_gemm_trace3_read.argtypes = [vp, ci]
SHAPES = {"qkv": (131072, 3840, 1280, "ln_map"), "lin1": (131072, 5120, 1280, "ln_gelu"), "proj": (131072, 1280, 1280, "rowstats"),
          "lin2": (131072, 1280, 5120, "rowstats"), "plain": (131072, 3840, 1280, "plain"), "gateup": (18624, 22016, 4096, "swiglu"),
          "down": (18624, 4096, 11008, "resid")}
dev = torch.device("cuda:0")
want = sys.argv[1:] or ["qkv", "lin1", "proj", "lin2"]
for name in want:
    M, N, K, kind = SHAPES[name]
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
    n_out = N // 2 if kind == "swiglu" else N
    out = torch.empty((M + 1, n_out), dtype=torch.bfloat16, device=dev)
    bias = torch.randn((N,), device=dev)
    ms = []
    st = torch.stack([torch.zeros(M, device=dev), torch.ones(M, device=dev)], 1).contiguous()
    cs = torch.randn((N,), device=dev)
    # the windowed q|k|v scatter: rows of one 14 x 14 window stay together (196-row runs), the runs are permuted
    rmap = (torch.arange(M, device=dev).view(-1, 64)[torch.randperm(M // 64, device=dev)].reshape(-1)).to(torch.int32) if kind == "ln_map" else None
    stat = torch.empty((M, max(N // 64, 1), 2), device=dev)
    resid = torch.randn((M, N), device=dev).to(torch.bfloat16) if kind in ("rowstats", "resid") else None
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if kind in ("ln_map", "ln_gelu"):
            rc = lib.haff_gemm_bf16_ln(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), n_out, bias.data_ptr(), None, 0,
                                       rmap.data_ptr() if rmap is not None else None, st.data_ptr(), cs.data_ptr(), M, N, K,
                                       1 if kind == "ln_gelu" else 0, 0, 0, None)
        elif kind == "rowstats":
            rc = lib.haff_gemm_bf16_rowstats(x.data_ptr(), K, None, 0, w.data_ptr(), K, resid.data_ptr(), N, bias.data_ptr(), resid.data_ptr(), N,
                                             M, N, K, stat.data_ptr(), None)
        else:
            rc = lib.haff_gemm_bf16_cfg(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), n_out, bias.data_ptr() if kind == "plain" else None,
                                        resid.data_ptr() if resid is not None else None, N, None, M, N, K, 0, 0, 1 if kind == "swiglu" else 0, 2, None)
        e1.record()
        assert rc == 0, rc
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    buf = np.zeros(256 * 2 * 32, dtype=np.uint64)
    assert lib.haff_gemm_trace3_read(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(256, 2, 32).astype(np.int64)
    d = (t - t[:, 0:1, 9:10]) / 100.0
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"## {name} {M}x{N}x{K} ({kind}): {min(ms) * 1e3:.1f} us per launch = {2.0 * M * N * K / min(ms) / 1e9:.0f} TFLOP/s, {tiles / 256:.2f} rounds, "
          f"{min(ms) * 1e3 / (tiles / 256):.2f} us per round", flush=True)
    for wv in (0, 1):
        m = np.median(d[:, wv, :], axis=0)
        print(f"  wave {4 * wv}: tile start {m[0]:7.2f} (requests out {m[1]:7.2f}) | last K-tile: top {m[2]:6.2f} slotA-load {m[3]:6.2f} slotA-mult {m[4]:6.2f} "
              f"before-wait {m[5]:6.2f} after-wait {m[6]:6.2f} slotB-load {m[7]:6.2f} | K loop end {m[9]:5.2f} | bias {m[28]:5.2f} ln {m[29]:5.2f} rowmap {m[30]:5.2f} pre {m[10]:5.2f} | passes "
              + " ".join(f"{v:5.2f}" for v in m[11:19]) + f" | pass0: fold {m[21]:5.2f} act {m[22]:5.2f} stores {m[23]:5.2f} | pass4: fold {m[25]:5.2f} act {m[26]:5.2f} "
              f"stores {m[27]:5.2f} | barrier {m[19]:5.2f} -> {m[20]:5.2f}", flush=True)
