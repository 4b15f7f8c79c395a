#!/usr/bin/env python3
"""Phase timeline of the fused window-attention kernel (workgroup 0 / wave 0), from a -DHAFF_WIN_TRACE build:
hipcc ... -DHAFF_WIN_TRACE -shared window_attention.hip -o lib/libhaff_win_trace.so"""
import ctypes
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(HERE, "2handedafforder_amd", "lib", "libhaff_win_trace.so"))
vp, cl, ci, cf = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
lib.haff_window_attention_bf16.argtypes = [vp, cl, cl, cl, vp, cl, cl, cl, vp, cl, cl, cl, vp, cl, cl, cl, ci, ci, ci, ci, cf, vp, vp, ci, ci, cl, vp]
lib.haff_window_attention_bf16.restype = ci
lib.haff_win_trace_read.argtypes = [vp, ci]

dev = torch.device("cuda:0")
S, d, H, n_win = 14, 80, 16, 400
N = S * S
qkv = torch.randn((n_win * N, 3 * H * d), device=dev).to(torch.bfloat16)
out = torch.empty((n_win, N, H * d), dtype=torch.bfloat16, device=dev)
th = (torch.randn((2 * S - 1, d), device=dev) * 0.3).to(torch.bfloat16)
tw = (torch.randn((2 * S - 1, d), device=dev) * 0.3).to(torch.bfloat16)
row = 3 * H * d
base = qkv.data_ptr()
for _ in range(3):
    rc = lib.haff_window_attention_bf16(base, N * row, d, row, base + H * d * 2, N * row, d, row, base + 2 * H * d * 2, N * row, d, row,
                                        out.data_ptr(), N * H * d, d, H * d, n_win, H, S, d, d ** -0.5, th.data_ptr(), tw.data_ptr(), 0, 0, 0, None)
    assert rc == 0
torch.cuda.synchronize()
buf = np.zeros(64 * 16, dtype=np.uint64)
lib.haff_win_trace_read(buf.ctypes.data, 64 * 16)
t = buf.reshape(64, 16).astype(np.int64)
names = ["issue loads", "rel-pos q0", "scores q0", "softmax q0", "PV q0", "store+relpos q1", "scores q1", "softmax q1", "PV q1", "store q1"]
idx = [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 12]
for it in range(2, 8):
    r = t[it]
    parts = [f"{names[i]} {(r[idx[i + 1]] - r[idx[i]]) / 100.0:5.2f}" for i in range(len(names))]
    print(f"item {it}: total {(t[it + 1][0] - r[0]) / 100.0:6.2f} us | " + " | ".join(parts) +
          f" | lds write {(r[13] - r[12]) / 100.0:5.2f} | barrier {(r[14] - r[13]) / 100.0:5.2f}")
