#!/usr/bin/env python3
"""Per-CU streaming rate of the decode-sized product (<= 16 rows against a [N][K] bf16 weight matrix) in several forms, on k of
the 256 CUs (the rest held by tools/probes/cu_blocker.hip; every probe claims LDS so that it is confined too): the library kernel
(ops.linear), a persistent register ring 8 k-steps deep with counted waits (plain and non-temporal loads; the 16- and 24-deep instances
of the probe give the same times but hipcc builds them with wrong sums once the kernel also claims LDS: not run), persistent LDS-DMA
rings 8 / 16 / 32 KB per wave deep under three cache policies (tools/probes/stream_probe.hip).
Round 5 result (profiles/r5_stream_probe_per_cu.txt): a CU takes in ~21 GB/s through register loads and ~31 GB/s through LDS-DMA
whatever the depth, the waves per CU or the policy — the decode kernels already sit at the first number.
Usage: stream_probe.py [rows] [N] [K]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import haff  # noqa
from haff import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12288
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096


def build(name):
    so = "/tmp/%s.so" % name
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools/probes/%s.hip" % name)])
    return ctypes.CDLL(so)


blk, prb = build("cu_blocker"), build("stream_probe")
blk.cu_blocker_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
prb.stream_probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
g = torch.Generator(device=dev).manual_seed(3)
n_copies = max(2, int(700e6 // (N * K * 2)) + 1)      # a repeat never finds its weights in the 256 MB MALL
ws = [(torch.randn((N, K), device=dev, generator=g) * K ** -0.5).bfloat16() for _ in range(n_copies)]
x = torch.randn((M, K), device=dev, generator=g).bfloat16()
out_lib = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
part = torch.zeros((4, 16, N), device=dev, dtype=torch.float32)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
names = {-1: "library kernel", 0: "register ring 8", 1: "register ring 16", 2: "register ring 24", 3: "LDS-DMA ring 8 KB",
         4: "LDS-DMA ring 16 KB", 5: "LDS-DMA ring 32 KB", 6: "reg ring 8, nt", 7: "reg ring 16, nt", 8: "LDS-DMA 16 KB aux 1", 9: "LDS-DMA 16 KB aux 2", 10: "LDS-DMA 16 KB aux 3"}

ref = x.float() @ ws[0].float().T
LIB_ONLY = bool(os.environ.get("LIB_ONLY"))
for v in (() if LIB_ONLY else (0, 3, 4, 5, 6, 8, 9, 10)):
    part.zero_()
    rc = prb.stream_probe_launch(v, ws[0].data_ptr(), x.data_ptr(), part.data_ptr(), N, K, M, 97, main.cuda_stream)
    assert rc == 0, (v, rc)
    torch.cuda.synchronize()
    got = part.sum(0)[:M]
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print("%-20s max rel err %.2e" % (names[v], err))
    assert err < 2e-3, "probe variant %d is wrong" % v

print("rows %d, weights %d x %d (%.0f MB)" % (M, N, K, N * K * 2 / 1e6))
for k in (256, 64, 32):
    for v in ((-1,) if LIB_ONLY else (-1, 0, 6, 3, 4, 5, 8, 9, 10)):
        for per_cu in ((0,) if v < 0 else (1, 2, 3)):
            if v in (4, 8, 9, 10) and per_cu == 3:
                continue    # 64 KB of LDS per workgroup: two per CU at most
            if v == 5 and per_cu >= 2:
                continue
            torch.cuda.synchronize()
            if k < 256:
                rc = blk.cu_blocker_launch(256 - k, 158 * 1024, 3_000_000, sink.data_ptr(), side.cuda_stream)   # 30 ms
                assert rc == 0
            n = 8
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                w = ws[i % n_copies]
                if v < 0:
                    ops.linear(x, w, out=out_lib)
                else:
                    prb.stream_probe_launch(v, w.data_ptr(), x.data_ptr(), part.data_ptr(), N, K, M, k * per_cu, main.cuda_stream)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            gb = N * K * 2 / us / 1e3
            print("  k = %3d  %-20s %s  %7.1f us  %5.0f GB/s  %5.1f per CU" % (k, names[v], ("%d WG/CU" % per_cu) if v >= 0 else "       ", us, gb, gb / k), flush=True)
