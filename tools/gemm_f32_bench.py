"""Timing of the f32-input MFMA GEMM (haff_gemm_f32) on the decoder-tail shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import haff  # noqa
from haff import ops

dev = torch.device("cuda:0")
for M, N, K in [(4096, 4096, 4096), (64 * 4096, 128, 256), (64 * 4096, 256, 128), (64 * 4096, 256, 256), (384, 2048, 256), (64 * 4096, 256, 2304)]:
    x = torch.randn((M, K), device=dev)
    w = torch.randn((N, K), device=dev) * K ** -0.5
    for _ in range(3):
        ops.linear(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.linear(x, w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"gemm_f32 {M}x{N}x{K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
