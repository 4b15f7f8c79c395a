"""Time haff_relpos_tables_bf16 + the generic global attention on the SAM global-block geometry (32 frames).
usage: python tools/relpos_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa: F401
from haff import ops
dev = torch.device("cuda:0")
B = 32
qkv = (torch.randn(B * 4096, 3840, device=dev) * 0.5).to(torch.bfloat16)
gh, gw = torch.randn(127, 80, device=dev) * 0.1, torch.randn(127, 80, device=dev) * 0.1
v = qkv.view(B, 4096, 3, 16, 80).permute(2, 0, 3, 1, 4)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rh, rw = ops.relpos_tables(v[0], gh, gw, 64)
print(f"relpos tables ({B} frames): {t(lambda: ops.relpos_tables(v[0], gh, gw, 64)):8.1f} us   ({2 * rh.numel() * 4 / 1e9:.2f} GB written)")
print(f"global attention (tables): {t(lambda: ops.attention(v[0], v[1], v[2], 80 ** -0.5, relh=rh, relw=rw, S=64)):8.1f} us")
