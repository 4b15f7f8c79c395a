"""Times the kernels of csrc/lora.hip alone at the configs[3] shapes (M = 8 x 351 token rows, H = K = 4096, rank 8):
python tools/lora_bench.py [M]. Prints us per launch and the HBM rate of each kernel's algorithmic bytes."""
import sys
import torch
sys.path.insert(0, ".")
import haff  # noqa: F401,E402
from haff import ops  # noqa: E402
from haff.lib import load_library, check  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2808
H = K = 4096
T, heads, d = 351, 32, 128
dev = torch.device("cuda:0")
lib = load_library()
bf = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(bf).to(dev)
qkv, x, a2 = rnd(M, 3 * H), rnd(M, K), rnd(16, K) * 0.02
b2 = rnd(2, H, 8) * 0.1
Mp = (M + 15) // 16 * 16
tT = torch.zeros((16, Mp), dtype=bf, device=dev)
inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
ang = torch.arange(T, dtype=torch.float32)[:, None] * inv[None, :]
cs = torch.cat([ang.cos(), ang.sin()], 1).contiguous().to(dev)
q, k, v = (torch.empty((M, H), dtype=bf, device=dev) for _ in range(3))
dqkv = torch.empty((M, 3 * H), dtype=bf, device=dev)
dx = rnd(M, K)
keep = (torch.rand((M, K), generator=g) > 0.05).to(bf).to(dev)
s = torch.cuda.current_stream().cuda_stream
n_ws = lib.haff_lora_tn_workspace_elems(M, 16, K)
ws = torch.empty((n_ws,), dtype=torch.float32, device=dev)
o8, o16 = torch.empty((H, 8), dtype=bf, device=dev), torch.empty((16, K), dtype=bf, device=dev)


def t_down():
    ops.linear(a2, x, out=tT[:, :M])


def fwd():
    check(lib.haff_lora_qkv_rope_fwd(qkv.data_ptr(), 3 * H, tT.data_ptr(), Mp, b2[0].data_ptr(), b2[1].data_ptr(), 8, cs.data_ptr(),
                                     q.data_ptr(), k.data_ptr(), v.data_ptr(), H, M, H, d, T, 2.0, s), "fwd")


def bwd():
    check(lib.haff_lora_qkv_rope_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), H, cs.data_ptr(), dqkv.data_ptr(), 3 * H, M, H, d, T, s), "bwd")


def dxk():
    check(lib.haff_lora_dx(tT.data_ptr(), Mp, a2.data_ptr(), K, keep.data_ptr(), K, dx.data_ptr(), K, 1, M, K, 2.0, s), "dx")


def tn8():
    check(lib.haff_lora_tn(tT.data_ptr(), Mp, 8, dqkv.data_ptr(), 3 * H, M, H, ws.data_ptr(), ws.numel(), o8.data_ptr(), 8, 0, 1, 8, 2.0, s), "tn8")


def tn16():
    check(lib.haff_lora_tn(tT.data_ptr(), Mp, 16, x.data_ptr(), K, M, K, ws.data_ptr(), ws.numel(), o16.data_ptr(), K, 0, 0, 16, 2.0, s), "tn16")


MB = 1e-6
cases = [("t^T = A2.x^T (weight-streaming product)", t_down, M * K * 2), ("lora_qkv_rope_fwd", fwd, M * H * 12),
         ("lora_qkv_rope_bwd", bwd, M * H * 12), ("lora_dx (keep, accumulate)", dxk, M * K * 6),
         ("lora_tn<8> + reduce", tn8, M * H * 2), ("lora_tn<16> + reduce", tn16, M * K * 2)]
for name, fn, nbytes in cases:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name:45s} {us:8.1f} us   {nbytes * MB:7.1f} MB   {nbytes / us * 1e-6:6.2f} TB/s")
