import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import haff
from haff import ops
dev = torch.device("cuda:0")
shapes = [("qkv", 12288, 4096, False), ("o_proj", 4096, 4096, False), ("gate_up", 22016, 4096, True), ("down", 4096, 11008, False), ("lm_head", 32003, 4096, False)]
for name, N, K, sw in shapes:
    copies = max(2, int(1.2e9 // (N * K * 2)))
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(copies)]
    line = f"{name:8s}"
    for M in (int(m) for m in os.environ.get('MS', '33,48,64').split(',')):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        for i in range(copies): ops.linear(x, ws[i], swiglu=sw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 4 * copies
        e0.record()
        for i in range(reps): ops.linear(x, ws[i % copies], swiglu=sw)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        line += f"  M={M}: {us:6.1f} us ({N * K * 2 / us / 1e6:4.2f} TB/s)"
    print(line, flush=True)
