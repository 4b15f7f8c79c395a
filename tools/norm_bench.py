import sys, os, torch
sys.path.insert(0, os.getcwd())
import haff
from haff import ops
dev = torch.device("cuda:0")
for R, C in [(65536, 1280), (18624, 4096), (16448, 1024)]:
    x = torch.randn((R, C), device=dev).to(torch.bfloat16)
    w = torch.randn((C,), device=dev); b = torch.randn((C,), device=dev)
    out = torch.empty_like(x)
    for name, fn in [("ln", lambda: ops.layernorm(x, w, b, 1e-6, out=out)), ("rms", lambda: ops.rmsnorm(x, w, 1e-5, out=out))]:
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{name} {R}x{C}: {t:7.1f} us  {R*C*4/t/1e6:5.2f} TB/s")
