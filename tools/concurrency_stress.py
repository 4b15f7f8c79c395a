"""Two-queue regression probe: a victim GEMM (structured data: A holds 1 + (k // 64) % 4, W is all
ones, so every wrong K fragment shows up as an exact multiple of 16 in the fp32 output) is launched 300 times on the
default stream while (layernorm, qkv GEMM) pairs of the SAM block run on a second stream. History: the 128x128 tile
waited for "the older K-tile" of its LDS-DMA double buffer with a COUNTED s_waitcnt vmcnt(8); LDS-DMA operations do not
retire in issue order under memory contention, and 5-10 % of its launches came back with one 32-deep k-step of 8 rows
holding the buffer's previous content (differences of exactly +-64). With vmcnt(0)-only waits every tile is clean.
usage: python tools/concurrency_stress.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import haff
from haff import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def mk(*s, sc=0.05): return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
M = 9800
x = mk(M, 1280, sc=1.0)
wqkv, bqkv = mk(3840, 1280), mk(3840).float()
wproj, bproj = mk(1280, 1280), mk(1280).float()
w1, b1 = mk(5120, 1280), mk(5120).float()
w2, b2 = mk(1280, 5120), mk(1280).float()
th, tw = torch.randn(27, 80, device=dev) * 0.1, torch.randn(27, 80, device=dev) * 0.1
lw, lb = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
st = {}
def ln():   st["h"] = ops.layernorm(st.get("x", x), lw, lb, 1e-6)
def qkv():  st["qkv"] = ops.linear(st.get("h", x), wqkv, bias=bqkv)
def win():
    v = st["qkv"].view(50, 196, 3, 16, 80).permute(2, 0, 3, 1, 4)
    st["a"] = ops.window_attention(v[0], v[1], v[2], 80 ** -0.5, th, tw, 14)
def proj(): st["x"] = ops.linear(st["a"].view(M, 1280), wproj, bias=bproj, resid=x)
def fc1():  st["f"] = ops.linear(st.get("h", x), w1, bias=b1, act=ops.ACT_GELU)
def fc2():  st["x"] = ops.linear(st["f"], w2, bias=b2, resid=x)
for f in (ln, qkv, win, proj, fc1, fc2): f()
torch.cuda.synchronize()
seqs = {"ln,qkv": (ln, qkv), "qkv,win": (qkv, win), "win,proj": (win, proj), "ln,fc1": (ln, fc1), "fc1,fc2": (fc1, fc2),
        "qkv,proj": (qkv, proj), "ln,win": (ln, win), "block": (ln, qkv, win, proj, ln, fc1, fc2),
        "block w/o win": (ln, qkv, proj, ln, fc1, fc2), "block w/o ln": (qkv, win, proj, fc1, fc2)}
k = torch.arange(4096, device=dev)
A = (1 + (k // 64) % 4).to(torch.bfloat16)[None, :].expand(592, 4096).contiguous()
W = torch.ones(4096, 4096, dtype=torch.bfloat16, device=dev)
side = torch.cuda.Stream(dev)
seq = (ln, qkv)
for tile, tname in ((2, "256x256, 8 waves"), (3, "256x256, 4 waves"), (1, "128x128, 4 waves")):
    f = lambda: ops.linear(A, W, tile_cfg=tile, out_dtype=torch.float32)
    ref = f().clone(); torch.cuda.synchronize()
    tot = 0
    for rep in range(4):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(100):
                for g in seq: g()
        outs = [f() for _ in range(300)]
        torch.cuda.synchronize()
        tot += sum(int(not torch.equal(o, ref)) for o in outs)
    print(f"victim tile {tname}: wrong {tot}/1200 beside (layernorm, qkv GEMM) on a second stream", flush=True)
