#!/usr/bin/env python3
"""Per-stage timeline of the chained decode step (a -DCH_TRACE build of csrc/decode_chain.hip, tools/build_chain_variant.sh trace -DCH_TRACE -DCH_PLACE;
run with HAFF_LIB_PATH=.../libhaff_chain_trace.so): for every (layer, stage) the first workgroup's start, the first and the last
satisfied wait and the last workgroup's end, in us from the launch's first stamp.   usage: chain_trace.py [B] [layers]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_decode_chain_gpu import _model  # noqa: E402
from haff import lib as hlib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = 291
dev = torch.device("cuda:0")
cfg, llm = _model("7b", layers, dev)
x = torch.randn((B, T + 1, cfg.llm.hidden), generator=torch.Generator().manual_seed(2)).to(dev, torch.bfloat16)
L = hlib.load_library()
rd = ctypes.CDLL(hlib.LIB_PATH).haff_decode_chain_trace_read
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
llm.decode_chain = True
cache = llm.new_cache(B, T + 9)
llm.forward(x[:, :T].clone(), cache)
cache["pos"].fill_(T)
cache["nk"].fill_(T + 1)
for _ in range(3):
    llm.decode_rows(x[:, T:T + 1].clone(), cache)
torch.cuda.synchronize()
rd(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
llm.decode_rows(x[:, T:T + 1].clone(), cache)
e1.record()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (48 * 5 * 4))()
rd(buf, 0)
t = [[buf[(i * 4) + k] for k in range(4)] for i in range(layers * 5)]
t0 = min(r[0] for r in t)
names = ["qkv", "attn", "o_proj", "gate|up", "down"]
print(f"B={B}, {layers} layers: decode_rows {e0.elapsed_time(e1) * 1e3:.1f} us by events; chain span {(max(r[3] for r in t) - t0) / 100:.1f} us")
print("%-5s %-8s %10s %12s %12s %10s %9s" % ("layer", "stage", "1st start", "1st wait ok", "last wait ok", "last end", "stage us"))
prev_end = 0.0
for i, r in enumerate(t):
    st, w0, w1, en = [(v - t0) / 100.0 if 0 < v < (1 << 63) else float("nan") for v in r]
    print("%-5d %-8s %10.2f %12.2f %12.2f %10.2f %9.2f" % (i // 5, names[i % 5], st, w0, w1, en, en - prev_end))
    prev_end = en

pr = ctypes.CDLL(hlib.LIB_PATH).haff_decode_chain_place_read
pr.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
pb = (ctypes.c_uint * (5 * 1024))()
pr(pb, None)
nbs = [3 * cfg.llm.hidden // 16, B * cfg.llm.heads, cfg.llm.hidden // 8, 2 * cfg.llm.ffn // 32, cfg.llm.hidden // 8]
from collections import Counter
print("placement of layer 2's workgroups (CU key = xcc, se, sh, cu of HW_ID):")
for st in range(5):
    keys = [((pb[st * 1024 + r] >> 16) & 15, (pb[st * 1024 + r] >> 13) & 7, (pb[st * 1024 + r] >> 12) & 1, (pb[st * 1024 + r] >> 8) & 15) for r in range(min(nbs[st], 1024))]
    c = Counter(keys)
    hist = Counter(c.values())
    xc = Counter(k[0] for k in keys)
    print("  %-8s %4d workgroups on %3d distinct CUs; workgroups per CU -> CUs: %s; per XCD: %s" % (names[st], len(keys), len(c), dict(sorted(hist.items())), [xc[i] for i in range(8)]))
