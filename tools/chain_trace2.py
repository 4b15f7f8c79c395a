#!/usr/bin/env python3
"""Per-stage timeline of the chained decode step inside tools/decode_step_bench.py's flow (full 7B model, hipGraph replay), for several
batch sizes in ONE process (the first configuration of a process runs ~25 % slower than the later ones: which stage?).
Needs a -DCH_TRACE build selected with HAFF_LIB_PATH.   usage: chain_trace2.py 4,1,2"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import checkpoint, config as hcfg, lib as hlib
from haff.lisa import LisaMI355

dev = torch.device("cuda:0")
cfg = hcfg.haff_7b()
model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), dtype=torch.bfloat16, device=dev)
rd = ctypes.CDLL(hlib.LIB_PATH).haff_decode_chain_trace_read
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["qkv", "attn", "o_proj", "gate|up", "down"]
T0 = 291
for B in [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "4,1,2").split(",")]:
    cache = model._persistent_cache(B, T0 + 8)
    tok = torch.zeros((B,), dtype=torch.long, device=dev)

    def one_step():
        cache["pos"].fill_(T0); cache["nk"].fill_(T0 + 1)
        return model._decode_step(tok, cache)
    for _ in range(3):
        one_step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        one_step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t1) / 10
    rd(None, 1)
    one_step()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (48 * 5 * 4))()
    rd(buf, 0)
    L = cfg.llm.layers
    t = [[buf[(i * 4) + k] for k in range(4)] for i in range(L * 5)]
    t0 = min(r[0] for r in t)
    span = (max(r[3] for r in t) - t0) / 100
    # mean duration of each stage over layers 4..27 (end of the previous stage -> end of this one) and the wait-release latency
    dur, rel = [0.0] * 5, [0.0] * 5
    n = 0
    for l in range(4, 28):
        for s_ in range(5):
            i = l * 5 + s_
            dur[s_] += (t[i][3] - t[i - 1][3]) / 100
            rel[s_] += (t[i][1] - t[i - 1][3]) / 100
        n += 1
    print(f"B={B}: {ms:.3f} ms per step; chain span {span:.1f} us = {span / L:.1f} per layer; per stage (us, mean of layers 4..27): " +
          ", ".join(f"{names[s_]} {dur[s_] / n:.1f} (released {rel[s_] / n:+.1f} after the previous stage's end)" for s_ in range(5)), flush=True)
