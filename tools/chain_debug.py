#!/usr/bin/env python3
"""Debug aid for csrc/decode_chain.hip: one decode step, chained launch vs the same kernel stage by stage vs the five-launch layer;
prints where the scratch buffers of the LAST layer (qkv, att, g) and the residual stream first differ.
usage: python3 tools/chain_debug.py [B] [layers] [T] [width]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_decode_chain_gpu import _model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 291
width = sys.argv[4] if len(sys.argv) > 4 else "7b"
dev = torch.device("cuda:0")
cfg, llm = _model(width, layers, dev)
x = torch.randn((B, T + 1, cfg.llm.hidden), generator=torch.Generator().manual_seed(2)).to(dev, torch.bfloat16)


def run(mode):
    llm.decode_chain = mode
    cache = llm.new_cache(B, T + 9)
    llm.forward(x[:, :T].clone(), cache)
    cache["pos"].fill_(T)
    cache["nk"].fill_(T + 1)
    h = llm.decode_rows(x[:, T:T + 1].clone(), cache).clone()
    torch.cuda.synchronize()
    ch = cache.get("chain")
    return {"h": h, "qkv": ch["qkv"].clone() if ch else None, "att": ch["att"].clone() if ch else None, "g": ch["g"].clone() if ch else None,
            "k": cache["k"][-1][:, :T + 1].clone(), "ssq_a": cache["ssq"][0].clone(), "ssq_b": cache["ssq"][1].clone()}


ref = run("stages")
lib = run(False)
print("stages vs library: h", (ref["h"].float() - lib["h"].float()).abs().max().item(), "k", (ref["k"].float() - lib["k"].float()).abs().max().item())
for rep in range(4):
    got = run(True)
    line = []
    for key in ("qkv", "att", "g", "h", "k", "ssq_a", "ssq_b"):
        a, b = got[key].float(), ref[key].float()
        if key.startswith("ssq"):
            a, b = a[:, :B], b[:, :B]
        d = (a - b).abs()
        n_bad = int((d > 0).sum().item())
        first = int(torch.nonzero(d.flatten() > 0)[0].item()) if n_bad else -1
        line.append(f"{key}: {n_bad} differ (max {d.max().item():.3e}, first flat index {first})")
    print(f"rep {rep}: " + "; ".join(line))
