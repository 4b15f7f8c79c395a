#!/usr/bin/env python3
"""The chained decode launch alone (ops.decode_chain on synthetic 7B-sized operands, nothing else on the stream): ms per launch by HIP
events, eager and inside a hipGraph, for several allocation orders of its small operands.   usage: chain_alone.py [B] [variant]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variant = sys.argv[2] if len(sys.argv) > 2 else "small_first"
dev = torch.device("cuda:0")
H, F, nh, Lyr, tmax = 4096, 11008, 32, 32, 299
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)


def small():
    d = {"x": torch.randn((B, H), device=dev, generator=g).to(bf), "qkv": torch.zeros((B, 3 * H), dtype=bf, device=dev),
         "att": torch.zeros((B, H), dtype=bf, device=dev), "g": torch.zeros((B, F), dtype=bf, device=dev),
         "ssq_a": torch.zeros((H // 16, 16), device=dev), "ssq_b": torch.zeros((H // 16, 16), device=dev),
         "ws": torch.zeros((H // 16, 2, 16, 16), device=dev), "stats": torch.ones((B, 2), device=dev),
         "sync": torch.zeros((ops.decode_chain_sync_words(Lyr, H),), dtype=torch.int32, device=dev),
         "nk": torch.full((B,), 292, dtype=torch.int32, device=dev)}
    ang = torch.arange(tmax, dtype=torch.float32)[:, None] * (1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128)))[None, :]
    d["cs"] = torch.cat([ang.cos(), ang.sin()], 1).contiguous().to(dev)
    return d


def big():
    w = []
    for _ in range(Lyr):
        w.append(tuple((torch.randn(s, device=dev, generator=g) * 0.02).to(bf) for s in ((3 * H, H), (H, H), (2 * F, H), (H, F))) +
                 (torch.randn((B, tmax, H), device=dev, generator=g).to(bf), torch.randn((B, tmax, H), device=dev, generator=g).to(bf)))
    return w


if variant == "small_first":
    s, w = small(), big()
else:
    w, s = big(), small()
table = ops.decode_chain_table(w)
x0 = s["x"].clone()


def launch():
    ops.decode_chain(table, Lyr, s["x"], s["qkv"], s["att"], s["g"], s["ssq_a"], s["ssq_b"], s["ws"], s["stats"], 1e-5, s["cs"], s["nk"], nh, tmax,
                     128 ** -0.5, s["sync"])


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"B={B} {variant}: eager {timed(launch):.3f} ms per launch", end="", flush=True)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, capture_error_mode="thread_local"):
    launch()
print(f"; hipGraph {timed(gr.replay):.3f} ms; status ok: {ops.decode_chain_status(s['sync'], Lyr)}; x finite: {bool(torch.isfinite(s['x'].float()).all())}", flush=True)
print("   ptrs: " + " ".join(f"{k}={v.data_ptr():#x}" for k, v in s.items()) + f" w0={w[0][0].data_ptr():#x} k0={w[0][4].data_ptr():#x}")


# data dependence: the same launch on other operand VALUES
def launch():
    ops.decode_chain(table, Lyr, s["x"], s["qkv"], s["att"], s["g"], s["ssq_a"], s["ssq_b"], s["ws"], s["stats"], 1e-5, s["cs"], s["nk"], nh, tmax,
                     128 ** -0.5, s["sync"])
for label in ("x = 0", "x = nan", "kv = 0", "kv = nan, x = randn", "weights o/down = 0"):
    if label == "x = 0":
        s["x"].zero_()
    elif label == "x = nan":
        s["x"].fill_(float("nan"))
    elif label == "kv = 0":
        s["x"].copy_(x0)
        for t in w:
            t[4].zero_(); t[5].zero_()
    elif label == "kv = nan, x = randn":
        for t in w:
            t[4].fill_(float("nan")); t[5].fill_(float("nan"))
        s["x"].copy_(x0)
    else:
        for t in w:
            t[4].normal_(); t[5].normal_(); t[1].zero_(); t[3].zero_()
        s["x"].copy_(x0)
    print(f"   {label}: {timed(launch):.3f} ms; x finite after: {bool(torch.isfinite(s['x'].float()).all())}", flush=True)
