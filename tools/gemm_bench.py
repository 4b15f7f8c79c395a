#!/usr/bin/env python3
"""Per-shape throughput of haff_gemm_bf16 on the shapes the 2Haff path launches (A/B across tile configs,
interleaved rounds in one process — cdna guide rule 24). Random operands (rule 25)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import ops

SHAPES = [  # name, M, N, K, kind
    ("sam qkv win   B16", 16 * 4900, 3840, 1280, "bias"),
    ("sam proj win  B16", 16 * 4900, 1280, 1280, "resid"),
    ("sam lin1      B16", 16 * 4096, 5120, 1280, "gelu"),
    ("sam lin2      B16", 16 * 4096, 1280, 5120, "resid"),
    ("sam qkv win   B8", 8 * 4900, 3840, 1280, "bias"),
    ("sam lin2      B8", 8 * 4096, 1280, 5120, "resid"),
    ("llama qkv     B64", 64 * 291, 12288, 4096, "none"),
    ("llama o       B64", 64 * 291, 4096, 4096, "resid"),
    ("llama gate/up B64", 64 * 291, 22016, 4096, "swiglu"),
    ("llama down    B64", 64 * 291, 4096, 11008, "resid"),
    ("llama qkv dec B1", 1, 12288, 4096, "none"),
    ("llama o   dec B1", 1, 4096, 4096, "resid"),
    ("llama gu  dec B8", 8, 22016, 4096, "swiglu"),
    ("llama dwn dec B8", 8, 4096, 11008, "resid"),
    ("13b qkv   dec B8", 8, 15360, 5120, "none"),
    ("llama qkv dec B64", 64, 12288, 4096, "none"),
    ("llama gu dec  B64", 64, 22016, 4096, "swiglu"),
    ("clip qkv      B64", 64 * 257, 3072, 1024, "bias"),
    ("clip fc1      B64", 64 * 257, 4096, 1024, "qgelu"),
    ("sam neck 3x3  B16", 16 * 4096, 256, 2304, "none"),
    ("dec kproj     P64", 64 * 4096, 128, 256, "bias"),
    ("square 8192", 8192, 8192, 8192, "none"),
]


TRAIN_SHAPES = [  # the fine-tune step's Llama products (8 samples x 351 ids = 2808 rows): SHAPESET=train
    ("ft o / dX     2808", 2808, 4096, 4096, "resid"),
    ("ft qkv        2808", 2808, 12288, 4096, "none"),
    ("ft gate/up    2808", 2808, 22016, 4096, "swiglu"),
    ("ft down       2808", 2808, 4096, 11008, "resid"),
    ("ft dX gate/up 2808", 2808, 4096, 22016, "none"),
    ("ft dX down    2808", 2808, 11008, 4096, "none"),
    ("ft lm_head    2808", 2808, 32003, 4096, "none"),
    ("rows 1500", 1500, 4096, 4096, "none"),
    ("rows 5000", 5000, 4096, 4096, "none"),
]


def main():
    dev = torch.device("cuda:0")
    global SHAPES
    if os.environ.get("SHAPESET") == "train":
        SHAPES = TRAIN_SHAPES
    rounds = int(os.environ.get("ROUNDS", "5"))
    cfgs = [int(c) for c in os.environ.get("CFGS", "1,2,0").split(",")]
    print(f"{'shape':22s} {'M':>7s} {'N':>6s} {'K':>6s} | " + " | ".join(f"cfg{c}: us   TF/s" for c in cfgs))
    for name, M, N, K, kind in SHAPES:
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
        n_out = N // 2 if kind == "swiglu" else N
        out = torch.empty((M, n_out), dtype=torch.bfloat16, device=dev)
        bias = torch.randn((N,), device=dev) if kind in ("bias", "gelu", "qgelu") else None
        resid = torch.randn((M, n_out), device=dev).to(torch.bfloat16) if kind == "resid" else None
        act = {"gelu": 1, "qgelu": 2}.get(kind, 0)
        times = {c: [] for c in cfgs}
        for r in range(rounds + 1):
            for c in cfgs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ops.linear(x, w, bias=bias, act=act, resid=resid, out=out, swiglu=(kind == "swiglu"), tile_cfg=c)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[c].append(e0.elapsed_time(e1) / 3 * 1e3)
        fl = 2.0 * M * N * K
        cols = []
        for c in cfgs:
            t = sorted(times[c])[len(times[c]) // 2]
            cols.append(f"{t:9.1f} {fl / t / 1e6:6.0f}")
        print(f"{name:22s} {M:7d} {N:6d} {K:6d} | " + " | ".join(cols), flush=True)


if __name__ == "__main__":
    main()
