#!/usr/bin/env python3
"""ONE full-depth frame of the headline geometry (BASELINE.json configs[1]) through the HIP path in both numeric modes and
through the CPU oracle on the SAME weights: bench.parity_full_frame as a stand-alone run, with optional numeric variants of the
bf16 mode (A/B of what each costs in distance to the oracle).
usage: python tools/full_frame_parity.py [--config 7b|13b] [--attribution] [--field gaussian|two_plateau] [--variants a,b,...] [--out file.json]
Round 6: --field two_plateau = the trained-like logit field (tools/parity_bimodal.py) AT FULL SIZE; the default variants are the Pareto of
DESIGN.md section 2 (bf16 / fused fp32 streams / + f32 neck)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from haff import config as hcfg  # noqa: E402


def _fp32_stream(model):
    model.sam_encoder.fp32_stream = True


def _fp32_stream_unfused(model):
    model.sam_encoder.fp32_stream = model.llm.fp32_stream = True
    model.sam_encoder.fused_fp32_stream = False


def _f32neck(model):
    model.sam_encoder.neck_f32 = True


def _fp32_stream_llm(model):
    model.llm.fp32_stream = True


def _fp32_stream_both(model):
    model.sam_encoder.fp32_stream = True
    model.llm.fp32_stream = True


def _no_fold(model):
    model.sam_encoder.fold_norms = False


def _tables_global(model):
    model.sam_encoder.fused_global = False


VARIANTS = {"bf16_fp32_stream_sam": _fp32_stream, "bf16_fp32_stream": bench._streams_fused, "bf16_fp32_stream_f32neck": bench._streams_fused_neck,
            "bf16_fp32_stream_unfused": _fp32_stream_unfused, "bf16_f32neck": _f32neck, "bf16_fp32_stream_llm": _fp32_stream_llm, "bf16_fp32_stream_both": _fp32_stream_both, "bf16_no_fold": _no_fold, "bf16_tables_global": _tables_global}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="7b", choices=["7b", "13b"])
    ap.add_argument("--attribution", action="store_true")
    ap.add_argument("--field", default="gaussian", choices=["gaussian", "two_plateau"])
    ap.add_argument("--modes", default="bf16,fp32")
    ap.add_argument("--variants", default="")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    cfg = {"7b": hcfg.haff_7b, "13b": hcfg.haff_13b}[args.config]()
    dev = torch.device("cuda", 0)
    threads = min(len(os.sched_getaffinity(0)), 32)
    variants = {k: VARIANTS[k] for k in args.variants.split(",") if k} or None
    res = bench.parity_full_frame(cfg, dev, threads, modes=tuple(m for m in args.modes.split(",") if m),
                                  attribution=args.attribution, seed=args.seed, variants=variants, field=args.field)
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
