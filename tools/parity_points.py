#!/usr/bin/env python3
"""bf16 HIP path against BOTH oracles — the exact fp32 restatement and its bf16-points mode (oracle.lisa_oracle.bf16_points:
fp32 arithmetic, rounding to bf16 at the kernel boundaries of the HIP path) — on the tiny and mid geometries, plus the ViT-H
width blocks. Also attributes the distance to the exact oracle to the three bf16 stacks: evaluate() with ONE stack's
contribution replaced by the oracle's exact fp32 result is not possible through the product API, so the attribution runs in
the oracle itself: bf16 points switched on for one stack at a time (SAM encoder / CLIP+projector / Llama).
usage: python tools/parity_points.py            (prints a table; the numbers set the test thresholds of tests/test_lisa_gpu.py)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import haff  # noqa
from haff.lisa import LisaMI355
from oracle import lisa_oracle as O
from test_lisa_gpu import _iou, _setup


def cmp(tag, got_l, got_r, ref_l, ref_r):
    rows = []
    for g, r in list(zip(got_l, ref_l)) + list(zip(got_r, ref_r)):
        g = g.float().cpu()
        rows.append(((g - r).abs().max().item() / r.abs().max().item(), _iou(g > 0, r > 0)))
    print(f"  {tag:46s} max rel logit err {max(e for e, _ in rows):.3e}   min IoU {min(i for _, i in rows):.5f}", flush=True)


def main():
    dev = torch.device("cuda:0")
    for cfg_name in ("tiny", "mid"):
        cfg, sd, images, images_clip, ids, forced = _setup(cfg_name, "bf16")
        S = cfg.sam.img_size
        resize = [(S, S), (S, S - 32)]
        orig = [(S, S), (S // 2 + 3, S // 2 - 10)]
        kw = dict(max_new_tokens=forced.shape[1], forced_answer=forced, use_cache=True)
        with torch.no_grad():
            _, el, er, _ = O.lisa_evaluate(sd, cfg, images_clip, images, ids, resize, orig, **kw)
            with O.bf16_points():
                _, pl, pr, _ = O.lisa_evaluate(sd, cfg, images_clip, images, ids, resize, orig, **kw)
        model = LisaMI355(cfg, sd, dtype=torch.bfloat16, device=dev)
        _, left, right, _ = model.evaluate(images_clip.to(dev), images.to(dev), ids.to(dev), resize, orig,
                                           max_new_tokens=forced.shape[1], forced_answer=forced)
        print(cfg_name)
        cmp("HIP bf16 vs exact fp32 oracle", left, right, el, er)
        cmp("HIP bf16 vs bf16-points oracle", left, right, pl, pr)
        cmp("bf16-points oracle vs exact oracle", pl, pr, el, er)
        # attribution: bf16 points in ONE stack of the oracle at a time
        for stack in ("sam", "clip", "llama"):
            real = {n: getattr(O, n) for n in ("sam_image_encoder", "encode_images", "llama_forward")}

            def wrap(fn):
                def f(*a, **k):
                    with O.bf16_points():
                        return fn(*a, **k)
                return f
            target = {"sam": "sam_image_encoder", "clip": "encode_images", "llama": "llama_forward"}[stack]
            setattr(O, target, wrap(real[target]))
            try:
                with torch.no_grad():
                    _, sl, sr, _ = O.lisa_evaluate(sd, cfg, images_clip, images, ids, resize, orig, **kw)
            finally:
                setattr(O, target, real[target])
            cmp(f"oracle, bf16 points in the {stack} stack only vs exact", sl, sr, el, er)


if __name__ == "__main__":
    main()
