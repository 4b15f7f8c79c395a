#!/usr/bin/env python3
"""Batch-1 evaluate() split by HIP events (no profiler): wall time of each phase on the single-stream schedule."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import checkpoint, config as hcfg
from haff.lisa import LisaMI355
from bench import make_inputs

dev = torch.device("cuda:0")
cfg = hcfg.haff_7b()
model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), device=dev, sam_chunk=1)
model.overlap_streams = False
frames, clip, ids, forced = make_inputs(cfg, 1, 32, 8, dev)
S = cfg.sam.img_size
marks = []


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def inner(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        marks.append((label, e0, e1))
        return r
    setattr(obj, name, inner)


wrap(model, "encode_images", "clip+projector")
wrap(model.llm, "forward", "prefill")
wrap(model, "generate", "generate (clip + prefill + decode)")
wrap(model, "_decode_book_step", "decode step")
for name in ("get_visual_embs_u8", "get_visual_embs_frames", "get_visual_embs"):
    if hasattr(model, name):
        wrap(model, name, "sam encoder")
wrap(model, "seg_embeddings", "seg gather + text_hidden_fcs")


def run():
    return model.evaluate(None, None, ids, [(S, S)], [(S, S)], max_new_tokens=8, forced_answer=forced, frames_u8=frames)


for _ in range(3):
    run()
torch.cuda.synchronize()
n = 10
tot = {}
t0 = time.perf_counter()
for _ in range(n):
    marks.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    for label, a, b in marks + [("evaluate", e0, e1)]:
        tot.setdefault(label, [0.0, 0])
        tot[label][0] += a.elapsed_time(b)
        tot[label][1] += 1
wall = (time.perf_counter() - t0) / n * 1e3
print(f"wall {wall:.2f} ms per evaluate")
for label, (t, c) in tot.items():
    print(f"{label:40s} {t / n:8.2f} ms per evaluate ({c // n} calls, {t / c * 1e3:8.1f} us each)")
