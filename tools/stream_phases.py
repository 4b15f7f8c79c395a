#!/usr/bin/env python3
"""Two-stream evaluate() at the bench geometry, split by HIP events (no profiler): when, relative to the start of the step, does
each phase of the caller's stream (CLIP, prefill, the decode steps, the decoder tail) and each chunk of the encoder's side stream
begin and end. Usage: [CONFIG=7b|13b] stream_phases.py [frames] [sam_chunk|auto] [caps|auto|off] [wait|nowait|auto]
(caps: workgroups per persistent GEMM launch for each encoder chunk, e.g. 256,256,224,224 — see overlap.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import checkpoint, config as hcfg
from haff.lisa import LisaMI355
from bench import make_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chunk = sys.argv[2] if len(sys.argv) > 2 else "auto"
caps = sys.argv[3] if len(sys.argv) > 3 else "auto"
dev = torch.device("cuda:0")
cfg = {"7b": hcfg.haff_7b, "13b": hcfg.haff_13b}[os.environ.get("CONFIG", "7b")]()
model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), device=dev, sam_chunk=chunk if chunk == "auto" else int(chunk))
model.sam_chunk_caps = {"auto": "auto", "off": None}[caps] if caps in ("auto", "off") else [int(c) for c in caps.split(",")]
if len(sys.argv) > 4:
    model.sam_waits_for_prefill = {"wait": True, "nowait": False, "auto": "auto"}[sys.argv[4]]
frames, clip, ids, forced = make_inputs(cfg, B, 32, 8, dev)
S = cfg.sam.img_size
marks = []


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def inner(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()     # on whichever stream is current: the side stream inside launch_sam
        r = fn(*a, **kw)
        e1.record()
        marks.append((label, e0, e1))
        return r
    setattr(obj, name, inner)


wrap(model, "encode_images", "clip+projector")
wrap(model.llm, "forward", "prefill")
wrap(model, "_decode_book_step", "decode step")
wrap(model, "get_visual_embs_u8", "sam encoder")
wrap(model, "_decoder_tail", "decoder tail")


def run():
    return model.evaluate(None, None, ids, [(S, S)] * B, [(S, S)] * B, max_new_tokens=8, forced_answer=forced, frames_u8=frames)


for _ in range(3):
    run()
torch.cuda.synchronize()
for rep in range(3):
    marks.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    print("step %d: %.1f ms  (%d frames, plan: caps %s, encoder waits for the prefill %s, chunk %s)" % ((rep, e0.elapsed_time(e1), B) + model.last_plan))
    dec = [m for m in marks if m[0] == "decode step"]
    for label, a, b in marks:
        if label == "decode step":
            continue
        print("   %-16s %7.1f .. %7.1f  (%6.1f ms)" % (label, e0.elapsed_time(a), e0.elapsed_time(b), a.elapsed_time(b)))
    if dec:
        print("   %-16s %7.1f .. %7.1f  (%6.1f ms, %d steps: %s)" % (
            "decode", e0.elapsed_time(dec[0][1]), e0.elapsed_time(dec[-1][2]), dec[0][1].elapsed_time(dec[-1][2]), len(dec),
            " ".join("%.1f" % a.elapsed_time(b) for _, a, b in dec)))
