#!/usr/bin/env python3
"""The fine-tune step (bench.py --mode train geometry) split by HIP events on its two streams: when, relative to the start of the
step, do the frozen encoder (side stream), CLIP, the Llama forward, the two mask decoders, the losses, backward and the optimizer
begin and end.   Usage: train_phases.py [samples] [sam_cap]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import config as hcfg, ops, train_ops as T, weights as hw
from haff.train_model import LisaTrainable
from bench import make_train_batch

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = hcfg.haff_7b()
sd = hw.make_state_dict_device(cfg, 1234, dev, torch.bfloat16)
model = LisaTrainable(cfg, sd, dtype=torch.bfloat16, device=dev)
del sd
batch = make_train_batch(cfg, b, 96, (1024, 1024), dev, seed=1234)
named = list(model.named_parameters())
reducer = T.GradBucketReducer(named)
opt = T.BucketAdamW(reducer, named)
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


def capped(fn):
    def inner(*a, **kw):
        old = ops.gemm_stream_cap(cap)
        try:
            return fn(*a, **kw)
        finally:
            ops.gemm_stream_cap(old)
    return inner


if cap != 256:
    model.base.get_visual_embs = capped(model.base.get_visual_embs)
wrap(model.base, "get_visual_embs", "sam encoder (side)")
wrap(model.base, "encode_images", "clip + projector")
wrap(model, "_llm", "llama forward")
wrap(model, "_decoder", "mask decoder fwd")


def mark(label):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((label, e, e))


def step():
    reducer.zero()
    reducer.begin(sync=True)
    out = model(**batch)
    mark("forward enqueued / losses done")
    out["loss"].backward()
    mark("backward done")
    reducer.finish()
    clip = T.clip_coef_device(T.grad_norm(reducer.grads()), 1.0)
    opt.step(lr=3e-4, gscale=1.0, gscale_dev=clip)
    mark("optimizer done")


for _ in range(3):
    step()
torch.cuda.synchronize()
for rep in range(3):
    marks.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    step()
    e1.record()
    torch.cuda.synchronize()
    print("step %d: %.1f ms (%d samples, encoder cap %d)" % (rep, e0.elapsed_time(e1), b, cap))
    for label, a, bb in marks:
        if a is bb:
            print("   %-32s at %7.1f" % (label, e0.elapsed_time(a)))
        else:
            print("   %-32s %7.1f .. %7.1f  (%6.1f ms)" % (label, e0.elapsed_time(a), e0.elapsed_time(bb), a.elapsed_time(bb)))
