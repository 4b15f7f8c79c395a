#!/usr/bin/env python3
"""Batch-1 evaluate() loop (BASELINE.json configs[1]) for profiling: `rocprofv3 --kernel-trace --stats -- python3 tools/b1_run.py`."""
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
model.overlap_streams = "--single-stream" not in sys.argv
frames, clip, ids, forced = make_inputs(cfg, 1, 32, 8, dev)
S = cfg.sam.img_size
n = int(os.environ.get("N", "10"))
for _ in range(3):
    model.evaluate(None, None, ids, [(S, S)], [(S, S)], max_new_tokens=8, forced_answer=forced, frames_u8=frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    model.evaluate(None, None, ids, [(S, S)], [(S, S)], max_new_tokens=8, forced_answer=forced, frames_u8=frames)
torch.cuda.synchronize()
print(f"batch-1 evaluate: {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
