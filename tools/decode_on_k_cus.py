#!/usr/bin/env python3
"""One KV-cached decode step (the hipGraph generate() replays) on k of the 256 CUs: the other 256 - k are held by sleeping
workgroups that claim a CU's LDS each (tools/probes/cu_blocker.hip, built here with hipcc). Says what the decode kernels
deliver per CU with nothing else running — the number the two-stream plan (overlap.py) rests on.
Usage: [CONFIG=7b|13b] decode_on_k_cus.py [batch]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import haff  # noqa
from haff import checkpoint, config as hcfg, overlap
from haff.lisa import LisaMI355
from bench import make_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
so = "/tmp/cu_blocker.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools/probes/cu_blocker.hip")])
blk = ctypes.CDLL(so)
blk.cu_blocker_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = {"7b": hcfg.haff_7b, "13b": hcfg.haff_13b}[os.environ.get("CONFIG", "7b")]()
model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), device=dev, sam_chunk="auto")
frames, clip, ids, forced = make_inputs(cfg, B, 32, 8, dev)
S = cfg.sam.img_size
model.evaluate(None, None, ids, [(S, S)] * B, [(S, S)] * B, max_new_tokens=8, forced_answer=forced, frames_u8=frames)
torch.cuda.synchronize()
T0 = ids.shape[1] + 255
cache = model._persistent_cache(B, T0 + 8)
st = cache["book"]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream(device=dev)
nbytes = overlap.decode_step_bytes(cfg, B, T0 + 4)


def reset():
    st["steps"].fill_(2)
    st["finished"].zero_()
    st["t_rows"].fill_(T0 + 2)
    cache["pos"].fill_(T0 + 2)
    cache["nk"].fill_(T0 + 3)


print("%s, batch %d: %.2f GB per decode step" % (cfg.name, B, nbytes / 1e9))
for k in (256, 224, 192, 160, 128, 96, 64, 32):
    n_block = 256 - k
    reset()
    torch.cuda.synchronize()
    if n_block:
        rc = blk.cu_blocker_launch(n_block, 158 * 1024, 100_000_00, sink.data_ptr(), side.cuda_stream)   # 100 ms
        assert rc == 0, rc
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 4
    e0.record()
    for _ in range(n):
        model._decode_book_step(cache)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("  k = %3d CUs: %6.2f ms per step  %5.2f TB/s  %5.1f GB/s per CU" % (k, ms, nbytes / ms / 1e9, nbytes / ms / 1e6 / k), flush=True)
