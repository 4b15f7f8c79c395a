#!/usr/bin/env python3
"""The two probes overlap.calibrate() reads on this device (an encoder-shaped MFMA product, a weight stream past the Infinity
Cache), five times, and the Rates it derives — what overlap.PROBE_NOMINAL was taken from (profiles/r6_overlap_probe.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa: E402,F401
from haff import overlap  # noqa: E402

dev = torch.device("cuda", 0)
for i in range(5):
    p = overlap.probe(dev)
    print("probe %d: mfma %.1f TFLOP/s, stream %.3f TB/s" % (i, p["mfma_flops"] / 1e12, p["stream_bytes"] / 1e12), flush=True)
print(overlap.calibrate(dev))
