#!/usr/bin/env python3
"""Repeatability of evaluate() at a few frames with the chained decode launch: full 7B geometry, every combination of
{two streams, one stream} x {hipGraph, eager} x {chain, same kernel stage by stage, five launches}; every run is compared bit for bit
with the first of its combination, and the chained runs with the stage-by-stage ones.   usage: chain_stress.py [B] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from bench import make_inputs
from haff import checkpoint, config as hcfg, ops
from haff.lisa import LisaMI355

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
cfg = hcfg.haff_7b()
model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), dtype=torch.bfloat16, device=dev, sam_chunk=2)
S = cfg.sam.img_size
frames, clip, ids, forced = make_inputs(cfg, B, 32, 8, dev)
sizes = [(S, S)] * B


def run():
    with torch.no_grad():
        o, l, r, t = model.evaluate(clip, None, ids, sizes, sizes, max_new_tokens=8, forced_answer=forced, frames_u8=frames)
    torch.cuda.synchronize()
    return [o.clone()] + [m.clone() for m in l] + [m.clone() for m in r] + [x.clone() for x in t]


def same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


ref = {}
for streams in (True, False):
    for graphs in (True, False):
        for chain in (True, "stages", False):
            model.overlap_streams, model.decode_graphs = streams, graphs
            model.llm.decode_chain = chain
            model.decode_chain = "auto" if chain else False
            outs = [run() for _ in range(reps)]
            n_bad = sum(not same(o, outs[0]) for o in outs[1:])
            st = []
            for c in model._caches.values():
                if "chain" in c:
                    st.append(ops.decode_chain_status(c["chain"]["sync"], cfg.llm.layers))
            key = (streams, graphs)
            note = ""
            if chain == "stages":
                ref[key] = outs[0]
            print(f"streams={'2' if streams else '1'} graphs={graphs!s:5} chain={chain!s:6}: {n_bad} of {reps - 1} repeats differ from the first; "
                  f"chain used: {model.last_decode_chain}; status ok: {st}", flush=True)
            if chain is True:
                first_chain = outs[0]
            if chain == "stages":
                print(f"      chained run == stage-by-stage run: {same(first_chain, outs[0])}", flush=True)


# where do repeats of generate() first differ?
for graphs in (True, False):
    model.overlap_streams, model.decode_graphs = False, graphs
    for chain in (True, "stages", False):
        model.llm.decode_chain = chain
        hs = []
        for rep in range(5):
            with torch.no_grad():
                o, h = model.generate(clip, ids, 8, forced)
            torch.cuda.synchronize()
            hs.append(h.clone())
        T = hs[0].shape[1]
        eq = [[int(torch.equal(hs[i], hs[j])) for j in range(5)] for i in range(5)]
        d = (hs[1].float() - hs[0].float()).abs().amax(dim=(0, 2))
        bad = torch.nonzero(d > 0).flatten().tolist()
        print(f"generate() alone, graphs={graphs}, chain={chain}: pairwise equality of 5 runs {eq}; run 1 vs 0 differs at positions {bad[:10]} (max {d.max().item():.3e})", flush=True)
