#!/usr/bin/env python3
"""Stage breakdown of the batch=1 latency (BASELINE.json configs[1]) — host-bound vs device-bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import checkpoint, config as hcfg
from haff.lisa import LisaMI355
from bench import make_inputs


def main():
    dev = torch.device("cuda:0")
    cfg = hcfg.haff_7b()
    model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), device=dev, sam_chunk=1)
    frames, clip, ids, forced = make_inputs(cfg, 1, 32, 8, dev)
    S = cfg.sam.img_size

    def t(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out
    ms_sam, emb = t(lambda: model.get_visual_embs_u8(frames, (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)))
    ms_clip, img = t(lambda: model.encode_images(clip))
    model.decode_graphs = False
    ms_gen_eager, _ = t(lambda: model.generate(clip, ids, 8, forced))
    model.decode_graphs = True
    ms_gen, (out_ids, hidden) = t(lambda: model.generate(clip, ids, 8, forced))
    print(f"generate eager {ms_gen_eager:.1f} ms | with decode graphs {ms_gen:.1f} ms")
    cache = model._persistent_cache(1, 291 + 8)
    cache["pos"].fill_(291); cache["nk"].fill_(291 + 1)
    nxt = torch.zeros((1,), dtype=torch.long, device=dev)

    def one_step(graph):
        model.decode_graphs = graph
        cache["pos"].fill_(291); cache["nk"].fill_(291 + 1)
        return model._decode_step(nxt, cache)
    ms_e, _ = t(lambda: one_step(False), 10)
    ms_g, _ = t(lambda: one_step(True), 10)
    print(f"one decode step: eager {ms_e:.2f} ms | graph replay {ms_g:.2f} ms")
    ms_seg, (pred, fidx, counts) = t(lambda: model.seg_embeddings(out_ids, hidden))
    ms_dec, dec = t(lambda: model.sam_decoder.decode(emb, fidx, pred))
    ms_post, _ = t(lambda: model.sam_decoder.postprocess(dec[0], (S, S), (S, S)))
    ms_all, _ = t(lambda: model.evaluate(clip, None, ids, [(S, S)], [(S, S)], max_new_tokens=8, forced_answer=forced, frames_u8=frames))
    print(f"sam_encoder {ms_sam:.1f} | clip+proj {ms_clip:.1f} | generate(incl clip) {ms_gen:.1f} | seg+fcs {ms_seg:.1f} | "
          f"decoders {ms_dec:.1f} | postprocess(1 side) {ms_post:.1f} | evaluate {ms_all:.1f} ms")


if __name__ == "__main__":
    main()
