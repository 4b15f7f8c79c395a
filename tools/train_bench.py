#!/usr/bin/env python3
"""LoRA fine-tune throughput on one MI355X (BASELINE.json configs[3] per-GPU share: 8 synthetic 2HANDS samples per
micro-batch, 96-id conversations (351 expanded tokens), 1024^2 masks, bf16). Prints samples/s for fwd+bwd+AdamW."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa
from haff import checkpoint, config as hcfg, train_ops as T
from haff.train_model import LisaTrainable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="7b")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ids", type=int, default=96)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--mask", type=int, default=1024)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = {"tiny": hcfg.tiny, "mid": hcfg.mid, "7b": hcfg.haff_7b, "13b": hcfg.haff_13b}[args.config]()
    sd = checkpoint.synthetic_state_dict(cfg, 1234, dev, torch.bfloat16)
    model = LisaTrainable(cfg, sd, dtype=torch.bfloat16, device=dev)
    del sd
    torch.cuda.empty_cache()
    b = args.batch
    import bench
    batch = bench.make_train_batch(cfg, b, args.ids, (args.mask, args.mask), dev)
    states = {k: T.AdamWState(p) for k, p in model.named_parameters()}
    print("trainable params %.1f M, HBM allocated %.1f GB" % (sum(p.numel() for p in model.parameters()) / 1e6,
                                                               torch.cuda.memory_allocated() / 2 ** 30), flush=True)

    def step():
        model.zero_grad()
        t0 = time.perf_counter()
        out = model(**batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out["loss"].backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        norm = float(T.grad_norm(grads))
        for k, p in model.named_parameters():
            if p.grad is not None:
                T.adamw_step(states[k], p.grad, lr=3e-4, gscale=min(1.0, 1.0 / (norm + 1e-6)), param_lp=p.data)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        return float(out["loss"].detach()), t1 - t0, t2 - t1, t3 - t2
    step()
    tot = [0.0, 0.0, 0.0]
    for i in range(args.steps):
        loss, f, bw, o = step()
        tot = [tot[0] + f, tot[1] + bw, tot[2] + o]
        print(f"step {i}: loss {loss:.4f} fwd {f*1e3:.0f} ms bwd {bw*1e3:.0f} ms opt {o*1e3:.0f} ms", flush=True)
    t = sum(tot) / args.steps
    print("samples/s/GPU %.2f  (fwd %.0f ms, bwd %.0f ms, opt %.0f ms per micro-batch of %d; peak HBM %.1f GB)" %
          (b / t, tot[0] / args.steps * 1e3, tot[1] / args.steps * 1e3, tot[2] / args.steps * 1e3, b, torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == "__main__":
    main()
