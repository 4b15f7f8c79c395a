"""Time one KV-cached decode step of the 7B language model at several batch sizes (weights are streamed once per step:
HBM-bound).   usage: python tools/decode_step_bench.py [--config 7b] [--batches 1,8,16,32,64]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa: F401
from haff import checkpoint, config as hcfg
from haff.lisa import LisaMI355

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="7b")
    ap.add_argument("--batches", default="1,8,16,32,64")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--no-chain", action="store_true", help="<= 8 rows: five launches per layer instead of the one chained launch per step")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = {"7b": hcfg.haff_7b, "13b": hcfg.haff_13b}[args.config]()
    model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), dtype=torch.bfloat16, device=dev)
    model.decode_graphs = not args.eager
    model.llm.decode_chain = not args.no_chain
    l = cfg.llm
    w_bytes = 2.0 * (l.layers * (4 * l.hidden * l.hidden + 3 * l.hidden * l.ffn) + l.vocab * l.hidden)
    T0 = 36 + 255
    for B in [int(b) for b in args.batches.split(",")]:
        cache = model._persistent_cache(B, T0 + 8)
        tok = torch.zeros((B,), dtype=torch.long, device=dev)
        def one_step():
            cache["pos"].fill_(T0); cache["nk"].fill_(T0 + 1)
            return model._decode_step(tok, cache)
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            one_step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t1) / 10
        print(f"batch {B:3d}: {ms:7.3f} ms/step   weight stream {w_bytes / (ms * 1e-3) / 1e12:5.2f} TB/s", flush=True)

if __name__ == "__main__":
    main()
