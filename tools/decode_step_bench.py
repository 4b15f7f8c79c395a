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
    ap.add_argument("--prefold", action="store_true", help="build the norm-folded weight copies before the first cache is allocated")
    ap.add_argument("--dummy-cache", type=int, default=0)
    ap.add_argument("--dummy-step", action="store_true")
    ap.add_argument("--dummy-mode", default="graph", choices=["graph", "eager_chain", "eager_lib"])
    ap.add_argument("--dummy-reps", type=int, default=1)
    ap.add_argument("--check", action="store_true", help="after timing: the chained step against itself eager / stage by stage / the five-launch layer")
    ap.add_argument("--fresh-pool", action="store_true", help="every configuration captures its graph into a memory pool of its own")
    ap.add_argument("--place", action="store_true", help="-DCH_PLACE build: where and when layer 2's workgroups of the last launch started")
    ap.add_argument("--addr", action="store_true", help="print the device addresses of the chained step's scratch buffers")
    ap.add_argument("--no-chain", action="store_true", help="<= 8 rows: five launches per layer instead of the one chained launch per step")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = {"7b": hcfg.haff_7b, "13b": hcfg.haff_13b}[args.config]()
    model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), dtype=torch.bfloat16, device=dev)
    model.decode_graphs = not args.eager
    model.llm.decode_chain = not args.no_chain
    if args.prefold:
        model.llm.fold_norm_weights()
        torch.cuda.synchronize()
    l = cfg.llm
    w_bytes = 2.0 * (l.layers * (4 * l.hidden * l.hidden + 3 * l.hidden * l.ffn) + l.vocab * l.hidden)
    T0 = 36 + 255
    if args.dummy_cache:   # experiment: the first KV cache a process allocates is not one that is timed
        dummy = model._persistent_cache(args.dummy_cache, T0 + 8)
        if args.dummy_step:
            tok = torch.zeros((args.dummy_cache,), dtype=torch.long, device=dev)
            dummy["pos"].fill_(T0); dummy["nk"].fill_(T0 + 1)
            if args.dummy_mode == "graph":
                model._decode_step(tok, dummy); model._decode_step(tok, dummy)
            else:
                model.decode_graphs = False
                model.llm.decode_chain = args.dummy_mode == "eager_chain"
                for _ in range(args.dummy_reps):
                    model._decode_step(tok, dummy)
                model.decode_graphs = not args.eager
                model.llm.decode_chain = not args.no_chain
            torch.cuda.synchronize()
    for B in [int(b) for b in args.batches.split(",")]:
        if args.fresh_pool:
            model._graph_pool = None
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
        if args.check and "chain" in cache:
            from haff import ops
            ok = ops.decode_chain_status(cache["chain"]["sync"], cfg.llm.layers)
            h_graph = one_step()[0].float()
            model.decode_graphs = False
            h_eager = one_step()[0].float()
            model.llm.decode_chain = "stages"
            h_stages = one_step()[0].float()
            model.llm.decode_chain = False
            h_lib = one_step()[0].float()
            model.llm.decode_chain = not args.no_chain
            model.decode_graphs = not args.eager
            sc = h_lib.abs().max().item()
            print(f"   status ok {ok}; finite {bool(torch.isfinite(h_graph).all())}; graph vs eager chain {(h_graph - h_eager).abs().max().item():.3e}, vs stage-by-stage "
                  f"{(h_graph - h_stages).abs().max().item():.3e}, vs five launches {(h_graph - h_lib).abs().max().item():.3e} (scale {sc:.3e})", flush=True)
        if args.place and "chain" in cache:
            import ctypes
            from collections import Counter
            from haff import lib as hlib
            pr = ctypes.CDLL(hlib.LIB_PATH).haff_decode_chain_place_read
            pr.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            pb, pt = (ctypes.c_uint * (5 * 1024))(), (ctypes.c_ulonglong * (5 * 1024))()
            pr(pb, pt)
            nbs = [3 * l.hidden // 16, B * l.heads, l.hidden // 8, 2 * l.ffn // 32, l.hidden // 8]
            t0 = min(pt[i] for i in range(min(nbs[0], 1024)))
            for st, name in enumerate(("qkv", "attn", "o_proj", "gate|up", "down")):
                n = min(nbs[st], 1024)
                keys = [((pb[st * 1024 + r] >> 16) & 15, (pb[st * 1024 + r] >> 13) & 7, (pb[st * 1024 + r] >> 12) & 1, (pb[st * 1024 + r] >> 8) & 15) for r in range(n)]
                c = Counter(keys)
                ts = sorted((pt[st * 1024 + r] - t0) / 100.0 for r in range(n))
                xcd_of_first8 = [k[0] for k in keys[:8]]
                print(f"   {name:8s} {n:4d} wgs on {len(c):3d} CUs, per CU {dict(sorted(Counter(c.values()).items()))}; start times us: first {ts[0]:.1f} median {ts[n // 2]:.1f} last {ts[-1]:.1f}; XCDs of workgroups 0..7: {xcd_of_first8}", flush=True)
        if args.addr and "chain" in cache:
            ch = cache["chain"]
            ent = model._graphs.get((B, cache["tmax"]))
            ptrs = {k: ch[k].data_ptr() for k in ("qkv", "att", "g", "ws", "sync")}
            ptrs["ssq_a"], ptrs["ssq_b"] = ch["ssq_a"].data_ptr(), ch["ssq_b"].data_ptr()
            ptrs["k0"], ptrs["nk"] = cache["k"][0].data_ptr(), cache["nk"].data_ptr()
            if ent is not None:
                ptrs["graph_in"], ptrs["graph_h1"] = ent[1].data_ptr(), ent[2].data_ptr()
            print("   " + "  ".join(f"{k}={v:#x}" for k, v in ptrs.items()), flush=True)

if __name__ == "__main__":
    main()
