"""Is the SAM encoder bit-stable while KV-cached decode steps of the language model run on another HIP stream, and are
the decode steps bit-stable beside the encoder? (the full two-stream schedule is not; this asks whether
the decode-only part of the overlap is.)   usage: python tools/overlap_decode_probe.py [runs]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa: F401
from bench import make_inputs
from haff import checkpoint, config as hcfg
from haff.lisa import LisaMI355

def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    cfg = hcfg.haff_7b()
    model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), dtype=torch.bfloat16, device=dev, sam_chunk=8)
    model.decode_graphs = False
    B = 64
    frames, clip, ids, forced = make_inputs(cfg, 8, 32, 8, dev)
    enc = model.sam_encoder
    from haff.preprocess import SAM_MEAN, SAM_STD
    T0 = 36 + 255
    cache = model._persistent_cache(B, T0 + 8)
    tok = torch.arange(B, device=dev) % 1000 + 5
    def decode_chain():
        outs = []
        cache["pos"].fill_(T0); cache["nk"].fill_(T0 + 1)
        t = tok
        for _ in range(8):
            h, t = model._decode_step(t, cache)
            outs.append(h)
        return torch.cat(outs, 1)
    def sam():
        return enc.forward_rows(enc.patch_rows_from_u8(frames, SAM_MEAN, SAM_STD), frames.shape[0])
    with torch.no_grad():
        d_ref = decode_chain().clone(); e_ref = sam().clone(); torch.cuda.synchronize()
        d2 = decode_chain(); e2 = sam(); torch.cuda.synchronize()
        print("serial repeat: decode", torch.equal(d_ref, d2), "sam", torch.equal(e_ref, e2), flush=True)
        side = torch.cuda.Stream(dev)
        bad_e = bad_d = 0
        for i in range(runs):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                e = sam()
            ds = [decode_chain() for _ in range(3)]      # ~3 x 8 decode steps beside one 8-frame encoder pass
            torch.cuda.synchronize()
            be = not torch.equal(e, e_ref); bd = sum(int(not torch.equal(d, d_ref)) for d in ds)
            bad_e += int(be); bad_d += bd
            if be or bd:
                print(f"run {i}: sam differs {be} ({int((e != e_ref).sum())} elems)  decode chains differing {bd}/3", flush=True)
        print(f"{runs} overlapped runs: sam encoder differed in {bad_e}, decode chains differed in {bad_d} of {3 * runs}")

if __name__ == "__main__":
    main()
