"""Soak: the two-stream schedule (SAM encoder beside the language model) against the single-stream result, bitwise, over
many runs of the full 7B geometry.   usage: python tools/two_stream_soak.py [runs] [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa: F401
from bench import make_inputs
from haff import checkpoint, config as hcfg
from haff.lisa import LisaMI355

def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda:0")
    cfg = hcfg.haff_7b()
    model = LisaMI355(cfg, checkpoint.synthetic_state_dict(cfg, 1234, dev), dtype=torch.bfloat16, device=dev, sam_chunk=4)
    S = cfg.sam.img_size
    frames, clip, ids, forced = make_inputs(cfg, B, 32, 8, dev)
    sizes = [(S, S)] * B
    def run():
        with torch.no_grad():
            o, l, r, t = model.evaluate(clip, None, ids, sizes, sizes, max_new_tokens=8, forced_answer=forced, frames_u8=frames)
        torch.cuda.synchronize()
        return [o.clone()] + [m.clone() for m in l] + [m.clone() for m in r] + [x.clone() for x in t]
    model.overlap_streams = False
    ref = run()
    model.overlap_streams = True
    bad = 0
    for i in range(runs):
        cur = run()
        bad += int(not all(torch.equal(a, b) for a, b in zip(cur, ref)))
        if (i + 1) % 25 == 0:
            print(f"{i + 1} two-stream runs, {bad} differ from the single-stream result", flush=True)
    print(f"batch {B}: {bad}/{runs} two-stream runs differ from the single-stream result")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
