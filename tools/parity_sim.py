"""CPU study (no GPU): how close can a bf16 pipeline get to the fp32 oracle's binary masks, and what does an fp32
decoder tail / a smooth ("separated-logit") weight + frame variant buy?  Emulation: the oracle run with every tensor
in torch.bfloat16 on the CPU (each op rounds its output to bf16, fp32 accumulate) — pessimistic w.r.t. the HIP path,
which keeps norm statistics and softmax in fp32.

    python tools/parity_sim.py [tiny|mid] [--smooth]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import haff  # noqa
from haff import config as hcfg, weights as hw
from oracle import lisa_oracle as O

V = "model.visual_model"


class Bf16Emu(torch.overrides.TorchFunctionMode):
    """Round the output of every contraction / activation / norm to bf16 values (kept in fp32 storage)."""
    NAMES = {"linear", "conv2d", "conv_transpose2d", "matmul", "einsum", "layer_norm", "gelu", "silu", "relu", "softmax",
             "bmm", "add", "mul", "__matmul__", "__add__", "__mul__"}

    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if getattr(func, "__name__", "") in self.NAMES and torch.is_tensor(out) and out.dtype == torch.float32:
            return out.to(torch.bfloat16).to(torch.float32)
        return out


def iou(a, b):
    i = (a & b).sum().item(); u = (a | b).sum().item()
    return i / u if u else 1.0


def run(cfg_name, smooth, seed=5):
    cfg = getattr(hcfg, cfg_name)()
    sd = hw.make_state_dict(cfg, seed, variant="separated" if smooth else None) if "variant" in hw.make_state_dict.__code__.co_varnames else hw.make_state_dict(cfg, seed)
    hw.round_to_bf16_(sd)
    rng = np.random.default_rng(seed + 7)
    S = cfg.sam.img_size
    B = 2
    if smooth and hasattr(hw, "smooth_frames"):
        images = hw.smooth_frames(B, S, seed)
    else:
        images = torch.from_numpy(rng.standard_normal((B, 3, S, S), dtype=np.float32))
    images = images.to(torch.bfloat16).float()
    images_clip = torch.from_numpy(rng.standard_normal((B, 3, 224, 224), dtype=np.float32)).to(torch.bfloat16).float()
    text = torch.from_numpy(rng.integers(3, cfg.llm.vocab - 3, size=(B, 8))).long()
    ids = torch.cat([torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]]).expand(B, -1), text], 1)
    forced = torch.from_numpy(rng.integers(3, cfg.llm.vocab - 3, size=(B, 4))).long()
    forced[:, 1] = cfg.seg_token_idx
    forced[:, -1] = cfg.eos_token_id
    sz = [(S, S)] * B
    with torch.no_grad():
        taps = {}
        _, l32, r32, t32 = O.lisa_evaluate(sd, cfg, images_clip, images, ids, sz, sz, 4, forced, use_cache=True, taps=taps)
        tb = {}
        with Bf16Emu():
            _, lb, rb, tb_ = O.lisa_evaluate(sd, cfg, images_clip, images, ids, sz, sz, 4, forced, use_cache=True, taps=tb)
        # bf16 encoder + LLM, fp32 decoder tail
        grid = (cfg.sam.grid,) * 2
        pe = O.sam_dense_pe(sd, V + ".prompt_encoder", grid)
        lm, rm = [], []
        for i in range(B):
            sp, de = O.sam_prompt_encoder_text(sd, V + ".prompt_encoder", tb["pred_embeddings"][i].float().unsqueeze(1), grid)
            e = tb["image_embeddings"][i:i + 1].float()
            lo, _, _ = O.sam_mask_decoder(sd, V + ".mask_decoder_left", e, pe, sp, de, True)
            lm.append(O.sam_postprocess_masks(lo, S, sz[i], sz[i])[:, 0])
            lo, _ = O.sam_mask_decoder(sd, V + ".mask_decoder_right", e, pe, sp, de, False)
            rm.append(O.sam_postprocess_masks(lo, S, sz[i], sz[i])[:, 0])
    e32, eb = taps["image_embeddings"], tb["image_embeddings"].float()
    print(f"{cfg_name} smooth={smooth}: emb rel err {((e32-eb).abs().max()/e32.abs().max()).item():.3e}  "
          f"pred_emb rel err {((taps['pred_embeddings'][0]-tb['pred_embeddings'][0].float()).abs().max()/taps['pred_embeddings'][0].abs().max()).item():.3e}")
    for i in range(B):
        for nm, a, b, c in (("L", l32[i], lb[i].float(), lm[i]), ("R", r32[i], rb[i].float(), rm[i])):
            sc = a.abs().max().item()
            near = (a.abs() < 0.01 * sc).float().mean().item()
            print(f"  frame{i} {nm}: pos frac {(a>0).float().mean():.3f}  |logit|<1%max: {near:.4f}  IoU all-bf16 {iou(a>0,b>0):.5f}  "
                  f"IoU fp32-tail {iou(a>0,c>0):.5f}  relerr bf16 {((a-b).abs().max()/sc).item():.3e} tail {((a-c).abs().max()/sc).item():.3e}")


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "tiny"
    run(name, "--smooth" in sys.argv)
