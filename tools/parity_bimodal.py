#!/usr/bin/env python3
"""Parity on a TRAINED-LIKE (bimodal) logit field. A random-init decoder's logit field is Gaussian around 0:
a perturbation eps flips ~0.8 eps / sigma of the pixels whatever the kernels do. A trained checkpoint's field
is two plateaus with a thin boundary. This construction gets that field out of random stacks without touching the forward pass:

  frame      two flat regions (left / right of a wavy vertical boundary), so the decoder's upscaled embedding U(x) [32 x h x w]
             clusters around two vectors u_A, u_B (measured on the ORACLE's own fp32 forward, away from the boundary);
  weights    the bias of the last layer of output_hypernetworks_mlps.0 of each decoder is shifted so that the hypernetwork output
             becomes the minimum-norm h' with h'.u_A = +M, h'.u_B = -M (M = 10): logit(x) = h'.U(x) is +-M on the plateaus;
  check      both the oracle and the HIP path run the SAME modified weights; IoU of (logit > 0) as everywhere else.

CPU part (this file run bare): builds the case for a list of seeds and prints how bimodal the oracle's field came out (fraction of
pixels with |logit| < 1 % / 5 % of M, plateau spread) — the evidence of whether the construction is robust. bench.py's parity object
and tests/test_lisa_gpu.py import build_case()."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa: E402,F401
from haff import config as hcfg  # noqa: E402
from haff import weights as hw  # noqa: E402

V = "model.visual_model"
M_PLATEAU = 10.0


def two_region_frame(S, seed, contrast=1.0):
    """[1, 3, S, S] fp32 (bf16-representable): two flat colours either side of x = S/2 + S/10 sin(2 pi y / S), + 2 % noise."""
    rng = np.random.default_rng(seed)
    ca, cb = contrast * rng.uniform(-1.5, -0.5, 3), contrast * rng.uniform(0.5, 1.5, 3)
    yy, xx = np.mgrid[0:S, 0:S]
    left = xx < S / 2 + S / 10 * np.sin(2 * np.pi * yy / S)
    img = np.where(left[None], ca[:, None, None], cb[:, None, None]) + 0.02 * rng.standard_normal((3, S, S))
    return torch.from_numpy(img[None].astype(np.float32)).to(torch.bfloat16).float(), torch.from_numpy(left)


def two_region_frame_u8(S, seed):
    """[1, S, S, 3] uint8 NHWC (what the timed step ingests) + the boolean left-region map: two flat colours either side of the same
    wavy boundary, + noise of 5 grey levels."""
    rng = np.random.default_rng(seed)
    ca, cb = rng.uniform(40, 100, 3), rng.uniform(150, 220, 3)
    yy, xx = np.mgrid[0:S, 0:S]
    left = xx < S / 2 + S / 10 * np.sin(2 * np.pi * yy / S)
    img = np.where(left[..., None], ca[None, None, :], cb[None, None, :]) + 5.0 * rng.standard_normal((S, S, 3))
    return torch.from_numpy(np.clip(np.rint(img), 0, 255).astype(np.uint8))[None], torch.from_numpy(left)


def plateau_bias(O, sd, pfx, image_embeddings, pe, sp, de, tax_on, left_region):
    """The re-aimed last-layer bias of output_hypernetworks_mlps.0 of decoder `pfx` (see the module docstring): from the ORACLE's own
    fp32 decoder pass on (image_embeddings, prompt) -> (state-dict key, new bias (bf16-representable fp32), diagnostics)."""
    t = {}
    O.sam_mask_decoder(sd, pfx, image_embeddings, pe, sp, de, tax_on, taps=t)
    U, h = t["upscaled"][0], t["hyper"][0, 0]                 # [32, hh, ww], [32]
    hh = U.shape[1]
    region = torch.nn.functional.interpolate(left_region[None, None].float(), size=(hh, hh), mode="area")[0, 0]
    inA, inB = region > 0.999, region < 0.001                 # low-res cells wholly inside a region
    XA, XB = U[:, inA].double(), U[:, inB].double()           # [32, nA], [32, nB]
    uA, uB = XA.mean(1), XB.mean(1)
    # Fisher's direction: the separation of the two clusters measured in units of their own scatter is largest along
    # Sw^-1 (uA - uB); a second vector Sw^-1 (uA + uB) carries the offset so that the plateaus sit at +M and -M
    Sw = ((XA - uA[:, None]) @ (XA - uA[:, None]).T + (XB - uB[:, None]) @ (XB - uB[:, None]).T) / (XA.shape[1] + XB.shape[1])
    Sw = Sw + 1e-6 * torch.trace(Sw) / Sw.shape[0] * torch.eye(Sw.shape[0], dtype=torch.float64)
    w1, w2 = torch.linalg.solve(Sw, uA - uB), torch.linalg.solve(Sw, uA + uB)
    A2 = torch.stack([torch.stack([w1 @ uA, w2 @ uA]), torch.stack([w1 @ uB, w2 @ uB])])
    ab = torch.linalg.solve(A2, torch.tensor([M_PLATEAU, -M_PLATEAU], dtype=torch.float64))
    hp = (ab[0] * w1 + ab[1] * w2).float()
    key = f"{pfx}.output_hypernetworks_mlps.0.layers.2.bias"
    new_bias = (sd[key] + (hp - h)).to(torch.bfloat16).float()
    return key, new_bias, {"plateau_gap_in_U": float((uA - uB).float().norm()), "hyper_norm": float(hp.norm())}


def build_case(cfg_name="tiny", seed=3, contrast=1.0):
    """-> dict(cfg, sd (modified, bf16-representable), images, images_clip, ids, forced, diag) ; the oracle is the only forward used."""
    from oracle import lisa_oracle as O
    cfg = getattr(hcfg, cfg_name)()
    sd = hw.round_to_bf16_(hw.make_state_dict(cfg, seed))
    S = cfg.sam.img_size
    images, left = two_region_frame(S, seed, contrast)
    rng = np.random.default_rng(seed)
    images_clip = torch.from_numpy(rng.standard_normal((1, 3, 224, 224), dtype=np.float32)).to(torch.bfloat16).float()
    ids = torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx, 11, 12, 13, 14, 15, 16, 17, 18]])
    forced = torch.tensor([[7, cfg.seg_token_idx, 9, cfg.eos_token_id]])
    diag = {}
    with torch.no_grad():
        taps = {}
        O.lisa_evaluate(sd, cfg, images_clip, images, ids, [(S, S)], [(S, S)], max_new_tokens=4, forced_answer=forced, taps=taps)
        g = (cfg.sam.grid,) * 2
        pe = O.sam_dense_pe(sd, V + ".prompt_encoder", g)
        sp, de = O.sam_prompt_encoder_text(sd, V + ".prompt_encoder", taps["pred_embeddings"][0].unsqueeze(1), g)
        for side, tax in (("left", True), ("right", False)):
            key, new_bias, diag[side] = plateau_bias(O, sd, f"{V}.mask_decoder_{side}", taps["image_embeddings"], pe, sp, de, tax, left)
            sd[key] = new_bias
        r_ids, r_left, r_right, r_tax = O.lisa_evaluate(sd, cfg, images_clip, images, ids, [(S, S)], [(S, S)], max_new_tokens=4,
                                                         forced_answer=forced)
    for side, m in (("left", r_left[0]), ("right", r_right[0])):
        a = m.abs() / M_PLATEAU
        diag[side].update({"frac_within_1pct_of_threshold": float((a < 0.01).float().mean()), "frac_within_5pct": float((a < 0.05).float().mean()),
                           "frac_within_20pct": float((a < 0.2).float().mean()), "positive_frac": float((m > 0).float().mean()),
                           "plateau_median_abs_logit": float(m.abs().median()), "logit_min_max": [float(m.min()), float(m.max())]})
    return {"cfg": cfg, "sd": sd, "images": images, "images_clip": images_clip, "ids": ids, "forced": forced, "diag": diag,
            "oracle": (r_ids, r_left, r_right, r_tax)}


if __name__ == "__main__":
    for cfg_name in ("tiny", "mid"):
        for seed in (3, 4, 5, 6, 7, 8):
            c = build_case(cfg_name, seed)
            print(cfg_name, "seed", seed, {k: {kk: (round(vv, 4) if isinstance(vv, float) else [round(x, 2) for x in vv]) for kk, vv in v.items()} for k, v in c["diag"].items()}, flush=True)
