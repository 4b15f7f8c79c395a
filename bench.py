#!/usr/bin/env python3
"""Benchmark of the 2Haff per-frame affordance path on MI355X (contract: see the task's bench.py section).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (LisaMI355.evaluate: CLIP -> projector -> Llama prefill + KV-cached greedy
decode -> [SEG] -> SAM ViT-H encoder -> left/right mask decoders -> postprocess) over one batch of synthetic
frames already resident in HBM. Default workload = BASELINE.json configs[2]: 2HandedAfforder-7B, 64 x 1024^2
uint8 NHWC frames, 32-token prompts (L=36, T=291), 8 forced answer tokens with [SEG], bf16. Frames are
independent units: for N>1 every rank processes its own batch (weak scaling, no data-path collective).
One JSON line on rank 0: metric/value (+ roofline of the dominant kernel, + cpu_baseline = the CPU oracle
timed on a bounded, depth-reduced sample and extrapolated by layer counts).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import haff  # noqa: E402,F401
from haff import config as hcfg  # noqa: E402
from haff import dist as hdist  # noqa: E402
from haff import flops as hflops  # noqa: E402
from haff import ops  # noqa: E402
from haff import weights as hw  # noqa: E402
from haff.lisa import LisaMI355  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md §Chip-level parameters)
SURVEY_FLOPS = {"2HandedAfforder-7B": 10.01e12, "2HandedAfforder-13B": 13.73e12}


def make_inputs(cfg, B, text_tokens, n_gen, device, seed=1234):
    g = torch.Generator(device="cpu").manual_seed(seed)
    S = cfg.sam.img_size
    noise = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.int32).float()
    yy = torch.linspace(0, 255, S).view(1, S, 1, 1)
    xx = torch.linspace(255, 0, S).view(1, 1, S, 1)
    frames = (0.5 * noise + 0.25 * yy + 0.25 * xx).round().clamp(0, 255).to(torch.uint8)
    images_clip = torch.randn((B, 3, cfg.clip.image, cfg.clip.image), generator=g).to(torch.bfloat16)
    hi = min(cfg.llm.vocab, cfg.seg_token_idx) - 1
    text = torch.randint(3, hi, (B, text_tokens), generator=g)
    head = torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]]).expand(B, -1)
    ids = torch.cat([head, text], 1).long()
    forced = torch.randint(3, hi, (B, n_gen), generator=g).long()
    forced[:, 2] = cfg.seg_token_idx
    forced[:, -1] = cfg.eos_token_id
    return frames.to(device), images_clip.to(device), ids.to(device), forced.to(device)


class GemmMeter:
    """HIP-event pairs around every bf16 GEMM launch (torch's current stream IS the launch stream)."""

    def __init__(self):
        self.records = []
        self._orig = None

    def __enter__(self):
        self._orig = ops.linear
        meter = self

        def timed(x, w, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = meter._orig(x, w, *a, **kw)
            e1.record()
            M, K, N = x.shape[0], x.shape[1], w.shape[0]
            n_out = N // 2 if kw.get("swiglu") else N
            byts = 2.0 * (M * K + N * K + M * n_out + (M * n_out if kw.get("resid") is not None else 0))
            if out.dtype == torch.float32:
                byts += 2.0 * M * n_out
            # two families with two rooflines: launches of <= 64 rows are decode steps / the [SEG] MLP — HBM-bound weight
            # streaming whichever kernel gemm_bf16_impl picks for them (the weight-streaming kernel, or since round 3 the split-K
            # tile path for wide / deep weights at 33..64 rows); everything larger is the MFMA-bound tile kernel
            stream = M <= 64 and not kw.get("tile_cfg")
            if x.dtype == torch.float32:
                stream = "f32"   # the fp32 decoder tail (f32-input MFMA GEMM): its own family, not part of the roofline kernel
            meter.records.append((e0, e1, 2.0 * M * K * N, byts, stream, 2.0 * N * K,
                                  "ln_fold" if kw.get("ln_stats") is not None else "plain", (M, N, K)))
            return out
        ops.linear = timed
        # the tile-kernel launches that do not go through ops.linear (round 3: residual products that emit the LayerNorm
        # statistics, the prefill q|k|v product with RoPE + cache append): same events, same bookkeeping
        self._orig_rs, self._orig_qr = ops.rowstats_gemm, ops.qkv_rope

        def timed_rs(x, w, bias, resid, out, a_map=None):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            part = meter._orig_rs(x, w, bias, resid, out, a_map)
            e1.record()
            M, K, N = out.shape[0], x.shape[1], w.shape[0]
            meter.records.append((e0, e1, 2.0 * M * K * N, 2.0 * (M * K + N * K + 2 * M * N) + 8.0 * M * (N // 64), False, 2.0 * N * K, "rowstats", (M, N, K)))
            return part

        def timed_qr(x, w_perm, kcache, vcache, cos_sin, B, T, H, d, pos0):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            q = meter._orig_qr(x, w_perm, kcache, vcache, cos_sin, B, T, H, d, pos0)
            e1.record()
            M, K, N = x.shape[0], x.shape[1], w_perm.shape[0]
            meter.records.append((e0, e1, 2.0 * M * K * N, 2.0 * (M * K + N * K + M * N), False, 2.0 * N * K, "qkv_rope", (M, N, K)))
            return q
        ops.rowstats_gemm, ops.qkv_rope = timed_rs, timed_qr
        self._orig_rs32 = ops.rowstats32_gemm

        def timed_rs32(x, w, bias, x32, out16, a_map=None):   # (round 6) the same products on the fp32 residual stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            part = meter._orig_rs32(x, w, bias, x32, out16, a_map)
            e1.record()
            M, K, N = out16.shape[0], x.shape[1], w.shape[0]
            meter.records.append((e0, e1, 2.0 * M * K * N, 2.0 * (M * K + N * K + 5 * M * N) + 8.0 * M * (N // 64), False, 2.0 * N * K, "rowstats32", (M, N, K)))
            return part
        ops.rowstats32_gemm = timed_rs32
        # round 5: the windowed q|k|v products scatter head-major (haff_gemm_bf16_heads): same tile kernel, same bookkeeping
        self._orig_lh = ops.linear_heads

        def timed_lh(x, w, bias, row_map, out, d, heads, part_stride, head_stride, ln_stats=None, ln_colsum=None):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = meter._orig_lh(x, w, bias, row_map, out, d, heads, part_stride, head_stride, ln_stats=ln_stats, ln_colsum=ln_colsum)
            e1.record()
            M, K, N = x.shape[0], x.shape[1], w.shape[0]
            meter.records.append((e0, e1, 2.0 * M * K * N, 2.0 * (M * K + N * K + M * N), False, 2.0 * N * K,
                                  "ln_fold" if ln_stats is not None else "plain", (M, N, K)))
            return r
        ops.linear_heads = timed_lh
        return self

    def __exit__(self, *exc):
        ops.linear = self._orig
        ops.rowstats_gemm, ops.qkv_rope = self._orig_rs, self._orig_qr
        ops.rowstats32_gemm = self._orig_rs32
        ops.linear_heads = self._orig_lh

    def summary(self):
        """(launches, ms, flop, algorithmic bytes) of the MFMA tile kernel launches."""
        torch.cuda.synchronize()
        rec = [r for r in self.records if r[4] is False]
        ms = sum(r[0].elapsed_time(r[1]) for r in rec)
        fl = sum(r[2] for r in rec)
        by = sum(r[3] for r in rec)
        return len(rec), ms, fl, by

    def fused_summary(self):
        """Tile-kernel launches whose epilogue carries extra work (round 3): {kind: (launches, ms, flop)}; and the plain ones."""
        out = {}
        for r in self.records:
            if r[4] is False:
                kind = r[6]
                n, ms, fl = out.get(kind, (0, 0.0, 0.0))
                out[kind] = (n + 1, ms + r[0].elapsed_time(r[1]), fl + r[2])
        return out

    def shape_summary(self, top=12):
        """The tile-kernel launches grouped by (M, N, K, epilogue kind), heaviest first."""
        agg = {}
        for r in self.records:
            if r[4] is False:
                key = r[7] + (r[6],)
                n, ms, fl = agg.get(key, (0, 0.0, 0.0))
                agg[key] = (n + 1, ms + r[0].elapsed_time(r[1]), fl + r[2])
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]
        return [{"M": k[0], "N": k[1], "K": k[2], "epilogue": k[3], "launches_per_step": n, "ms_per_step": round(ms, 3),
                 "achieved": round(fl / (ms * 1e-3) / 1e12, 1)} for k, (n, ms, fl) in rows]

    def stream_summary(self):
        """(launches, ms, weight bytes) of the weight-streaming (M <= 64) launches."""
        rec = [r for r in self.records if r[4] is True]
        return len(rec), sum(r[0].elapsed_time(r[1]) for r in rec), sum(r[5] for r in rec)

    def f32_summary(self):
        """(launches, ms, flop) of the fp32 decoder-tail GEMMs."""
        rec = [r for r in self.records if r[4] == "f32"]
        return len(rec), sum(r[0].elapsed_time(r[1]) for r in rec), sum(r[2] for r in rec)

    def total_gemm_flop(self):
        return sum(r[2] for r in self.records)


def parity_vs_oracle(device):
    """The 'mask IoU vs ref' half of the metric: BASELINE.json configs[0] (tiny LISA, the reference's own CPU-runnable case)
    and the mid geometry through the HIP path against the CPU oracle on the same seeded weights and inputs, in three numeric
    configurations: bf16 (what the throughput line measures: bf16 MFMA stacks + fp32 decoder tail), bf16_all (decoder tail in
    bf16 too: the round-1 path) and fp32. Beside each IoU: how many pixels disagree and how far from zero the oracle's logit is
    at those pixels, in units of the logit field's standard deviation — and the floor, the IoU an EXACT fp32 pipeline reaches
    when nothing but the image embedding is rounded to bf16 once. The oracle is the checker only."""
    import numpy as np
    from oracle import lisa_oracle as O
    V = "model.visual_model"
    out = {"oracle": "oracle/lisa_oracle.py (fp32, CPU)", "weights": "seeded random init, rounded to bf16",
           "note": "random-init logits are Gaussian around 0 (no saturated inside/outside plateaus as trained checkpoints have): "
                   "the fraction of pixels a perturbation flips is ~0.8 x |logit error| / logit std, so IoU >= 0.999 needs a "
                   "relative logit error <= 6e-4 — below one bf16 rounding (see iou_floor_*); DESIGN.md section 2"}
    for cfg_name in ("tiny", "mid"):
        cfg = getattr(hcfg, cfg_name)()
        sd = hw.round_to_bf16_(hw.make_state_dict(cfg, 3))
        rng = np.random.default_rng(3)
        S = cfg.sam.img_size
        images = torch.from_numpy(rng.standard_normal((1, 3, S, S), dtype=np.float32)).to(torch.bfloat16).float()
        images_clip = torch.from_numpy(rng.standard_normal((1, 3, 224, 224), dtype=np.float32)).to(torch.bfloat16).float()
        ids = torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx, 11, 12, 13, 14, 15, 16, 17, 18]])
        forced = torch.tensor([[7, cfg.seg_token_idx, 9, cfg.eos_token_id]])
        taps = {}
        with torch.no_grad():
            r_ids, r_left, r_right, r_tax = O.lisa_evaluate(sd, cfg, images_clip, images, ids, [(S, S)], [(S, S)],
                                                             max_new_tokens=4, forced_answer=forced, taps=taps)
            r_greedy = O.lisa_evaluate(sd, cfg, images_clip, images, ids, [(S, S)], [(S, S)], max_new_tokens=8)[0]
            # floor: exact fp32 everywhere, the image embedding alone rounded to bf16 once
            g = (cfg.sam.grid,) * 2
            pe = O.sam_dense_pe(sd, V + ".prompt_encoder", g)
            sp, de = O.sam_prompt_encoder_text(sd, V + ".prompt_encoder", taps["pred_embeddings"][0].unsqueeze(1), g)
            e16 = taps["image_embeddings"].to(torch.bfloat16).float()
            fl_l = O.sam_postprocess_masks(O.sam_mask_decoder(sd, V + ".mask_decoder_left", e16, pe, sp, de, True)[0], S, (S, S), (S, S))[:, 0]
            fl_r = O.sam_postprocess_masks(O.sam_mask_decoder(sd, V + ".mask_decoder_right", e16, pe, sp, de, False)[0], S, (S, S), (S, S))[:, 0]

        def stats(pairs):
            ious, errs, flips, margins = [], [], 0, []
            n_pix = 0
            for got, ref in pairs:
                gg, r = got.float().cpu(), ref
                a, b = gg > 0, r > 0
                inter, union = (a & b).sum().item(), (a | b).sum().item()
                ious.append(inter / union if union else 1.0)
                errs.append((gg - r).abs().max().item() / max(r.abs().max().item(), 1e-30))
                dis = a != b
                flips += int(dis.sum())
                n_pix += dis.numel()
                if dis.any():
                    margins.append((r[dis].abs().max() / r.std()).item())
            # IoU over the pixels OUTSIDE a thin band around the oracle's threshold (|oracle logit| >= 1 % / 2 % of its maximum):
            # on a trained checkpoint's two-plateau field that band is the object boundary and nothing else; here it says
            # whether every disagreement is a near-zero logit (1.0) or a real difference (< 1)
            band = {}
            for tau in (0.01, 0.02):
                ib, fr = [], []
                for got, ref in pairs:
                    gg, r = got.float().cpu(), ref
                    keep = r.abs() >= tau * r.abs().max()
                    a, b = (gg > 0) & keep, (r > 0) & keep
                    union = (a | b).sum().item()
                    ib.append((a & b).sum().item() / union if union else 1.0)
                    fr.append(1.0 - keep.float().mean().item())
                band["%g" % tau] = {"mask_iou": min(ib), "band_pixel_frac": max(fr)}
            return {"mask_iou_vs_oracle": min(ious), "mask_logit_max_err_rel": max(errs),
                    "pixels_disagreeing_frac": flips / n_pix,
                    "max_oracle_logit_at_disagreeing_pixels_in_logit_std": max(margins) if margins else 0.0,
                    "mask_iou_outside_threshold_band": band}
        res = {"iou_floor_exact_pipeline_with_bf16_rounded_embedding": stats(((fl_l, r_left[0]), (fl_r, r_right[0])))["mask_iou_vs_oracle"]}
        for name, dt, tail in (("bf16", torch.bfloat16, True), ("bf16_all", torch.bfloat16, False), ("fp32", torch.float32, True),
                               ("bf16_fp32_stream", torch.bfloat16, True)):
            # bf16_fp32_stream: the bf16 mode with the ViT-H and Llama residual streams kept in fp32 (LisaMI355(fp32_stream=True))
            model = LisaMI355(cfg, sd, dtype=dt, device=device, fp32_tail=tail, fp32_stream=name == "bf16_fp32_stream")
            o_ids, left, right, tax = model.evaluate(images_clip.to(device), images.to(device), ids.to(device), [(S, S)], [(S, S)],
                                                     max_new_tokens=4, forced_answer=forced)
            st = stats(((left[0], r_left[0]), (right[0], r_right[0])))
            st["taxonomy_max_abs_err"] = (tax[0].float().cpu() - r_tax[0]).abs().max().item()
            st["token_ids_equal"] = bool(torch.equal(o_ids.cpu(), r_ids))
            # free-running greedy decode (no forced answer): the argmax chain itself, 8 tokens
            g_ids = model.evaluate(images_clip.to(device), images.to(device), ids.to(device), [(S, S)], [(S, S)], max_new_tokens=8)[0]
            st["greedy_token_ids_equal"] = bool(g_ids.shape == r_greedy.shape and torch.equal(g_ids.cpu(), r_greedy))
            res[name] = st
            del model
        out[cfg_name] = res
    # Round 4: a frame of two flat regions + the hypernetwork bias re-aimed along Fisher's direction between
    # the two clusters of the oracle's upscaled embedding, plateaus at +-10 (tools/parity_bimodal.py; profiles/r4_parity_bimodal_cpu.txt
    # lists 12 seeds x 2 hands: no collapse — 45..59 % positive — but the clusters of a RANDOM encoder are only ~3 scatter widths
    # apart, so the field is still far from a trained checkpoint's: the number is reported beside the Gaussian-field one, not
    # instead of it)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import parity_bimodal as PB
        case = PB.build_case("tiny", 3)
        cfgb, S = case["cfg"], case["cfg"].sam.img_size
        model = LisaMI355(cfgb, case["sd"], dtype=torch.bfloat16, device=device, fp32_tail=True)
        o_ids, left, right, tax = model.evaluate(case["images_clip"].to(device), case["images"].to(device), case["ids"].to(device),
                                                 [(S, S)], [(S, S)], max_new_tokens=4, forced_answer=case["forced"])
        r_ids, r_left, r_right, _ = case["oracle"]
        st = stats(((left[0], r_left[0]), (right[0], r_right[0])))
        st["token_ids_equal"] = bool(torch.equal(o_ids.cpu(), r_ids))
        st["oracle_field"] = case["diag"]
        out["two_region_tiny_bf16"] = st
        del model
        model = LisaMI355(cfgb, case["sd"], dtype=torch.bfloat16, device=device, fp32_tail=True, fp32_stream=True)
        o_ids, left, right, tax = model.evaluate(case["images_clip"].to(device), case["images"].to(device), case["ids"].to(device),
                                                 [(S, S)], [(S, S)], max_new_tokens=4, forced_answer=case["forced"])
        st = stats(((left[0], r_left[0]), (right[0], r_right[0])))
        st["token_ids_equal"] = bool(torch.equal(o_ids.cpu(), r_ids))
        out["two_region_tiny_bf16_fp32_stream"] = st
        del model
    except Exception as e:   # the extra case must not take the benchmark line down
        out["two_region_tiny_bf16"] = {"error": repr(e)}
    # the keys round 1 reported, for continuity: configs[0]
    out["config"] = "BASELINE.json configs[0] (tiny) and the mid geometry, 1 frame each, forced answer with one [SEG]"
    out["bf16"], out["fp32"] = out["tiny"]["bf16"], out["tiny"]["fp32"]
    # the gate bench.py enforces (exit code 3 after the line is printed): fp32 mode meets BASELINE's targets outright; the
    # bf16 mode is held to the band the random-weight floor allows and to the forced-token identity
    fails = []
    for cfg_name in ("tiny", "mid"):
        f32, b16 = out[cfg_name]["fp32"], out[cfg_name]["bf16"]
        if f32["mask_iou_vs_oracle"] < 0.999 or f32["mask_logit_max_err_rel"] > 1e-3 or not f32["token_ids_equal"] or not f32["greedy_token_ids_equal"]:
            fails.append(f"{cfg_name}/fp32")
        if (b16["mask_iou_vs_oracle"] < 0.98 or b16["mask_logit_max_err_rel"] > 1e-2 or not b16["token_ids_equal"]
                or not b16["greedy_token_ids_equal"]):
            fails.append(f"{cfg_name}/bf16")
    out["gate"] = {"fp32": "IoU >= 0.999, logits within 1e-3, forced and free-running greedy tokens identical",
                   "bf16": "IoU >= 0.98, logits within 1e-2 of the logit scale (measured 3.7e-3 / 4.7e-3: ~2x), forced and "
                           "free-running greedy tokens identical",
                   "bf16_meets_iou_0.999_target": False,
                   "bf16_target_note": "BASELINE's IoU >= 0.999 is met by the fp32 mode only; on random-init weights no bf16 "
                                       "pipeline can meet it (iou_floor_*: the exact pipeline with ONE bf16 rounding of the embedding "
                                       "is already below it; the oracle with bf16 roundings at this path's kernel boundaries "
                                       "(oracle.bf16_points) sits at the same distance from the exact forward as the HIP path: "
                                       "tools/parity_points.py)",
                   "failed": fails}
    return out


def cpu_baseline(cfg, text_tokens, n_gen, threads):
    """The CPU oracle (fp32 torch eager restatement of the reference) on a bounded sample: full-width layers,
    reduced depth, one 1024^2 frame; per-layer times are extrapolated to the full depth."""
    import copy
    from oracle import lisa_oracle as O
    torch.set_num_threads(threads)
    V = "model.visual_model"
    s = copy.deepcopy(cfg.sam)
    s.depth, s.global_idx = 2, (1,)
    c = copy.deepcopy(cfg.clip)
    c.layers, c.select_layer = 2, 2
    small = copy.deepcopy(cfg)
    small.sam, small.clip = s, c
    small.llm = copy.deepcopy(cfg.llm)
    small.llm.layers = 1
    shapes = hw.all_shapes(small)
    sd = hw.make_state_dict(small, 99, shapes)
    g = torch.Generator().manual_seed(0)

    def tm(fn, reps=1):
        best = 1e30
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t)
        return best
    with torch.no_grad():
        S, gr, C = s.img_size, s.grid, s.embed_dim
        img = torch.randn((1, 3, S, S), generator=g)
        x = torch.randn((1, gr, gr, C), generator=g)
        E = V + ".image_encoder"
        t_all = tm(lambda: O.sam_image_encoder(sd, E, img, s))
        t_win = tm(lambda: O.sam_block(sd, E + ".blocks.0", x, s.heads, s.window))
        t_glob = tm(lambda: O.sam_block(sd, E + ".blocks.1", x, s.heads, 0))
        n_glob = len(cfg.sam.global_idx)
        t_sam = max(t_all - t_win - t_glob, 0.0) + (cfg.sam.depth - n_glob) * t_win + n_glob * t_glob
        ic = torch.randn((1, 3, c.image, c.image), generator=g)
        c1 = copy.deepcopy(c)
        c1.select_layer = 1
        t_c2 = tm(lambda: O.clip_vision_features(sd, "model.vision_tower.vision_tower", ic, c), 2)
        t_c1 = tm(lambda: O.clip_vision_features(sd, "model.vision_tower.vision_tower", ic, c1), 2)
        per_clip = max(t_c2 - t_c1, 1e-6)
        n_clip = cfg.clip.layers + 1 + cfg.clip.select_layer if cfg.clip.select_layer < 0 else cfg.clip.select_layer
        t_clip = max(t_c1 - per_clip, 0.0) + n_clip * per_clip
        T = 4 + text_tokens + cfg.clip.n_patches - 1
        xe = torch.randn((1, T, cfg.llm.hidden), generator=g)
        cache = [None]
        t_pre = tm(lambda: O.llama_forward(sd, xe, small.llm, [None]))
        O.llama_forward(sd, xe, small.llm, cache)
        x1 = torch.randn((1, 1, cfg.llm.hidden), generator=g)
        kv = cache[0]

        def dec():
            O.llama_forward(sd, x1, small.llm, [kv])
        t_dec = tm(dec, 3)
        hrow = torch.randn((1, cfg.llm.hidden), generator=g)
        t_head = tm(lambda: torch.nn.functional.linear(hrow, sd["lm_head.weight"]), 3)
        t_llm = cfg.llm.layers * (t_pre + (n_gen - 1) * t_dec) + n_gen * t_head
        t_llm_ref = cfg.llm.layers * n_gen * t_pre + n_gen * t_head  # reference: no KV cache (LISA.py:115)
        emb = torch.randn((1, s.out_chans, gr, gr), generator=g)
        pe = O.sam_dense_pe(sd, V + ".prompt_encoder", (gr, gr))
        txt = torch.randn((1, 1, s.out_chans), generator=g)

        def decs():
            sp, de = O.sam_prompt_encoder_text(sd, V + ".prompt_encoder", txt, (gr, gr))
            for side, tax in (("left", True), ("right", False)):
                lo = O.sam_mask_decoder(sd, f"{V}.mask_decoder_{side}", emb, pe, sp, de, tax)[0]
                O.sam_postprocess_masks(lo, S, (S, S), (S, S))
        t_dec2 = tm(decs)
    t_frame = t_sam + t_clip + t_llm + t_dec2
    t_frame_ref = t_sam + n_gen * t_clip + t_llm_ref + t_dec2
    return {
        "value": 1.0 / t_frame, "unit": "frames/s", "cores": threads, "kind": "port",
        "extrapolated": True,
        "extrapolation": "NOT one end-to-end frame: per-stage times of full-width single layers multiplied by the layer counts "
                         "(a full 7B fp32 frame needs ~31 GB of host weights and minutes of CPU time; the bounded sample is "
                         "what fits the default run). parts_s lists the per-stage totals.",
        "sample": ("CPU oracle (oracle/lisa_oracle.py, fp32 torch eager) on ONE 1024^2 frame with full-width, "
                   "reduced-depth stacks (SAM: patch+neck + 1 windowed + 1 global block; CLIP: 1-2 layers; Llama: 1 "
                   "layer prefill T=%d + 1 cached step + lm_head; both mask decoders + postprocess), each per-layer "
                   "time multiplied by the full layer count; KV-cached schedule" % T),
        "seconds_per_frame": t_frame,
        "reference_semantics_value": 1.0 / t_frame_ref,
        "reference_semantics_note": "no KV cache and CLIP re-run per generated token (LISA.py:115, llava_llama.py:82-90)",
        "parts_s": {"sam_encoder": t_sam, "clip": t_clip, "llm": t_llm, "decoders": t_dec2},
    }


def cpu_full_frame(cfg, text_tokens, n_gen, threads):
    """ONE real end-to-end frame of the full model through the CPU oracle (fp32, KV-cached schedule): 1024^2 frame, 36-id
    prompt (T = 291), n_gen forced answer tokens with one [SEG], every layer of every stack. Weights: fp32 tensors of the full
    inventory (~31 GB at 7B), one random template per distinct (shape, kind) copied into separate storage for every tensor
    (values only matter for timing; generating 7.7 G normals would take longer than the frame). Skipped when host memory is short."""
    from collections import OrderedDict
    from oracle import lisa_oracle as O
    try:
        import psutil
        need = sum(int(torch.tensor(sh).prod()) for sh in hw.all_shapes(cfg).values()) * 4 * 1.15
        if psutil.virtual_memory().available < need:
            return {"skipped": "host memory: %.0f GB available, %.0f GB needed" % (psutil.virtual_memory().available / 1e9, need / 1e9)}
    except ImportError:
        pass
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    templates, sd = {}, OrderedDict()
    for key, shape in hw.all_shapes(cfg).items():
        std = hw._std_for(key, shape)
        tk = (tuple(shape), std)
        if tk not in templates:
            z = torch.empty(shape, dtype=torch.float32).normal_()
            templates[tk] = z.mul_(0.1).add_(1.0) if std is None else z.mul_(float(std))
            sd[key] = templates[tk]
        else:
            sd[key] = templates[tk].clone()
    t_build = time.perf_counter() - t0
    g = torch.Generator().manual_seed(0)
    S = cfg.sam.img_size
    images = torch.randn((1, 3, S, S), generator=g)
    clip = torch.randn((1, 3, cfg.clip.image, cfg.clip.image), generator=g)
    hi = min(cfg.llm.vocab, cfg.seg_token_idx) - 1
    ids = torch.cat([torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]]),
                     torch.randint(3, hi, (1, text_tokens), generator=g)], 1)
    forced = torch.randint(3, hi, (1, n_gen), generator=g)
    forced[:, 2], forced[:, -1] = cfg.seg_token_idx, cfg.eos_token_id
    with torch.no_grad():
        t0 = time.perf_counter()
        out = O.lisa_evaluate(sd, cfg, clip, images, ids, [(S, S)], [(S, S)], max_new_tokens=n_gen, forced_answer=forced, use_cache=True)
        t_frame = time.perf_counter() - t0
    ok = out[1][0].shape == (1, S, S) and bool(torch.isfinite(out[1][0]).all())
    return {"seconds": t_frame, "frames_per_s": 1.0 / t_frame, "weights_build_seconds": t_build, "outputs_finite": ok,
            "what": "oracle.lisa_evaluate(use_cache=True) on one %dx%d frame, %d-id prompt, %d forced tokens, full depth (%d ViT-H blocks, "
                    "%d CLIP layers, %d Llama layers), fp32, %d threads" % (S, S, ids.shape[1], n_gen, cfg.sam.depth, cfg.clip.layers,
                                                                           cfg.llm.layers, threads)}


def _streams_fused(m):
    """bf16 mode, both residual streams fp32 (ViT-H: the fused form of round 6 — proj / lin2 write the fp32 stream and its bf16 copy)."""
    m.sam_encoder.fp32_stream = m.llm.fp32_stream = True


def _streams_fused_neck(m):
    _streams_fused(m)
    m.sam_encoder.neck_f32 = True


FULL_FRAME_VARIANTS = {"bf16_fp32_stream": _streams_fused, "bf16_fp32_stream_f32neck": _streams_fused_neck}


def parity_full_frame(cfg, device, threads, text_tokens=32, n_gen=8, modes=("bf16", "fp32"), attribution=False, seed=1234,
                      variants=None, field="gaussian"):
    """The 'mask IoU vs ref' half of the metric AT THE HEADLINE GEOMETRY: ONE full-depth frame (BASELINE.json configs[1]: every
    ViT-H block, CLIP layer and Llama layer; 1024^2 uint8 frame, 36-id prompt, n_gen forced answer tokens with one [SEG]) through
    LisaMI355.evaluate in each numeric mode and through the CPU oracle ON THE SAME WEIGHTS — the set is generated once in HBM
    (bf16 values, what the timed mode stores), handed to the HIP models as it is (fp32 mode: the same values widened) and copied
    to the host as fp32 for the oracle (~31 GB at 7B). The oracle's inputs come from the reference's host recipe
    (inference.preprocess for SAM, transformers' CLIPImageProcessor for CLIP: inference.py:229-256), the HIP path starts from the
    uint8 frame like the timed step does. The oracle's exact pass is timed: it IS cpu_baseline's one real frame (no second CPU
    frame in the run). attribution=True adds the oracle's bf16-points mode for one stack at a time and for all three (one stage
    re-run per variant: oracle.lisa_evaluate(points=, memo=)). Returns {"parity": {...}, "cpu_frame": {...}} or {"skipped": why}.
    variants: optional {name: callable(model)} applied to a bf16 model before its run (A/B of numeric options; default
    FULL_FRAME_VARIANTS). field: "gaussian" = the bench's noise frame on the weights as generated (a random decoder's logit field:
    Gaussian around 0); "two_plateau" = a two-region frame and the hypernetwork biases of both decoders re-aimed so that the
    oracle's logit field is +-10 on the regions with a thin boundary, as a trained checkpoint's is (tools/parity_bimodal.py's
    construction at full size; the same patched weights go to the oracle and to every HIP model)."""
    from collections import OrderedDict
    import numpy as np
    from oracle import lisa_oracle as O
    V = "model.visual_model"
    shapes = hw.all_shapes(cfg)
    n_par = sum(int(np.prod(sh)) for sh in shapes.values())
    try:
        import psutil
        need = n_par * 4 * 1.25 + 8e9
        if psutil.virtual_memory().available < need:
            return {"skipped": "host memory: %.0f GB available, %.0f GB needed for the oracle's fp32 copy of the weights" %
                               (psutil.virtual_memory().available / 1e9, need / 1e9)}
    except ImportError:
        pass
    try:
        from transformers import CLIPImageProcessor
    except ImportError:
        return {"skipped": "transformers (CLIPImageProcessor: the oracle's CLIP input) is not importable"}
    torch.set_num_threads(threads)
    S = cfg.sam.img_size
    sizes = [(S, S)]
    sd_dev = hw.make_state_dict_device(cfg, seed, device, torch.bfloat16)
    frames, _, ids, forced = make_inputs(cfg, 1, text_tokens, n_gen, device, seed=seed)
    region = None
    if field == "two_plateau":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import parity_bimodal as PB
        fr, region = PB.two_region_frame_u8(S, seed)
        frames = fr.to(device)
    frame_np = frames[0].cpu().numpy()
    images = O.sam_preprocess(frame_np, S)[None]
    clip = CLIPImageProcessor().preprocess(frame_np, return_tensors="pt")["pixel_values"].float()

    # the oracle first (round 6): its exact pass is the reference of every HIP run below, and in the two-plateau case it is what
    # the weights are patched from
    t0 = time.perf_counter()
    sd = OrderedDict((k, v.cpu().float()) for k, v in sd_dev.items())
    t_copy = time.perf_counter() - t0
    memo, taps = {}, {}
    kw = dict(max_new_tokens=n_gen, forced_answer=forced.cpu(), use_cache=True, memo=memo)
    field_diag = None
    with torch.no_grad():
        t0 = time.perf_counter()
        r_ids, r_left, r_right, r_tax = O.lisa_evaluate(sd, cfg, clip, images, ids.cpu(), sizes, sizes, taps=taps, **kw)
        t_frame = time.perf_counter() - t0
        if field == "two_plateau":
            g2 = (cfg.sam.grid,) * 2
            pe = O.sam_dense_pe(sd, V + ".prompt_encoder", g2)
            sp, de = O.sam_prompt_encoder_text(sd, V + ".prompt_encoder", taps["pred_embeddings"][0].unsqueeze(1), g2)
            field_diag = {}
            for side, tax_on in (("left", True), ("right", False)):
                key, new_bias, d = PB.plateau_bias(O, sd, f"{V}.mask_decoder_{side}", taps["image_embeddings"], pe, sp, de, tax_on, region)
                sd[key] = new_bias
                sd_dev[key] = new_bias.to(device, torch.bfloat16)
                field_diag[side] = d
            outs = {}
            for side, tax_on in (("left", True), ("right", False)):
                lo = O.sam_mask_decoder(sd, f"{V}.mask_decoder_{side}", taps["image_embeddings"], pe, sp, de, tax_on)
                outs[side] = O.sam_postprocess_masks(lo[0], S, sizes[0], sizes[0])[:, 0]
                if tax_on:
                    r_tax = [lo[2]]
            r_left, r_right = [outs["left"]], [outs["right"]]
        ref = {"left": r_left[0], "right": r_right[0]}
        oracle_variants = {}
        if attribution and field == "gaussian":
            t0 = time.perf_counter()
            for tag, pts in (("sam", ("sam",)), ("llama", ("llama",)), ("clip", ("clip",)), ("all", ("sam", "clip", "llama"))):
                _, vl, vr, _ = O.lisa_evaluate(sd, cfg, clip, images, ids.cpu(), sizes, sizes, points=pts, **kw)
                oracle_variants[tag] = {"left": vl[0], "right": vr[0]}
            t_attr = time.perf_counter() - t0
        else:
            attribution = False
    del sd, memo

    if variants is None:
        variants = FULL_FRAME_VARIANTS
    runs = [(m, None) for m in modes] + [(k, f) for k, f in variants.items()]
    hip = {}
    for name, tweak in runs:
        dt = torch.float32 if name == "fp32" else torch.bfloat16
        sd_m = sd_dev if dt == torch.bfloat16 else OrderedDict((k, v.float()) for k, v in sd_dev.items())
        model = LisaMI355(cfg, sd_m, dtype=dt, device=device, sam_chunk=1)
        del sd_m
        if tweak is not None:
            tweak(model)
        t0 = time.perf_counter()
        o_ids, left, right, tax = model.evaluate(None, None, ids, sizes, sizes, max_new_tokens=n_gen, forced_answer=forced,
                                                 frames_u8=frames)
        torch.cuda.synchronize()
        first_call_s = time.perf_counter() - t0
        # the stage outputs behind the masks (same kernels, same inputs: evaluate() is deterministic), for the per-stage distances
        from haff.preprocess import SAM_MEAN, SAM_STD
        ing = model.frame_ingest()
        o2, hidden = model.generate(ing.clip_pixels(frames, cfg.clip.image, model.dtype), ids, n_gen, forced)
        pred = model.seg_embeddings(o2, hidden)[0]
        emb = model.get_visual_embs_u8(ing.sam_frames(frames, S)[0], SAM_MEAN, SAM_STD)
        g = cfg.sam.grid
        hip[name] = {"ids": o_ids.cpu(), "left": left[0].float().cpu(), "right": right[0].float().cpu(), "tax": tax[0].float().cpu(),
                     "hidden": hidden.float().cpu(), "pred": pred.float().cpu(),
                     "emb": emb.float().view(1, g, g, -1).permute(0, 3, 1, 2).contiguous().cpu(), "first_call_s": first_call_s}
        del model, hidden, pred, emb, left, right
        torch.cuda.empty_cache()
    del sd_dev
    torch.cuda.empty_cache()

    def rel(a, b):
        return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

    def rms_rel(a, b):
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30)).item()

    def mask_stats(got, want):
        out = {}
        for hand in ("left", "right"):
            gm, r = got[hand], want[hand]
            a, b = gm > 0, r > 0
            inter, union = (a & b).sum().item(), (a | b).sum().item()
            dis = a != b
            err = (gm - r).abs()
            st = {"mask_iou": inter / union if union else 1.0, "logit_max_abs_err": err.max().item(),
                  "logit_max_rel_err": err.max().item() / r.abs().max().item(), "logit_rms_rel_err": rms_rel(gm, r),
                  "pixels_disagreeing": int(dis.sum()), "pixels": dis.numel(),
                  "max_abs_oracle_logit_at_disagreeing_pixels": r[dis].abs().max().item() if dis.any() else 0.0}
            for band in (1e-3,):   # north_star: logits within 1e-3, binary masks bit-exact: exact outside the 1e-3 band around 0
                keep = r.abs() >= band
                st["masks_equal_outside_abs_band_%g" % band] = bool(((a == b) | ~keep).all())
            for tau in (0.01,):
                keep = r.abs() >= tau * r.abs().max()
                u = ((a | b) & keep).sum().item()
                st["mask_iou_outside_%g_of_scale_band" % tau] = ((a & b) & keep).sum().item() / u if u else 1.0
            out[hand] = st
        out["mask_iou_min"] = min(out["left"]["mask_iou"], out["right"]["mask_iou"])
        out["logit_max_rel_err"] = max(out["left"]["logit_max_rel_err"], out["right"]["logit_max_rel_err"])
        out["logit_max_abs_err"] = max(out["left"]["logit_max_abs_err"], out["right"]["logit_max_abs_err"])
        return out

    seg_rows = O.seg_token_mask(r_ids, cfg.seg_token_idx)
    field_name = field
    field_stats = {h: {"std": ref[h].std().item(), "abs_max": ref[h].abs().max().item(), "positive_frac": (ref[h] > 0).float().mean().item(),
                       "frac_within_5pct_of_scale_of_threshold": (ref[h].abs() < 0.05 * ref[h].abs().max()).float().mean().item()}
                   for h in ("left", "right")}
    par = {"what": "ONE full-depth frame: %s, %dx%d uint8 frame, %d-id prompt (T = %d), %d forced answer tokens with one [SEG]; HIP "
                   "path from the uint8 frame (device ingest), oracle from inference.preprocess + CLIPImageProcessor; ONE weight set "
                   "(device generator, seed %d, bf16 values; fp32 mode and the oracle see the same values widened)" %
                   (cfg.name, S, S, ids.shape[1], ids.shape[1] + 255, n_gen, seed),
           "reference": "/root/reference/2Haff/model/LISA.py:432-534 (evaluate) restated in oracle/lisa_oracle.py",
           "oracle_logit_field": field_stats, "field": field_name}
    if field_diag is not None:
        par["two_plateau_construction"] = field_diag
    for name in hip:
        h = hip[name]
        st = mask_stats(h, ref)
        st["token_ids_equal"] = bool(torch.equal(h["ids"], r_ids))
        st["taxonomy_max_abs_err"] = (h["tax"] - r_tax[0]).abs().max().item()
        st["stage_rel_err"] = {"image_embedding_max": rel(h["emb"], taps["image_embeddings"]),
                               "image_embedding_rms": rms_rel(h["emb"], taps["image_embeddings"]),
                               "seg_hidden_state_max": rel(h["hidden"][seg_rows], taps["hidden"][seg_rows]),
                               "seg_hidden_state_rms": rms_rel(h["hidden"][seg_rows], taps["hidden"][seg_rows]),
                               "text_embedding_max": rel(h["pred"], taps["pred_embeddings"][0]),
                               "text_embedding_rms": rms_rel(h["pred"], taps["pred_embeddings"][0])}
        st["first_call_seconds"] = h["first_call_s"]
        if attribution and name != "fp32":
            st["vs_oracle_bf16_points"] = {k: v for k, v in mask_stats(h, oracle_variants["all"]).items() if not isinstance(v, dict)}
        par[name] = st
    if attribution:
        par["oracle_bf16_points_vs_exact"] = {
            tag: {k: v for k, v in mask_stats(oracle_variants[tag], ref).items() if not isinstance(v, dict)}
            for tag in ("sam", "llama", "clip", "all")}
        par["oracle_bf16_points_vs_exact"]["note"] = ("the oracle with fp32 arithmetic and bf16 roundings at the HIP path's kernel "
                                                      "boundaries, in ONE stack at a time / in all three (oracle.bf16_points): how "
                                                      "much of the bf16 mode's distance is storage rounding, and where")
        par["attribution_seconds"] = t_attr
    fails = []
    if "fp32" in hip:
        f = par["fp32"]
        ok = f["logit_max_abs_err"] <= 1e-3 and f["token_ids_equal"] and \
            all(f[h]["masks_equal_outside_abs_band_0.001"] for h in ("left", "right"))
        if not ok:
            fails.append("full_frame/fp32")
    par["gate"] = {"fp32": "mask logits within 1e-3 (absolute) of the oracle, binary masks equal wherever |oracle logit| >= 1e-3, ids equal",
                   "failed": fails}
    cpu = {"seconds": t_frame, "frames_per_s": 1.0 / t_frame, "weights_copy_seconds": t_copy, "outputs_finite": bool(torch.isfinite(ref["left"]).all()),
           "what": "oracle.lisa_evaluate(use_cache=True) on one %dx%d frame, %d-id prompt, %d forced tokens, full depth (%d ViT-H blocks, "
                   "%d CLIP layers, %d Llama layers), fp32, %d threads; the weights are the HIP model's (this frame is also the parity "
                   "reference: parity.full_frame)" % (S, S, ids.shape[1], n_gen, cfg.sam.depth, cfg.clip.layers, cfg.llm.layers, threads)}
    return {"parity": par, "cpu_frame": cpu}


LINE_BUDGET_BYTES = 4096   # the driver keeps a bounded tail of stdout: round 5's 20.7 KB line was not parsed (BENCH_r05.json)


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(full):
    """The ONE stdout line the driver parses: the contract's keys + roofline + cpu_baseline + the parity summary of the timed mode,
    every nested diagnostic (by_shape, the per-geometry parity objects, extrapolations ...) left to bench_detail.json. Pure function
    of the full line object (tests/test_bench_line_cpu.py holds it to LINE_BUDGET_BYTES)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    out = {k: full[k] for k in keep if k in full}     # the contract's numbers at full precision
    cfg = dict(full.get("config") or {})
    if isinstance(cfg.get("workload"), str) and len(cfg["workload"]) > 400:
        cfg["workload"] = cfg["workload"][:397] + "..."
    out["config"] = cfg
    rf = full.get("roofline")
    if rf:
        r_keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch_avg",
                  "launches_per_step", "avg_launch_us", "gemm_share_of_step")
        out["roofline"] = {k: _r(rf.get(k)) for k in r_keep if k in rf}
        if isinstance(out["roofline"].get("kernel"), str):
            out["roofline"]["kernel"] = out["roofline"]["kernel"][:120]
        if rf.get("whole_path"):
            out["roofline"]["whole_path_frac"] = _r(rf["whole_path"].get("frac"))
        ws = rf.get("weight_streaming_gemm")
        if ws and ws.get("achieved"):
            out["roofline"]["weight_streaming_GBps"] = _r(ws["achieved"], 1)
    else:
        out["roofline"] = None
    cb = full.get("cpu_baseline")
    if cb:
        c_keep = ("value", "unit", "cores", "kind", "seconds_per_frame", "extrapolated")
        out["cpu_baseline"] = {k: _r(cb.get(k), 6) for k in c_keep if k in cb}
        smp = cb.get("sample") or ""
        out["cpu_baseline"]["sample"] = smp if len(smp) <= 300 else smp[:297] + "..."
    else:
        out["cpu_baseline"] = None
    par = full.get("parity")
    if par:
        p = {"timed_mode": par.get("timed_mode"), "reference": "oracle/lisa_oracle.py (CPU fp32 restatement of LISA.py:432-534)"}
        ff = par.get("full_frame") or {}
        tm = ff.get(par.get("timed_mode") or "bf16") or {}
        if tm:
            p.update({"geometry": "one full-depth frame of the timed config", "mask_iou_min": _r(tm.get("mask_iou_min"), 5),
                      "mask_iou_left_right": [_r((tm.get(h) or {}).get("mask_iou"), 5) for h in ("left", "right")],
                      "logit_max_rel_err": _r(tm.get("logit_max_rel_err"), 6), "token_ids_equal": tm.get("token_ids_equal")})
        elif par.get("tiny"):
            tm = par["tiny"].get(par.get("timed_mode") or "bf16") or {}
            p.update({"geometry": "tiny (configs[0]); full-depth frame skipped", "mask_iou_min": _r(tm.get("mask_iou_vs_oracle"), 5),
                      "logit_max_rel_err": _r(tm.get("mask_logit_max_err_rel"), 6), "token_ids_equal": tm.get("token_ids_equal")})
        f32 = ff.get("fp32") or {}
        if f32:
            p["fp32_mask_iou_min"] = _r(f32.get("mask_iou_min"), 5)
            p["fp32_logit_max_abs_err"] = _r(f32.get("logit_max_abs_err"), 8)
        p["meets_iou_0.999"] = bool(p.get("mask_iou_min") is not None and p["mask_iou_min"] >= 0.999)
        p["gate_failed"] = (par.get("gate") or {}).get("failed")
        out["parity"] = p
    for k in ("rccl_ranks", "frames_per_s_per_gpu", "samples_per_s_per_gpu", "latency_batch1_ms", "outputs_finite", "detail",
              "host_enqueue_ms_per_step", "peak_hbm_gb"):
        if k in full:
            out[k] = _r(full[k])
    d1 = full.get("decode_step_batch1")
    if d1:
        out["decode_step_batch1"] = {"ms": _r(d1.get("ms")), "achieved_TBps": _r(d1.get("achieved_TBps")), "peak_TBps": d1.get("peak_TBps")}
    s = json.dumps(out)
    if len(s) > LINE_BUDGET_BYTES:     # never let a long free-text field push the line over: shorten those, keep every number
        for path in (("config", "workload"), ("cpu_baseline", "sample"), ("roofline", "kernel"), ("config", "parallelism")):
            o = out.get(path[0])
            if isinstance(o, dict) and isinstance(o.get(path[1]), str):
                o[path[1]] = o[path[1]][:80]
        s = json.dumps(out)
    # ... and never fail to print a line: drop the optional extras, then everything but the contract, should that still not fit
    for k in ("decode_step_batch1", "detail", "outputs_finite", "latency_batch1_ms", "host_enqueue_ms_per_step", "peak_hbm_gb"):
        if len(s) <= LINE_BUDGET_BYTES:
            break
        out.pop(k, None)
        s = json.dumps(out)
    if len(s) > LINE_BUDGET_BYTES:
        out["config"] = {"workload": str((full.get("config") or {}).get("workload", ""))[:80]}
        for k in ("roofline", "cpu_baseline", "parity"):
            if isinstance(out.get(k), dict):
                out[k] = {kk: vv for kk, vv in out[k].items() if not isinstance(vv, (str, list, dict))}
    return out


def emit(full):
    """Everything measured goes to bench_detail.json (under gpurun_out/ so it travels back from a GPU box; beside this file
    otherwise); stdout gets the compact line and NOTHING after it."""
    dst = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_detail.json")
            with open(path, "w") as fh:
                json.dump(full, fh, indent=1)
            dst = os.path.relpath(path, ROOT)
            break
        except (OSError, TypeError, ValueError):      # an unwritable directory or an unserialisable diagnostic must not cost the line
            continue
    full["detail"] = dst
    sys.stderr.flush()
    print(json.dumps(compact_line(full)), flush=True)


def base_line(fps, world, steps, warmup, ms_per_step, workload, B, extra_cfg):
    cfg = {"workload": workload, "frames_per_step_per_gpu": B,
           "parallelism": "frame-sharded replicas x%d (no collective)" % world}
    cfg.update(extra_cfg)
    return {"metric": "affordance frames/sec/GPU @1024^2, 32-tok prompt; mask IoU vs ref",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic", "config": cfg, "frames_per_s_per_gpu": fps / world}


def make_train_batch(cfg, b, n_ids, mask_hw, device, seed=0):
    """One synthetic 2HANDS micro-batch in collate_fn's layout (utils/dataset.py:152-169): b conversations of n_ids ids (n_ids + 255
    expanded tokens), one [SEG] each, left / right masks of mask_hw, soft taxonomy targets."""
    g = torch.Generator().manual_seed(seed)
    S = cfg.sam.img_size
    hi = min(cfg.llm.vocab, cfg.seg_token_idx) - 1
    ids = torch.randint(3, hi, (b, n_ids), generator=g)
    ids[:, 0], ids[:, 1], ids[:, 2], ids[:, 3] = cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx
    ids[:, n_ids - 3], ids[:, n_ids - 1] = cfg.seg_token_idx, cfg.eos_token_id
    labels = ids.clone()
    labels[:, :n_ids - 8] = -100
    return dict(images=torch.randn((b, 3, S, S), generator=g).to(device, torch.bfloat16),
                images_clip=torch.randn((b, 3, cfg.clip.image, cfg.clip.image), generator=g).to(device, torch.bfloat16),
                input_ids=ids.to(device), labels=labels.to(device), attention_masks=torch.ones_like(ids, dtype=torch.bool).to(device),
                offset=torch.arange(b + 1).to(device),
                masks_list_left=[(torch.rand((1,) + mask_hw, generator=g) > 0.5).float().to(device) for _ in range(b)],
                masks_list_right=[(torch.rand((1,) + mask_hw, generator=g) > 0.5).float().to(device) for _ in range(b)],
                label_list=[{"left": torch.zeros(mask_hw), "right": torch.zeros(mask_hw)} for _ in range(b)], resize_list=[(S, S)] * b,
                taxonomies_list=torch.eye(4)[torch.arange(b) % 4].to(device), inference=False)


def cpu_train_baseline(cfg, n_ids, mask_hw, threads):
    """cpu_baseline of the --mode train line: ONE fine-tune step (forward + backward of model_forward, LISA.py:175-430) of the CPU
    oracle under torch autograd on ONE sample at FULL WIDTH and REDUCED DEPTH (2 ViT-H blocks, 2 CLIP layers; 1 and 2 Llama
    layers), fp32, `threads` host threads; the full-depth step is extrapolated from the per-layer differences and labelled so.
    The reference's trainable set (LoRA r = 8 on q/v_proj + embed_tokens, lm_head, text_hidden_fcs, both mask decoders)."""
    import copy
    from oracle import lisa_oracle as O
    torch.set_num_threads(threads)
    V = "model.visual_model"
    g = torch.Generator().manual_seed(0)

    def one(llm_layers):
        small = copy.deepcopy(cfg)
        small.sam.depth, small.sam.global_idx = 2, (1,)
        small.clip.layers, small.clip.select_layer = 2, 2
        small.llm.layers = llm_layers
        sd = hw.make_state_dict(small, 99)
        lora = {}
        for i in range(llm_layers):
            for n in ("q_proj", "v_proj"):
                k = f"model.layers.{i}.self_attn.{n}"
                lora[k + ".lora_A"] = (torch.randn((8, small.llm.hidden), generator=g) * 0.02).requires_grad_(True)
                lora[k + ".lora_B"] = (torch.randn((small.llm.hidden, 8), generator=g) * 0.02).requires_grad_(True)
        for k in list(sd):
            if k in ("lm_head.weight", "model.embed_tokens.weight") or "text_hidden_fcs" in k or "mask_decoder_" in k:
                sd[k] = sd[k].requires_grad_(True)
        batch = make_train_batch(small, 1, n_ids, mask_hw, "cpu", seed=5)
        batch = {k: (v.float() if torch.is_tensor(v) and v.dtype == torch.bfloat16 else v) for k, v in batch.items()}
        t0 = time.perf_counter()
        out = O.lisa_model_forward(sd, small, batch, lora=lora, lora_alpha=16.0)
        t1 = time.perf_counter()
        out["loss"].backward()
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, float(out["loss"].detach())
    f1, b1, _ = one(1)
    f2, b2, loss = one(2)
    per_llm = max((f2 + b2) - (f1 + b1), 1e-6)
    # the frozen towers run forward only: per-block times of the encoder / CLIP from the inference baseline's recipe
    s = copy.deepcopy(cfg.sam)
    s.depth, s.global_idx = 2, (1,)
    sd = hw.make_state_dict(copy.deepcopy(cfg), 98, {k: v for k, v in hw.all_shapes(cfg).items() if ".image_encoder.blocks.0." in k or ".image_encoder.blocks.7." in k})
    x = torch.randn((1, s.grid, s.grid, s.embed_dim), generator=g)
    with torch.no_grad():
        t = time.perf_counter(); O.sam_block(sd, V + ".image_encoder.blocks.0", x, s.heads, s.window); t_win = time.perf_counter() - t
        t = time.perf_counter(); O.sam_block(sd, V + ".image_encoder.blocks.7", x, s.heads, 0); t_glob = time.perf_counter() - t
    n_glob = len(cfg.sam.global_idx)
    extra_sam = (cfg.sam.depth - n_glob - 1) * t_win + (n_glob - 1) * t_glob
    # the frozen CLIP tower too: the reduced model ran 2 of its layers (forward only), the full one runs select_layer's count
    c2 = copy.deepcopy(cfg.clip)
    c2.layers, c2.select_layer = 2, 2
    c1 = copy.deepcopy(c2)
    c1.select_layer = 1
    sdc = hw.make_state_dict(copy.deepcopy(cfg), 97, {k: v for k, v in hw.clip_shapes(c2).items()})
    ic = torch.randn((1, 3, cfg.clip.image, cfg.clip.image), generator=g)
    with torch.no_grad():
        t = time.perf_counter(); O.clip_vision_features(sdc, "model.vision_tower.vision_tower", ic, c2); t_c2 = time.perf_counter() - t
        t = time.perf_counter(); O.clip_vision_features(sdc, "model.vision_tower.vision_tower", ic, c1); t_c1 = time.perf_counter() - t
    t_clip_layer = max(t_c2 - t_c1, 0.0)
    n_clip = cfg.clip.layers + 1 + cfg.clip.select_layer if cfg.clip.select_layer < 0 else cfg.clip.select_layer
    extra_clip = max(n_clip - 2, 0) * t_clip_layer
    t_full = (f1 + b1) + (cfg.llm.layers - 1) * per_llm + extra_sam + extra_clip
    return {"value": 1.0 / t_full, "unit": "samples/s", "cores": threads, "kind": "port", "extrapolated": True,
            "sample": ("CPU oracle (oracle/lisa_oracle.py under torch autograd, fp32) — ONE sample, %d-id conversation, %dx%d masks, full "
                       "width, reduced depth: measured forward + backward with 1 and 2 Llama layers (2 ViT-H blocks, 2 CLIP layers), "
                       "full depth = that + (layers - 1) x the difference + the remaining frozen ViT-H blocks and CLIP layers forward" %
                       (n_ids, mask_hw[0], mask_hw[1])),
            "measured_s": {"fwd_1_layer": f1, "bwd_1_layer": b1, "fwd_2_layers": f2, "bwd_2_layers": b2, "vit_h_window_block_fwd": t_win,
                           "vit_h_global_block_fwd": t_glob, "clip_layer_fwd": t_clip_layer},
            "seconds_per_sample_full_depth_extrapolated": t_full, "loss_of_the_reduced_model": loss}


def train_main(args):
    """--mode train: BASELINE.json configs[3] — LoRA fine-tune (train_ds.py path), bf16, 8 synthetic 2HANDS samples per GPU per
    step (global batch 64 on 8 GPUs), 96-id conversations (351 expanded tokens), 1024^2 masks. A step = forward + backward of
    one micro-batch, the all-reduce of the trainable set's gradients (GradBucketReducer: RCCL when N > 1, launched from backward
    hooks) and the fused clip + AdamW update — nothing skipped. value = samples/s over all ranks."""
    from haff import train_ops as T
    from haff.train_model import LisaTrainable
    if args.materialised_attention:
        from haff import autograd as hag
        hag.FLASH_TRAINING_ATTENTION = False
    if args.separate_lora:
        from haff import autograd as hag
        hag.FUSED_LORA_QKV = False
    rank, world, local_rank = hdist.init_from_env("nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cfg = {"7b": hcfg.haff_7b, "13b": hcfg.haff_13b, "tiny": hcfg.tiny, "mid": hcfg.mid}[args.config]()
    sd = hw.make_state_dict_device(cfg, 1234, device, torch.bfloat16)
    model = LisaTrainable(cfg, sd, dtype=torch.bfloat16, device=device)
    model.overlap_sam = not args.single_stream   # --single-stream: the frozen SAM encoder in front of the Llama forward instead of beside it
    model.independent_lora_dropout = not args.shared_lora_dropout
    del sd
    torch.cuda.empty_cache()
    b = args.batch if args.batch != 64 else 8
    batch = make_train_batch(cfg, b, args.train_ids, (args.train_mask, args.train_mask), device, seed=1234 + rank)
    named = list(model.named_parameters())
    reducer = T.GradBucketReducer(named)
    opt = T.BucketAdamW(reducer, named)   # one fused AdamW launch per gradient bucket
    losses = []

    def step():
        reducer.zero()
        reducer.begin(sync=True)
        out = model(**batch)
        out["loss"].backward()
        reducer.finish()
        # clip_grad_norm's coefficient (train_ds.py:381) stays on the device: the optimizer launches queue up behind backward
        # instead of waiting for a host read of the norm (the timed region's host reads are the step-start copies of the ids)
        clip = T.clip_coef_device(T.grad_norm(reducer.grads()), 1.0)
        opt.step(lr=3e-4, gscale=1.0, gscale_dev=clip)
        losses.append(out["loss"].detach())
    for _ in range(args.warmup):
        step()
    elapsed = hdist.timed_steps(step, args.steps, device)
    ms_per_step = 1e3 * elapsed / args.steps
    sps = world * b * args.steps / elapsed
    n_ranks = rccl_ranks(device)
    # how long the host needs to ENQUEUE one step on an idle GPU (no read-back waits on device work then): the step is GPU-bound
    # while this stays below ms_per_step
    torch.cuda.synchronize()
    t_host = time.time()
    step()
    host_enqueue_ms = 1e3 * (time.time() - t_host)
    torch.cuda.synchronize()
    if rank == 0:
        # per-launch event timing needs ONE HIP stream: with the frozen SAM encoder on its side stream the main stream's products
        # were timed while they waited for CUs the encoder's persistent tiles held (round 4's by_shape listed the CLIP products at
        # M = 2056 at 52 ... 211 TFLOP/s for that reason: single-stream they run at what the inference line shows)
        overlap_prev, model.overlap_sam = model.overlap_sam, False
        step()
        torch.cuda.synchronize()
        with GemmMeter() as meter:
            step()
        torch.cuda.synchronize()
        model.overlap_sam = overlap_prev
        n_launch, gemm_ms, gemm_fl, gemm_bytes = meter.summary()
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12
        n_train = sum(p.numel() for _, p in named)
        line = {"metric": "LoRA fine-tune samples/sec (train_ds.py path), bf16", "value": sps, "unit": "samples/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "BASELINE.json configs[3]: %s LoRA fine-tune (r=8 on q/v_proj + embed_tokens, lm_head, text_hidden_fcs, "
                                       "both mask decoders: %.0f M trainable), %d samples/step/GPU, %d-id conversations (%d expanded tokens), "
                                       "%dx%d masks, forward + backward + gradient all-reduce + AdamW" %
                                       (cfg.name, n_train / 1e6, b, args.train_ids, args.train_ids + 255, args.train_mask, args.train_mask),
                           "samples_per_step_per_gpu": b,
                           "parallelism": "sample-sharded DDP x%d (all-reduce of %.2f GB of bf16/fp32 gradients per step in %d buckets)" %
                                          (world, sum(f.numel() * f.element_size() for f in reducer.grads()) / 1e9, len(reducer.buckets))},
                "samples_per_s_per_gpu": sps / world, "rccl_ranks": n_ranks,
                "host_enqueue_ms_per_step": host_enqueue_ms,
                "loss_first_last": [float(losses[0]), float(losses[-1])],
                "roofline": {"bound": "mfma", "kernel": "gemm_bf16_kernel (every haff_gemm_bf16 launch with M > 64 of the step: forward, dX and dW products)",
                             "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                             "traffic": None, "launches_per_step": n_launch, "avg_launch_us": 1e3 * gemm_ms / max(n_launch, 1),
                             "flops_per_launch_avg": gemm_fl / max(n_launch, 1), "algorithmic_bytes_per_launch_avg": gemm_bytes / max(n_launch, 1),
                             "gemm_share_of_step": gemm_ms / ms_per_step,
                             "executed_gemm_flop_per_sample": meter.total_gemm_flop() / b,
                             "by_shape": meter.shape_summary(top=24)},
                "cpu_baseline": None,
                "peak_hbm_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_train_baseline(cfg, args.train_ids, (args.train_mask, args.train_mask), min(len(os.sched_getaffinity(0)), 32))
        emit(line)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


def stub_main(args):
    """The N-rank control path of this file with the model replaced by a sleep (CPU, gloo): same rendezvous, same
    fence | K steps | fence | max-over-ranks, same single JSON line from rank 0, same teardown."""
    rank, world, _ = hdist.init_from_env("gloo")
    B = args.batch

    def step():
        time.sleep(args.stub_step_ms * 1e-3)
    for _ in range(args.warmup):
        step()
    elapsed = hdist.timed_steps(step, args.steps, None)
    n_ranks = rccl_ranks(None)
    if rank == 0:
        line = base_line(world * B * args.steps / elapsed, world, args.steps, args.warmup, 1e3 * elapsed / args.steps,
                         "stub (sleep %.0f ms per step)" % args.stub_step_ms, B, {"stub": True})
        line["cpu_baseline"] = None
        line["rccl_ranks"] = n_ranks
        emit(line)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def self_launch(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves — the reference launches with
    `deepspeed --master_port=24999 train_ds.py` (2Haff/README.md:69), i.e. one command. The parent has touched no GPU at this point
    (no HIP call, no torch.cuda.is_available(): a process that initialised the GPU must not be the one that forks / execs rank
    processes); it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py <same
    flags>` as a CHILD, forwards rank 0's JSON line (the children inherit stdout) and exits with the child's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // args.gpus)))
    print("bench.py: --gpus %d without WORLD_SIZE: launching %d ranks (%s)" % (args.gpus, args.gpus, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def rccl_ranks(device):
    """Ranks that took part in one real all-reduce on the benchmark's process group (RCCL on the GPU node): every rank adds 1."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return 1
    one = torch.ones((1,), dtype=torch.float32, device=device if device is not None else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    n = int(round(float(one.item())))
    assert n == dist.get_world_size(), (n, dist.get_world_size())
    return n


def main(argv=None):
    exit_code = 0
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="7b", choices=["7b", "13b", "tiny", "mid"])
    ap.add_argument("--batch", type=int, default=64, help="frames per step per GPU")
    ap.add_argument("--text-tokens", type=int, default=32)
    ap.add_argument("--n-gen", type=int, default=8)
    ap.add_argument("--sam-chunk", default="auto",
                    help="frames per pass of the SAM encoder: a number, or 'auto' = overlap.auto_chunk (64 frames: 16)")
    ap.add_argument("--sam-caps", default="auto",
                    help="workgroups per persistent GEMM launch for each encoder chunk (haff_gemm_stream_cap), e.g. 256,256,224,224; "
                         "'auto' = LisaMI355's rule, 'off' = one per CU everywhere (A/B)")
    ap.add_argument("--sam-waits-for-prefill", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--single-stream", action="store_true",
                    help="serialise the SAM encoder and the language model on one HIP stream (default: two streams)")
    ap.add_argument("--fold-norms", action="store_true", help="(default since round 3 for the ViT-H geometry; kept for old command lines)")
    ap.add_argument("--no-fused-qkv-rope", action="store_true", help="Llama prefill: q|k|v product + haff_rope_cache instead of RoPE / cache append in the product's epilogue (A/B)")
    ap.add_argument("--no-producer-stats", action="store_true", help="folded norms: row statistics by haff_row_stats instead of the producing GEMM's epilogue (A/B)")
    ap.add_argument("--no-fold-norms", action="store_true", help="SAM blocks: LayerNorm kernels instead of the norm carried into the qkv / lin1 products (A/B)")
    ap.add_argument("--tables-global", action="store_true",
                    help="SAM global blocks: rel-pos as fp32 tables + the plain attention kernel instead of the fused kernel (A/B)")
    ap.add_argument("--token-major-windows", action="store_true",
                    help="SAM windowed blocks: the token-major q|k|v buffer of rounds 1-4 instead of head-major planes (A/B)")
    ap.add_argument("--sam-beside-decode", default="auto", choices=["auto", "on", "off"],
                    help="where the SAM encoder is enqueued on its stream: on = behind the prefill (beside the HBM-bound decode steps), "
                         "off = first (beside CLIP + prefill), auto = by batch size (lisa.py)")
    ap.add_argument("--no-prune-last-layer", action="store_true",
                    help="Llama prefill: the last layer's o_proj / MLP / norm on every row instead of the rows evaluate() reads (A/B)")
    ap.add_argument("--decode-chain", default="auto", choices=["auto", "on"], help="on: the chained decode launch even beside a capped encoder (A/B)")
    ap.add_argument("--no-decode-chain", action="store_true",
                    help="decode steps of <= 8 rows: five launches per Llama layer instead of the one chained launch per step (csrc/decode_chain.hip; A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-full-frame", action="store_true", help="cpu_baseline: skip the one real end-to-end CPU frame (~31 GB host RAM, ~1 min)")
    ap.add_argument("--no-parity", action="store_true", help="skip the tiny-config HIP-vs-oracle parity object")
    ap.add_argument("--no-b1", action="store_true", help="skip the batch=1 latency line (configs[1])")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"],
                    help="numeric mode of the timed model: bf16 (bf16 MFMA stacks + fp32 decoder tail: the headline) or f32 (the "
                         "parity mode, f32-input MFMA everywhere: what IoU >= 0.999 / logits within 1e-3 is measured in)")
    ap.add_argument("--fp32-stream", default="off", choices=["off", "sam", "llm", "both"],
                    help="bf16 mode with the ViT-H and / or Llama residual stream kept in fp32 between the bf16 MFMA products "
                         "(LisaMI355(fp32_stream=...): 2-3x closer to the reference at depth 32; DESIGN.md section 2)")
    ap.add_argument("--neck-f32", action="store_true", help="bf16 mode: the ViT-H neck on the f32-input MFMA path (LisaMI355(neck_f32=True))")
    ap.add_argument("--unfused-fp32-stream", action="store_true",
                    help="--fp32-stream sam/both: round 5's unfused form (LayerNorm kernels on the fp32 stream) instead of the fused epilogue (A/B)")
    ap.add_argument("--no-full-frame-parity", action="store_true",
                    help="parity: skip the one full-depth frame against the oracle on shared weights (~31 GB host RAM, ~1 min of CPU)")
    ap.add_argument("--attribution", action="store_true",
                    help="full-frame parity: add the oracle's per-stack bf16-points passes (~45 s of CPU; off by default since round 6: "
                         "the default run is the driver's bench, tools/full_frame_parity.py is where attribution lives)")
    ap.add_argument("--no-attribution", action="store_true", help="(default since round 6; kept for old command lines)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer: BASELINE configs[2] (the headline metric); train: configs[3], one LoRA fine-tune step per step")
    ap.add_argument("--materialised-attention", action="store_true",
                    help="--mode train: the Llama self-attention with probabilities in HBM (batched products + softmax kernels) instead of the flash pair (A/B)")
    ap.add_argument("--separate-lora", action="store_true",
                    help="--mode train: the q / v adapters as separate product / scale / add / RoPE nodes instead of the fused node of csrc/lora.hip (A/B)")
    ap.add_argument("--shared-lora-dropout", action="store_true",
                    help="--mode train: ONE dropout mask per layer for the q and the v adapter (rounds 3-4) instead of peft's two (A/B)")
    ap.add_argument("--train-ids", type=int, default=96)
    ap.add_argument("--train-mask", type=int, default=1024)
    ap.add_argument("--stub-step-ms", type=float, default=None,
                    help="TEST ONLY (tests/test_dist_gloo.py): replace the model by a sleep of this many ms and rendezvous "
                         "over gloo on the CPU, to exercise the multi-rank fence / timing / reporting path without GPUs")
    args = ap.parse_args(argv)

    # --gpus N must mean N ranks: under torchrun WORLD_SIZE says so; bare, this process becomes the launcher. Never an n_gpus: 1
    # line for --gpus 8.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        return self_launch(args, argv)
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher's WORLD_SIZE is %s: refusing to report a line for a different rank "
                         "count than the one asked for" % (args.gpus, env_world))
    if args.stub_step_ms is not None:
        return stub_main(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.mode == "train":
        return train_main(args)
    rank, world, local_rank = hdist.init_from_env("nccl")  # "nccl" IS RCCL on ROCm
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cfg = {"7b": hcfg.haff_7b, "13b": hcfg.haff_13b, "tiny": hcfg.tiny, "mid": hcfg.mid}[args.config]()
    run_dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    sd = hw.make_state_dict_device(cfg, 1234, device, torch.bfloat16)   # bf16 VALUES in either mode (f32: widened below)
    if run_dtype == torch.float32:
        for k in list(sd.keys()):
            sd[k] = sd[k].float()
    if args.fold_norms:
        cfg.sam.fold_norms = True
    model = LisaMI355(cfg, sd, dtype=run_dtype, device=device, sam_chunk="auto" if args.sam_chunk == "auto" else int(args.sam_chunk),
                      fp32_stream=False if args.fp32_stream == "off" else args.fp32_stream, neck_f32=args.neck_f32)
    model.sam_encoder.fused_fp32_stream = not args.unfused_fp32_stream
    if args.fold_norms:
        model.sam_encoder.fold_norms = True
    model.overlap_streams = not args.single_stream
    if args.sam_waits_for_prefill != "auto":
        model.sam_waits_for_prefill = args.sam_waits_for_prefill == "on"
    model.sam_chunk_caps = {"auto": "auto", "off": None}.get(args.sam_caps) if args.sam_caps in ("auto", "off") else \
        [int(c) for c in args.sam_caps.split(",")]
    if args.no_prune_last_layer:
        model.prune_last_layer = False
    if args.no_decode_chain:
        model.decode_chain = False
    elif args.decode_chain == "on":
        model.decode_chain = True
    if args.tables_global:
        model.sam_encoder.fused_global = False
    if args.no_fold_norms:
        model.sam_encoder.fold_norms = False
    if args.no_producer_stats:
        model.sam_encoder.producer_stats = False
    if args.no_fused_qkv_rope:
        model.llm.fused_qkv_rope = False
    if args.token_major_windows:
        model.sam_encoder.head_major_windows = False
    if args.sam_beside_decode != "auto":
        model.sam_beside_decode = args.sam_beside_decode == "on"
    del sd
    torch.cuda.empty_cache()
    B, S = args.batch, cfg.sam.img_size
    frames, images_clip, ids, forced = make_inputs(cfg, B, args.text_tokens, args.n_gen, device, seed=1234 + rank)
    sizes = [(S, S)] * B

    def step(n=B):
        # the whole per-frame path from the uint8 frame: CLIP preprocessing (a2) and the SAM ingest (a1) run on the device
        # inside the timed region; the only inputs are the frames and the prompt ids
        return model.evaluate(None, None, ids[:n], sizes[:n], sizes[:n], max_new_tokens=args.n_gen,
                              forced_answer=forced[:n], frames_u8=frames[:n])

    for _ in range(args.warmup):
        step()
    outs = []
    elapsed = hdist.timed_steps(lambda: outs.append(step()), args.steps, device)  # fence | K steps | fence | max over ranks
    out = outs[-1]
    chain_timed = bool(getattr(model, "last_decode_chain", False))
    plan_timed = model.last_plan     # (workgroup caps per encoder chunk, encoder waits for the prefill, frames per chunk) of the timed steps
    ms_per_step = 1e3 * elapsed / args.steps
    fps = world * B * args.steps / elapsed
    n_ranks = rccl_ranks(device)   # one real all-reduce over the group the timing fence used: the rank count that actually ran

    if rank == 0:
        flops_frame = SURVEY_FLOPS.get(cfg.name) or hflops.frame_flops(cfg, args.text_tokens, args.n_gen)["total"]
        prev = (model.overlap_streams, model.decode_graphs)
        model.overlap_streams = False  # per-launch event timing needs a single HIP stream
        model.decode_graphs = False    # ... and every GEMM launch to go through the metered wrapper (no graph replays)
        step()                         # one untimed step in this (eager, single-stream) mode: its first pass allocates the eager
        torch.cuda.synchronize()       # decode loop's buffers / workspaces inside what would be metered launches (seen once: frac 0.39)
        with GemmMeter() as meter:
            step()
        model.overlap_streams, model.decode_graphs = prev
        n_launch, gemm_ms, gemm_fl, gemm_bytes = meter.summary()
        ws_n, ws_ms, ws_bytes = meter.stream_summary()
        f32_n, f32_ms, f32_fl = meter.f32_summary()
        # FLOPs the step actually executed: every GEMM launch as issued (the padded window rows SURVEY's 10.01 TFLOP counts
        # are skipped by the kernels) + the attention / rel-pos terms of flops.py
        parts = hflops.frame_flops(cfg, args.text_tokens, args.n_gen)
        s_ = cfg.sam
        attn_fl = 0.0
        for i in range(s_.depth):
            glob = i in s_.global_idx
            ntok, n_seq = (s_.grid ** 2, 1) if glob else (s_.window ** 2, ((s_.grid + s_.window - 1) // s_.window) ** 2)
            attn_fl += 2.0 * 2 * n_seq * ntok * ntok * s_.embed_dim + 2.0 * n_seq * ntok * 2 * (s_.grid if glob else s_.window) * s_.embed_dim
        n_clip = cfg.clip.n_patches + 1
        n_clip_layers = cfg.clip.layers + 1 + cfg.clip.select_layer if cfg.clip.select_layer < 0 else cfg.clip.select_layer
        attn_fl += n_clip_layers * 2.0 * 2 * n_clip * n_clip * cfg.clip.hidden
        T_all = 4 + args.text_tokens + cfg.clip.n_patches - 1 + args.n_gen - 1
        attn_fl += 2.0 * cfg.llm.layers * cfg.llm.hidden * sum(t + 1 for t in range(T_all))
        executed_frame = meter.total_gemm_flop() / B + attn_fl
        # HBM traffic of the dominant kernel cannot be sampled from inside this process: it comes from the two
        # rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of THIS command, summarised by tools/pmc_traffic.py into
        # profiles/ (FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes). null when no summary matches.
        traffic, traffic_src = None, None
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_gemm_traffic.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            from haff import lib as hlib
            if pmc.get("config") == cfg.name and pmc.get("batch") == B and pmc.get("library_source_sha16") == hlib.source_hash():
                traffic = pmc["hbm_bytes_per_launch"]
                traffic_src = "profiles/pmc_gemm_traffic.json (%s)" % pmc.get("collected", "")
            else:   # collected on another tree / workload: not this run's traffic
                traffic_src = "profiles/pmc_gemm_traffic.json does not match this tree (config %s, batch %s, csrc sha %s vs %s): null" % (
                    pmc.get("config"), pmc.get("batch"), pmc.get("library_source_sha16"), hlib.source_hash())
        f32_mode = run_dtype == torch.float32
        if f32_mode:   # the parity mode: every product is gemm_f32_kernel (f32-input MFMA, 157.3 TFLOP/s dense)
            n_launch, gemm_ms, gemm_fl = f32_n, f32_ms, f32_fl
            gemm_bytes, traffic, traffic_src = 0.0, None, None
        peak = 157.3 if f32_mode else PEAK_BF16_TFLOPS
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12
        roofline = {
            "bound": "mfma", "kernel": "gemm_f32_kernel (haff_gemm_f32: v_mfma_f32_16x16x4_f32 tiles; every product of the fp32 parity mode)" if f32_mode else
                                       "gemm_bf16_kernel (haff_gemm_bf16 with M > 64: the 256x256 / 128x128 MFMA tiles, all epilogue variants)",
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch_avg": gemm_bytes / n_launch,
            "launches_per_step": n_launch, "avg_launch_us": 1e3 * gemm_ms / n_launch,
            "flops_per_launch_avg": gemm_fl / n_launch, "gemm_share_of_step": gemm_ms / ms_per_step,
            # the same launches split by what their epilogue carries besides bias / activation / residual (round 3 moved the
            # LayerNorm fold, the LayerNorm statistics of the output rows, RoPE + KV-cache append INTO these launches: the work
            # left other kernels, the FLOPs did not change, so `frac` pays for it)
            "by_epilogue": {k: {"launches_per_step": n, "achieved": fl / (ms * 1e-3) / 1e12 if ms > 0 else None,
                                "share_of_step": ms / ms_per_step} for k, (n, ms, fl) in sorted(meter.fused_summary().items())},
            "by_shape": meter.shape_summary(),
            "weight_streaming_gemm": {"kernel": "haff_gemm_bf16 launches with M <= 64 (KV-cached decode steps, [SEG] MLP): gemm_skinny_kernel, or the split-K 128x128 tile path for wide / deep weights at 33..64 rows",
                                      "bound": "hbm", "launches_per_step": ws_n, "share_of_step": ws_ms / ms_per_step,
                                      "achieved": (ws_bytes / (ws_ms * 1e-3) / 1e9) if ws_ms > 0 else None, "peak": 8000.0,
                                      "unit": "GB/s", "algorithmic_bytes": "2*N*K (the weight matrix, read once)"},
            "fp32_decoder_tail_gemm": {"kernel": "gemm_f32_kernel (v_mfma_f32_16x16x4_f32; text_hidden_fcs + both mask decoders)",
                                       "launches_per_step": f32_n, "share_of_step": f32_ms / ms_per_step,
                                       "achieved": (f32_fl / (f32_ms * 1e-3) / 1e12) if f32_ms > 0 else None, "peak": 157.3,
                                       "unit": "TFLOP/s"},
            "whole_path": {"flops_per_frame": flops_frame, "flops_per_frame_source": "SURVEY.md 8(d) (counts the padded window rows)",
                           "achieved": fps / world * flops_frame / 1e12,
                           "frac": fps / world * flops_frame / 1e12 / peak,
                           "executed_flops_per_frame": executed_frame,
                           "executed_note": "every GEMM launch as issued (padded window rows skipped) + attention / rel-pos terms",
                           "executed_achieved": fps / world * executed_frame / 1e12,
                           "executed_frac": fps / world * executed_frame / 1e12 / peak},
        }
        T_exp = 4 + args.text_tokens + cfg.clip.n_patches - 1
        line = base_line(fps, world, args.steps, args.warmup, ms_per_step,
                         "BASELINE.json %s: %s, %d x %dx%d uint8 NHWC frames/step/GPU, %d-token prompt (T=%d), %d forced "
                         "answer tokens with [SEG], KV-cached greedy decode, random-init weights; CLIP + SAM preprocessing of the "
                         "uint8 frames on the device inside the step" % (
                             {("2HandedAfforder-7B", 64): "configs[2]", ("2HandedAfforder-7B", 1): "configs[1]",
                              ("2HandedAfforder-13B", 8): "configs[4]"}.get((cfg.name, B), "geometry (no BASELINE config of this batch)"),
                             cfg.name, B, S, S, args.text_tokens, T_exp, args.n_gen),
                         B, {"hip_streams": 2 if model.overlap_streams else 1, "fp32_decoder_tail": bool(model.fp32_tail),
                             "fp32_residual_stream": args.fp32_stream, "neck_f32": bool(args.neck_f32),
                             "overlap_rates": model.last_rates.source if model.last_rates is not None else None,
                             # how the two streams shared the CUs in the timed steps (overlap.py): frames per encoder pass, workgroups per
                             # persistent GEMM launch of each pass (null = one per CU everywhere)
                             "sam_chunk": plan_timed[2], "sam_chunk_workgroup_caps": plan_timed[0],
                             "sam_waits_for_prefill": plan_timed[1],
                             # decode steps of <= 8 rows as ONE chained launch per step (csrc/decode_chain.hip) in the timed steps
                             "decode_chain": chain_timed})
        line["roofline"] = roofline
        line["rccl_ranks"] = n_ranks
        if f32_mode:
            line["dtype"] = "f32"
            line["config"]["numeric_mode"] = "fp32 parity mode (f32-input MFMA products, fp32 attention / norms): the mode the IoU / 1e-3 targets are met in"
        # sanity of the produced masks (finite, right shapes)
        ok = all(m.shape == (1, S, S) and bool(torch.isfinite(m).all()) for m in out[1] + out[2])
        line["outputs_finite"] = ok
        if not args.no_b1:
            for _ in range(2):
                step(1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                step(1)
            torch.cuda.synchronize()
            line["latency_batch1_ms"] = 1e3 * (time.perf_counter() - t1) / 5
            # one KV-cached decode step at batch 1: HBM-bound on the weight stream — report its rate
            T0 = ids.shape[1] + 255
            cache = model._persistent_cache(1, T0 + args.n_gen)
            st = cache["book"]   # the step generate() replays: embedding -> 32 layers -> logits -> argmax -> bookkeeping
            st["t_rows"].fill_(T0); st["lens"].fill_(ids.shape[1]); st["use_forced"].fill_(0); st["finished"].zero_()

            def one_step():
                st["steps"].fill_(1); st["pos"].fill_(T0); st["nk"].fill_(T0 + 1)
                model._decode_book_step(cache)
            for _ in range(3):
                one_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                one_step()
            torch.cuda.synchronize()
            step_ms = 1e3 * (time.perf_counter() - t1) / 10
            l = cfg.llm
            w_bytes = 2.0 * (l.layers * (4 * l.hidden * l.hidden + 3 * l.hidden * l.ffn) + l.vocab * l.hidden)
            line["decode_step_batch1"] = {"ms": step_ms, "weight_bytes": w_bytes, "bound": "hbm",
                                          "achieved_TBps": w_bytes / (step_ms * 1e-3) / 1e12, "peak_TBps": 8.0,
                                          "note": "a read-only streaming kernel reaches 6.25 TB/s on this part (tools/probes/lds_dma_bw.hip)"}
        threads = min(len(os.sched_getaffinity(0)), 32)
        full_parity = None
        if world == 1 and not args.no_parity and not args.no_full_frame_parity and cfg.name in SURVEY_FLOPS:
            # ONE full-depth frame through the HIP path (both numeric modes) and the oracle on shared weights: the parity object's
            # full_frame entry AND cpu_baseline's one real frame (the oracle's exact pass is the timed CPU frame)
            del model
            model = None
            torch.cuda.empty_cache()
            full_parity = parity_full_frame(cfg, device, threads, args.text_tokens, args.n_gen, attribution=args.attribution)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.text_tokens, args.n_gen, threads)
            if not args.no_cpu_full_frame:
                # one REAL frame beside the sample: the full-frame parity's oracle pass when that ran, else a pass of its own
                full = full_parity["cpu_frame"] if full_parity and "cpu_frame" in full_parity else \
                    cpu_full_frame(cfg, args.text_tokens, args.n_gen, threads)
                line["cpu_baseline"]["full_frame"] = full
                if "frames_per_s" in full:
                    # the measured end-to-end frame IS the baseline: value and seconds_per_frame are that frame's; everything that
                    # belongs to the layer-count extrapolation moves into its own sub-object
                    cb = line["cpu_baseline"]
                    cb["extrapolation"] = {"value": cb.pop("value"), "seconds_per_frame": cb.pop("seconds_per_frame"),
                                           "parts_s": cb.pop("parts_s"), "sample": cb.pop("sample"), "note": cb.pop("extrapolation"),
                                           "reference_semantics_value": cb.pop("reference_semantics_value"),
                                           "reference_semantics_note": cb.pop("reference_semantics_note")}
                    cb["value"] = full["frames_per_s"]
                    cb["seconds_per_frame"] = full["seconds"]
                    cb["extrapolated"] = False
                    cb["sample"] = full["what"] + " - ONE measured end-to-end frame (the `extrapolation` object is the per-layer estimate beside it)"
        else:
            line["cpu_baseline"] = None
        if world == 1 and not args.no_parity:
            del model
            line["parity"] = parity_vs_oracle(device)
            # which entry of the parity objects belongs to the number this line prints (the compact line quotes that one)
            line["parity"]["timed_mode"] = "fp32" if f32_mode else ("bf16" if args.fp32_stream == "off" else
                                                                    ("bf16_fp32_stream_f32neck" if args.neck_f32 else "bf16_fp32_stream"))
            if full_parity is not None:
                line["parity"]["full_frame"] = full_parity.get("parity", full_parity)
                line["parity"]["gate"]["failed"] += (full_parity.get("parity") or {}).get("gate", {}).get("failed", [])
        emit(line)
        if (line.get("parity") or {}).get("gate", {}).get("failed"):
            print("parity gate failed: " + ", ".join(line["parity"]["gate"]["failed"]), file=sys.stderr)
            exit_code = 3
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return exit_code


if __name__ == "__main__":
    sys.exit(main())
