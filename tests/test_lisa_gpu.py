"""End-to-end and stage-level parity of the HIP path (through the C-ABI) against the CPU oracle on the same
seeded weights/inputs (BASELINE.json configs[0] "tiny" plus a "mid" geometry with window padding, d=80, d=128).

Tolerances (stated per BASELINE.json north_star):
  * fp32 parity mode: final mask logits within 1e-3 abs of the oracle; binary masks (logit > 0) bit-exact wherever
    the oracle logit is farther than 1e-3 from zero; taxonomy within 1e-4.
  * bf16 throughput mode: the oracle runs on the SAME bf16-rounded weights/inputs in fp32; bf16 storage of
    activations bounds the error by a few percent of the logit scale, checked as rel <= 6e-2 of the max |logit|
    and mask IoU >= 0.97 on these random-weight models (random logits are dense around 0; SURVEY §7 hard part 3).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

V = "model.visual_model"


def _setup(cfg_name, mode, seed=5, B=2, n_gen=4, prompt_len=8):
    import haff  # noqa: F401
    from haff import config as hcfg
    from haff import weights as hw
    cfg = getattr(hcfg, cfg_name)()
    sd = hw.make_state_dict(cfg, seed)
    rng = np.random.default_rng(seed + 7)
    S = cfg.sam.img_size
    images = torch.from_numpy(rng.standard_normal((B, 3, S, S), dtype=np.float32))
    images_clip = torch.from_numpy(rng.standard_normal((B, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    if mode == "bf16":
        hw.round_to_bf16_(sd)
        images = images.to(torch.bfloat16).float()
        images_clip = images_clip.to(torch.bfloat16).float()
    text = torch.from_numpy(rng.integers(3, cfg.llm.vocab - 3, size=(B, prompt_len))).long()
    ids = torch.cat([torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]]).expand(B, -1), text], 1)
    forced = torch.from_numpy(rng.integers(3, cfg.llm.vocab - 3, size=(B, n_gen))).long()
    forced[:, 1] = cfg.seg_token_idx
    forced[:, -1] = cfg.eos_token_id
    return cfg, sd, images, images_clip, ids, forced


def _iou(a, b):
    inter = (a & b).sum().item()
    union = (a | b).sum().item()
    return inter / union if union else 1.0


@pytest.mark.parametrize("cfg_name", ["tiny", "mid"])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_evaluate_matches_oracle(dev, cfg_name, mode):
    from haff.lisa import LisaMI355
    from oracle import lisa_oracle as O
    cfg, sd, images, images_clip, ids, forced = _setup(cfg_name, mode)
    S = cfg.sam.img_size
    B = ids.shape[0]
    resize = [(S, S), (S, S - 32)]
    orig = [(S, S), (S // 2 + 3, S // 2 - 10)]
    with torch.no_grad():
        ref_ids, ref_l, ref_r, ref_t = O.lisa_evaluate(sd, cfg, images_clip, images, ids, resize, orig,
                                                       max_new_tokens=forced.shape[1], forced_answer=forced, use_cache=False)
        if mode == "bf16":
            # the oracle with bf16 roundings at the HIP path's kernel boundaries (fp32 arithmetic between them): its own
            # distance to the exact forward is the noise floor of ANY implementation with this storage format
            with O.bf16_points():
                _, pts_l, pts_r, _ = O.lisa_evaluate(sd, cfg, images_clip, images, ids, resize, orig,
                                                     max_new_tokens=forced.shape[1], forced_answer=forced, use_cache=True)
    dtype = torch.float32 if mode == "f32" else torch.bfloat16
    model = LisaMI355(cfg, sd, dtype=dtype, device=dev)
    out_ids, left, right, tax = model.evaluate(images_clip.to(dev), images.to(dev), ids.to(dev), resize, orig,
                                               max_new_tokens=forced.shape[1], forced_answer=forced)
    assert torch.equal(out_ids.cpu(), ref_ids)
    assert len(left) == len(right) == len(tax) == B
    for i in range(B):
        for got, ref, name in ((left[i], ref_l[i], "left"), (right[i], ref_r[i], "right")):
            assert got.shape == ref.shape and got.dtype == torch.float32
            g = got.cpu()
            err = (g - ref).abs().max().item()
            scale = ref.abs().max().item()
            iou = _iou(g > 0, ref > 0)
            print(f"{cfg_name}/{mode} frame{i} {name}: max|err|={err:.3e} scale={scale:.3f} std={ref.std():.3f} IoU={iou:.5f}")
            if mode == "f32":
                assert err <= 1e-3, f"{name} logits off by {err}"
                safe = ref.abs() > 1e-3
                assert torch.equal((g > 0)[safe], (ref > 0)[safe])
            else:
                # measured on MI355X (round 3): err / scale 5.7e-3 ... 7.2e-3, IoU 0.9949 ... 0.9979 on these seeds; the bounds
                # are ~2x that, so a kernel that loses one more bit (let alone a 10x regression) turns this red. The
                # bf16-points oracle is just as far from the exact forward (5e-3 ... 8.4e-3): the path is at the floor of
                # its storage format, and it must stay within 3x of that floor measured on the same inputs.
                pts = (pts_l if name == "left" else pts_r)[i]
                floor = (pts - ref).abs().max().item()
                print(f"    bf16-points oracle vs exact: {floor / scale:.3e} of scale; HIP vs bf16-points: "
                      f"{(g - pts).abs().max().item() / scale:.3e}")
                assert err <= 1.5e-2 * scale, f"{name} logits rel err {err / scale}"
                assert err <= 3.0 * floor + 2e-3 * scale, f"{name}: {err / scale:.3e} vs floor {floor / scale:.3e}"
                assert iou >= 0.985
                # every pixel the bf16 path decides differently is a near-zero logit of the exact forward: outside a band of 2 % of
                # the logit scale around the threshold the two masks are IDENTICAL (on a trained checkpoint's two-plateau field
                # that band is the object boundary; on these Gaussian fields it holds ~1-2 % of the pixels)
                keep = ref.abs() >= 0.02 * scale
                assert torch.equal((g > 0)[keep], (ref > 0)[keep]), f"{name}: a disagreement at |logit| >= 2 % of the scale"
        terr = (tax[i].cpu() - ref_t[i]).abs().max().item()
        print(f"{cfg_name}/{mode} frame{i} taxonomy err {terr:.3e}")
        assert terr <= (1e-4 if mode == "f32" else 1e-3)   # measured 1.1e-4 ... 2.4e-4 in bf16 mode


@pytest.mark.parametrize("cfg_name", ["tiny", "mid"])
def test_fp32_residual_streams_are_closer_to_the_oracle(dev, cfg_name):
    """LisaMI355(fp32_stream=True): bf16 MFMA products on fp32 ViT-H / Llama residual streams (image_encoder.py:186-193 and the
    LlamaDecoderLayer residual adds, each carried in fp32 between the products). Same guards as the default bf16 mode, the image
    embedding must be CLOSER to the exact oracle than the default mode's, and the fp32-tail outputs stay finite / shaped."""
    from haff.lisa import LisaMI355
    from oracle import lisa_oracle as O
    cfg, sd, images, images_clip, ids, forced = _setup(cfg_name, "bf16")
    S = cfg.sam.img_size
    resize, orig = [(S, S), (S, S - 32)], [(S, S), (S // 2 + 3, S // 2 - 10)]
    with torch.no_grad():
        taps = {}
        ref_ids, ref_l, ref_r, ref_t = O.lisa_evaluate(sd, cfg, images_clip, images, ids, resize, orig, max_new_tokens=forced.shape[1],
                                                       forced_answer=forced, use_cache=True, taps=taps)
    emb_ref = taps["image_embeddings"]
    g = cfg.sam.grid
    res = {}
    for name, fs in (("default", False), ("stream32", True)):
        model = LisaMI355(cfg, sd, dtype=torch.bfloat16, device=dev, fp32_stream=fs)
        out_ids, left, right, tax = model.evaluate(images_clip.to(dev), images.to(dev), ids.to(dev), resize, orig,
                                                   max_new_tokens=forced.shape[1], forced_answer=forced)
        assert torch.equal(out_ids.cpu(), ref_ids)
        emb = model.sam_encoder(images.to(dev)).float().view(-1, g, g, cfg.sam.out_chans).permute(0, 3, 1, 2).cpu()
        e_emb = ((emb - emb_ref).pow(2).mean().sqrt() / emb_ref.pow(2).mean().sqrt()).item()
        errs, ious = [], []
        for i in range(len(left)):
            for got, ref in ((left[i], ref_l[i]), (right[i], ref_r[i])):
                gm = got.cpu()
                errs.append((gm - ref).abs().max().item() / ref.abs().max().item())
                ious.append(_iou(gm > 0, ref > 0))
            assert (tax[i].cpu() - ref_t[i]).abs().max().item() <= 1e-3
        res[name] = (e_emb, max(errs), min(ious))
        print(f"{cfg_name} {name}: embedding rms rel {e_emb:.3e}, logits max rel {max(errs):.3e}, min IoU {min(ious):.5f}")
        del model
    assert res["stream32"][1] <= 1.5e-2 and res["stream32"][2] >= 0.985
    assert res["stream32"][0] < res["default"][0], res   # 4 blocks deep the gain is modest; at depth 32 it is 2.8x


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_stages_match_oracle(dev, mode):
    """Stage taps on the mid geometry: SAM encoder blocks, CLIP features, projector, Llama hidden (prefill +
    KV-cached steps vs the oracle's no-cache recompute), decoder low-res logits."""
    from haff.lisa import LisaMI355
    from oracle import lisa_oracle as O
    cfg, sd, images, images_clip, ids, forced = _setup("mid", mode)
    dtype = torch.float32 if mode == "f32" else torch.bfloat16
    tol = 2e-4 if mode == "f32" else 2e-2   # bf16 measured: 4.9e-3 (CLIP) ... 9.8e-3 (SAM block 3)
    model = LisaMI355(cfg, sd, dtype=dtype, device=dev)

    def rel(got, ref):
        return ((got.float().cpu() - ref).abs().max() / (ref.abs().max() + 1e-12)).item()
    with torch.no_grad():
        taps_ref, taps = {}, {}
        emb_ref = O.sam_image_encoder(sd, V + ".image_encoder", images, cfg.sam, taps_ref)
        emb = model.sam_encoder(images.to(dev), taps)
        g = cfg.sam.grid
        for i in range(cfg.sam.depth):
            r = rel(taps[f"block{i}"], taps_ref[f"block{i}"])
            print(f"{mode} sam block{i} rel {r:.3e}")
            assert r <= tol
        r = rel(emb.view(-1, g, g, cfg.sam.out_chans).permute(0, 3, 1, 2), emb_ref)
        print(f"{mode} sam neck rel {r:.3e}")
        assert r <= tol * 2
        f_ref = O.encode_images(sd, cfg, images_clip)
        f = model.encode_images(images_clip.to(dev))
        r = rel(f, f_ref)
        print(f"{mode} clip+projector rel {r:.3e}")
        assert r <= tol
        ids_ref, hid_ref = O.lisa_generate(sd, cfg, images_clip, ids, forced.shape[1], forced, use_cache=False)
        out_ids, hid = model.generate(images_clip.to(dev), ids.to(dev), forced.shape[1], forced)
        assert torch.equal(out_ids.cpu(), ids_ref) and hid.shape == hid_ref.shape
        r = rel(hid, hid_ref)
        print(f"{mode} llama hidden rel {r:.3e}")
        assert r <= tol
        # decoder fed with the ORACLE's embedding and text so this stage is checked in isolation
        text = torch.from_numpy(np.random.default_rng(3).standard_normal((3, cfg.out_dim), dtype=np.float32))
        if mode == "bf16":
            text = text.to(torch.bfloat16).float()
            emb_ref = emb_ref.to(torch.bfloat16).float()
        pe = O.sam_dense_pe(sd, V + ".prompt_encoder", (g, g))
        fidx = torch.tensor([0, 1, 1])
        emb_cl = emb_ref.flatten(2).permute(0, 2, 1).contiguous().to(dev, dtype)
        lo_l, lo_r, tax, iou_l, iou_r = model.sam_decoder.decode(emb_cl, fidx.to(dev), text.to(dev, dtype))
        for p in range(3):
            sp, de = O.sam_prompt_encoder_text(sd, V + ".prompt_encoder", text[p:p + 1, None], (g, g))
            rl, ril, rt = O.sam_mask_decoder(sd, V + ".mask_decoder_left", emb_ref[fidx[p]:fidx[p] + 1], pe, sp, de, True)
            rr, rir = O.sam_mask_decoder(sd, V + ".mask_decoder_right", emb_ref[fidx[p]:fidx[p] + 1], pe, sp, de, False)
            e1, e2 = rel(lo_l[p], rl[0, 0]), rel(lo_r[p], rr[0, 0])
            print(f"{mode} decoder prompt{p} rel L {e1:.3e} R {e2:.3e} tax {rel(tax[p], rt[0]):.3e} iou {rel(iou_l[p], ril[0]):.3e}")
            assert max(e1, e2) <= tol * 2 and rel(tax[p], rt[0]) <= tol * 2 and rel(iou_l[p], ril[0]) <= tol * 2


def test_u8_ingest_equals_float_contract(dev):
    """uint8 NHWC frame ingest (fused normalise+pad+patchify) == the evaluate() float-tensor contract."""
    from haff.lisa import LisaMI355
    from haff.preprocess import sam_preprocess
    cfg, sd, _, images_clip, ids, forced = _setup("tiny", "bf16")
    S = cfg.sam.img_size
    rng = np.random.default_rng(0)
    frames = torch.from_numpy(rng.integers(0, 256, size=(2, S, S, 3), dtype=np.uint8))
    images = torch.stack([sam_preprocess(f, S) for f in frames])
    model = LisaMI355(cfg, sd, dtype=torch.bfloat16, device=dev)
    a = model.get_visual_embs(images.to(dev))
    b = model.get_visual_embs_u8(frames.to(dev), (123.675, 116.28, 103.53), (58.395, 57.12, 57.375))
    assert torch.equal(a, b)


def test_no_seg_token_gives_empty_masks(dev):
    from haff.lisa import LisaMI355
    cfg, sd, images, images_clip, ids, forced = _setup("tiny", "bf16")
    forced[:, 1] = 5  # no [SEG] anywhere
    model = LisaMI355(cfg, sd, dtype=torch.bfloat16, device=dev)
    S = cfg.sam.img_size
    _, left, right, tax = model.evaluate(images_clip.to(dev), images.to(dev), ids.to(dev), [(S, S)] * 2, [(S, S)] * 2,
                                         max_new_tokens=4, forced_answer=forced)
    assert all(m.shape == (0, S, S) for m in left + right) and all(t.shape == (0, 4) for t in tax)


def test_batch_invariance_and_determinism(dev):
    """Repeated runs are bitwise identical; frame i's masks do not depend on its batch neighbours beyond bf16
    accumulation-order noise (decode-sized products, M <= 16, take the weight-streaming GEMM, whose K split differs from
    the tiled kernel's: like the reference's cuBLAS path, results are not bitwise batch-invariant)."""
    from haff.lisa import LisaMI355
    cfg, sd, images, images_clip, ids, forced = _setup("tiny", "bf16", B=3)
    S = cfg.sam.img_size
    model = LisaMI355(cfg, sd, dtype=torch.bfloat16, device=dev)
    args = lambda sl: (images_clip[sl].to(dev), images[sl].to(dev), ids[sl].to(dev), [(S, S)] * len(ids[sl]), [(S, S)] * len(ids[sl]))
    _, l3, r3, t3 = model.evaluate(*args(slice(0, 3)), max_new_tokens=4, forced_answer=forced)
    _, l3b, _, _ = model.evaluate(*args(slice(0, 3)), max_new_tokens=4, forced_answer=forced)
    _, l1, r1, t1 = model.evaluate(*args(slice(1, 2)), max_new_tokens=4, forced_answer=forced[1:2])
    assert all(torch.equal(a, b) for a, b in zip(l3, l3b))
    for a, b in ((l3[1], l1[0]), (r3[1], r1[0])):
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
        assert _iou(a > 0, b > 0) >= 0.97   # random-weight logits are dense around 0 (module docstring)
    assert (t3[1] - t1[0]).abs().max().item() <= 2e-2


def test_stream_schedules_do_not_change_results(dev):
    """overlap.py / haff_gemm_stream_cap: which CUs the encoder's GEMM launches take, how the frames are grouped into encoder
    passes, whether the encoder's stream waits for the prefill, one stream or two — scheduling only: ids, masks and taxonomy
    rows are bit for bit the same under every setting (here on a small geometry with 5 frames, so that the 'late' mode with a
    real wait, several passes and a ragged last pass are all exercised; the full-size schedules, where the grouping of the
    frames does not change the bits either: test_fullsize_gpu.py)."""
    from haff.lisa import LisaMI355
    cfg, sd, images, images_clip, ids, forced = _setup("mid", "bf16", B=5)
    S = cfg.sam.img_size
    model = LisaMI355(cfg, sd, dtype=torch.bfloat16, device=dev, sam_chunk=2)
    args = (images_clip.to(dev), images.to(dev), ids.to(dev), [(S, S)] * 5, [(S, S)] * 5)

    def run():
        o, l, r, t = model.evaluate(*args, max_new_tokens=4, forced_answer=forced)
        return [o] + l + r + t
    refs = {}
    for chunk, caps, wait, two in ((2, None, "auto", False), (2, None, "auto", True), (2, [256, 128, 64], True, True), (2, [8], True, True),
                                   (5, None, "auto", False), (5, [224], False, True), (5, "auto", "auto", True),
                                   (1, None, "auto", False), (1, [256, 256, 32, 32, 32], True, True),
                                   (3, None, "auto", False), (3, [96, 160], "auto", True)):
        model.sam_chunk, model.sam_chunk_caps, model.sam_waits_for_prefill, model.overlap_streams = chunk, caps, wait, two
        got = run()
        if chunk not in refs:     # (small passes take other kernels by row count: a reference per grouping, one stream, no caps)
            assert not two and caps is None
            refs[chunk] = got
            continue
        assert all(torch.equal(a, b) for a, b in zip(refs[chunk], got)), (chunk, caps, wait, two)
        if two and caps not in (None, "auto"):
            assert model.last_plan[0] == caps and model.last_plan[1] == (wait is True or wait == "auto")   # 5 frames: late mode
    from haff import ops
    # every evaluate() leaves both streams' settings where it found them
    assert ops.gemm_stream_cap(0) == 256 and ops.gemm_stream_cap(0, stream=model._sam_stream) == 256


def test_sam_vith_width_windowed_blocks(dev):
    """ViT-H block geometry (dim 1280, 16 heads of 80, 14x14 windows on a 64x64 grid -> 5x5 windows with padding),
    depth cut to 3 (windowed, global, windowed): exercises the fused window-attention kernel, the pad-token
    substitution and the gather/scatter GEMMs, which the tiny/mid geometries cannot reach.
    (1) compact path (real tokens only) == padded path (every window row computed) up to bf16 rounding noise;
    (2) both within the bf16 tolerance of the fp32 CPU oracle on the same bf16-rounded weights."""
    import copy
    import haff  # noqa: F401
    from haff import config as hcfg
    from haff import weights as hw
    from haff.sam import SamEncoderHip
    from oracle import lisa_oracle as O
    cfg = copy.deepcopy(hcfg.haff_7b())
    cfg.sam.depth, cfg.sam.global_idx = 3, (1,)
    shapes = {k: v for k, v in hw.all_shapes(cfg).items() if k.startswith(V + ".image_encoder")}
    sd = hw.make_state_dict(cfg, 11, shapes)
    hw.round_to_bf16_(sd)
    rng = np.random.default_rng(12)
    images = torch.from_numpy(rng.standard_normal((2, 3, 1024, 1024), dtype=np.float32)).to(torch.bfloat16).float()
    cfg.sam.fold_norms = True        # also build the norm-folded weights (haff_gemm_bf16_ln path, checked below)
    enc = SamEncoderHip(sd, cfg.sam, torch.bfloat16, dev)
    enc.fold_norms = False           # first the LayerNorm kernels; the folded path (the ViT-H default) is compared below
    with torch.no_grad():
        taps_c, taps_p, taps_ref = {}, {}, {}
        enc.compact_windows = True
        out_c = enc(images.to(dev), taps_c).float().cpu()
        assert enc.head_major_windows   # the default: q|k|v of the windowed blocks scattered head-major (haff_gemm_bf16_heads)
        enc.head_major_windows = False
        out_tm = enc(images.to(dev)).float().cpu()
        enc.head_major_windows = True
        assert torch.equal(out_c, out_tm), "head-major q|k|v planes must be a pure re-layout of the token-major buffer"
        enc.compact_windows = False
        out_p = enc(images.to(dev), taps_p).float().cpu()
        ref = O.sam_image_encoder(sd, V + ".image_encoder", images, cfg.sam, taps_ref)
    g = cfg.sam.grid
    ref_cl = ref.permute(0, 2, 3, 1).reshape(2, g * g, -1)
    scale = ref_cl.abs().max().item()
    d_cp = (out_c - out_p).abs().max().item() / scale
    d_ref = (out_c - ref_cl).abs().max().item() / scale
    for i in range(3):
        r = (taps_c[f"block{i}"] - taps_ref[f"block{i}"]).abs().max().item() / taps_ref[f"block{i}"].abs().max().item()
        print(f"vit-h width block{i} rel {r:.3e}")
        assert r <= 2e-2          # measured 7.9e-3 ... 9.8e-3
    print(f"compact vs padded {d_cp:.3e}, compact vs oracle {d_ref:.3e}")
    assert d_cp <= 1e-3 and d_ref <= 1.5e-2   # measured 0 (bit-identical) and 6.3e-3
    # norms folded into qkv / lin1 (row statistics + GEMM epilogue) against the same oracle
    with torch.no_grad():
        enc.compact_windows, enc.fold_norms = True, True
        out_f = enc(images.to(dev)).float().cpu()
    d_f = (out_f - ref_cl).abs().max().item() / scale
    print(f"folded norms vs oracle {d_f:.3e}")
    assert d_f <= 1.5e-2 and (out_f - out_c).abs().max().item() / scale <= 1.5e-2   # measured 7.1e-3
    # ... with the row statistics handed on by the producing products' epilogues (proj / lin2: haff_gemm_bf16_rowstats; what
    # batches of >= 4 frames run) instead of a statistics pass over the stored rows
    with torch.no_grad():
        enc.producer_stats = "force"
        out_s = enc(images.to(dev)).float().cpu()
        enc.producer_stats = False
        out_n = enc(images.to(dev)).float().cpu()
    d_s = (out_s - ref_cl).abs().max().item() / scale
    print(f"folded norms + producer statistics vs oracle {d_s:.3e}, vs statistics pass {(out_s - out_n).abs().max().item() / scale:.3e}")
    assert d_s <= 1.5e-2 and (out_s - out_n).abs().max().item() / scale <= 1.5e-2
    # fp32 RESIDUAL STREAM (round 6): fused — proj / lin2 write the fp32 stream and its bf16 copy in one epilogue
    # (haff_gemm_bf16_rowstats32), norms stay folded — against round 5's unfused form (LayerNorm kernels on the fp32 stream) and the
    # oracle: the stream taps of both must be closer to the oracle's than the bf16 stream's, and close to each other
    with torch.no_grad():
        enc.producer_stats, enc.fp32_stream, enc.fused_fp32_stream = "force", True, True
        taps_f = {}
        out_32f = enc(images.to(dev), taps_f).float().cpu()
        enc.fused_fp32_stream = False
        taps_u = {}
        out_32u = enc(images.to(dev), taps_u).float().cpu()
        enc.fused_fp32_stream, enc.neck_f32 = True, True
        out_32n = enc(images.to(dev)).float().cpu()
        enc.neck_f32, enc.fp32_stream, enc.producer_stats = False, False, True
    for i in range(3):
        ref_t = taps_ref[f"block{i}"]
        rf, ru, rb = [((t[f"block{i}"] - ref_t).pow(2).mean().sqrt() / ref_t.pow(2).mean().sqrt()).item() for t in (taps_f, taps_u, taps_c)]
        print(f"vit-h width block{i} stream rms rel: fused fp32 {rf:.3e}, unfused fp32 {ru:.3e}, bf16 {rb:.3e}")
        assert rf < rb and ru < rb, (i, rf, ru, rb)
        assert rf <= 1.6 * ru + 1e-4, (i, rf, ru)       # fused rounds the stream once per operand, unfused the normalised row: same size
    d32 = [(o - ref_cl).abs().max().item() / scale for o in (out_32f, out_32u, out_32n)]
    print(f"fp32 stream vs oracle: fused {d32[0]:.3e}, unfused {d32[1]:.3e}, fused + f32 neck {d32[2]:.3e} (bf16 stream {d_s:.3e})")
    assert max(d32) <= 1.5e-2


def test_decode_graphs_match_eager(dev):
    """hipGraph-captured decode steps (capture on the first call, replay on the second) reproduce the eager greedy
    loop bit for bit: ids, hidden states, masks."""
    from haff.lisa import LisaMI355
    cfg, sd, images, images_clip, ids, forced = _setup("tiny", "bf16", B=3, n_gen=5)
    model = LisaMI355(cfg, sd, dtype=torch.bfloat16, device=dev)
    sizes = [(cfg.sam.img_size, cfg.sam.img_size)] * 3

    def run(free_running):
        kw = dict(max_new_tokens=5, forced_answer=None if free_running else forced)
        with torch.no_grad():
            o, h = model.generate(images_clip.to(dev), ids.to(dev), kw["max_new_tokens"], kw["forced_answer"])
            ev = model.evaluate(images_clip.to(dev), images.to(dev), ids.to(dev), sizes, sizes, **kw)
        return o.cpu(), h.float().cpu(), [m.float().cpu() for m in ev[1]]
    for free_running in (False, True):
        model.decode_graphs = False
        ref = run(free_running)
        model.decode_graphs = True
        cap = run(free_running)      # captures
        rep = run(free_running)      # replays
        assert len(model._graphs) > 0
        for got in (cap, rep):
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
            assert all(torch.equal(a, b) for a, b in zip(got[2], ref[2]))


def test_multiple_and_missing_seg_tokens(dev):
    """n_seg differs per sample (LISA.py:467-485 cumsum split): two [SEG] in frame 0, none in frame 1, one in frame 2 —
    mask lists, taxonomy rows and values against the oracle (fp32 parity mode)."""
    from haff.lisa import LisaMI355
    from oracle import lisa_oracle as O
    cfg, sd, images, images_clip, ids, forced = _setup("tiny", "f32", B=3, n_gen=5)
    forced = forced.clone()
    forced[:, :-1] = 7
    forced[0, 1] = forced[0, 3] = cfg.seg_token_idx
    forced[2, 2] = cfg.seg_token_idx
    forced[:, -1] = cfg.eos_token_id
    S = cfg.sam.img_size
    sizes = [(S, S)] * 3
    with torch.no_grad():
        ref_ids, ref_l, ref_r, ref_t = O.lisa_evaluate(sd, cfg, images_clip, images, ids, sizes, sizes,
                                                       max_new_tokens=5, forced_answer=forced, use_cache=False)
        model = LisaMI355(cfg, sd, dtype=torch.float32, device=dev)
        out_ids, left, right, tax = model.evaluate(images_clip.to(dev), images.to(dev), ids.to(dev), sizes, sizes,
                                                   max_new_tokens=5, forced_answer=forced)
    assert torch.equal(out_ids.cpu(), ref_ids)
    assert [m.shape[0] for m in left] == [2, 0, 1] == [m.shape[0] for m in ref_l]
    assert [t.shape for t in tax] == [t.shape for t in ref_t]
    for got, ref in zip(left + right, ref_l + ref_r):
        assert got.shape == ref.shape
        if ref.numel():
            assert (got.cpu() - ref).abs().max().item() <= 1e-3
    for got, ref in zip(tax, ref_t):
        if ref.numel():
            assert (got.cpu() - ref).abs().max().item() <= 1e-4


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_evaluate_matches_the_reference_own_evaluate(dev, mode):
    """LisaMI355.evaluate END TO END against what the reference's OWN `LISAForCausalLM.evaluate` source returned
    (tests/golden/lisa_evaluate_tiny.npz: LISA.py:432-534 run unchanged on the reference's Sam classes by
    oracle/make_golden.py::lisa_evaluate_golden, its generate() fed by the CPU oracle): three frames, two [SEG] / none / one,
    three different (resize, original) size pairs. fp32 mode: ids equal, masks within 1e-3 (north_star's tolerance), taxonomy
    within 1e-5; bf16: inside the bf16 band."""
    import haff  # noqa: F401
    from haff import config as hcfg, weights as hw
    from haff.lisa import LisaMI355
    cfg = hcfg.tiny()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lisa_evaluate_tiny.npz"))
    seed = int(g["seed"])
    sd = hw.make_state_dict(cfg, seed)
    S = cfg.sam.img_size
    rng = np.random.default_rng(seed + 6000)
    images = torch.from_numpy(rng.standard_normal((3, 3, S, S), dtype=np.float32))
    images_clip = torch.from_numpy(rng.standard_normal((3, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    assert abs(float(images.double().sum()) - float(g["images_checksum"])) < 1e-6
    if mode == "bf16":
        hw.round_to_bf16_(sd)      # (the golden is the exact forward of the unrounded weights: the band below covers the weights' rounding too)
    resize = [tuple(int(v) for v in r) for r in g["resize_list"]]
    orig = [tuple(int(v) for v in r) for r in g["original_size_list"]]
    dtype = torch.float32 if mode == "f32" else torch.bfloat16
    model = LisaMI355(cfg, sd, dtype=dtype, device=dev)
    ids, forced = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["forced"])
    with torch.no_grad():
        out_ids, left, right, tax = model.evaluate(images_clip.to(dev), images.to(dev), ids.to(dev), resize, orig,
                                                   max_new_tokens=5, forced_answer=forced)
    assert torch.equal(out_ids.cpu(), torch.from_numpy(g["output_ids"]))
    assert [m.shape[0] for m in left] == [2, 0, 1] and [m.shape[0] for m in right] == [2, 0, 1]
    for i in range(3):
        for got, key in ((left[i], f"left{i}"), (right[i], f"right{i}"), (tax[i], f"tax{i}")):
            ref = torch.from_numpy(g[key])
            assert tuple(got.shape) == tuple(ref.shape), (key, got.shape, ref.shape)
            if ref.numel() == 0:
                continue
            err = (got.float().cpu() - ref).abs().max().item()
            if key.startswith("tax"):
                assert err <= (1e-5 if mode == "f32" else 3e-3), (key, err)
            elif mode == "f32":
                assert err <= 1e-3, (key, err)
                safe = ref.abs() > 1e-3
                assert torch.equal((got.cpu() > 0)[safe], (ref > 0)[safe])
            else:
                assert err <= 3e-2 * ref.abs().max().item(), (key, err / ref.abs().max().item())


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_greedy_decode_matches_transformers_generate(dev, mode):
    """LisaMI355.generate (KV-cached greedy decode, hipGraph per step) against transformers' own `generate(num_beams=1)` on the same
    weights (tests/golden/greedy_generate_tiny.npz, oracle/make_golden.py::greedy_generate_golden): the free-running tokens, a row
    that emits EOS at its third step and is padded from then on, and the early end of the loop when every row has finished."""
    import copy
    import haff  # noqa: F401
    from haff import config as hcfg, weights as hw
    from haff.lisa import LisaMI355
    cfg = hcfg.tiny()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "greedy_generate_tiny.npz"))
    seed = int(g["seed"])
    sd = hw.make_state_dict(cfg, 3)
    sd.update(hw.make_state_dict(cfg, seed, {**hw.clip_shapes(cfg.clip), **hw.llm_shapes(cfg)}))
    images = torch.from_numpy(np.random.default_rng(seed + 9000).standard_normal((3, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    assert abs(float(images.double().sum()) - float(g["images_checksum"])) < 1e-6
    dtype = torch.float32 if mode == "f32" else torch.bfloat16
    ids = torch.from_numpy(g["input_ids"]).to(dev)
    L = ids.shape[1]
    cfg_eos = copy.deepcopy(cfg)
    cfg_eos.eos_token_id = int(g["eos_token_id"])
    for c, key, n in ((cfg, "free_tokens", 3), (cfg_eos, "tokens", 3), (cfg_eos, "tokens_row0_alone", 1)):
        model = LisaMI355(c, sd, dtype=dtype, device=dev)
        with torch.no_grad():
            out, _ = model.generate(images[:n].to(dev, dtype), ids[:n], 8)
        got, want = out[:, L:].cpu().tolist(), g[key].tolist()
        if mode == "f32":
            assert got == want, (key, got, want)
        else:   # bf16: an argmax between two near-equal logits may go the other way; every token up to a first such flip must agree
            agree = sum(int(a == b) for ra, rb in zip(got, want) for a, b in zip(ra, rb))
            assert agree >= 0.9 * sum(len(r) for r in want), (key, got, want)
        del model


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_language_half_matches_the_reference_own_forward(dev, mode):
    """The HIP path's CLIP tower -> projector -> splice -> Llama prefill -> lm_head against what the reference's OWN
    `LlavaLlamaForCausalLM.forward` returned over its own llava_arch / clip_encoder code and transformers' models
    (tests/golden/llava_llama_forward_tiny.npz, oracle/make_golden.py::llava_llama_forward_golden): post-norm hidden states of every
    position and the logits of the last 8 — fp32 mode within 2e-4, bf16 inside the bf16 band."""
    import haff  # noqa: F401
    from haff import config as hcfg, ops, weights as hw
    from haff.lisa import LisaMI355
    cfg = hcfg.tiny()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llava_llama_forward_tiny.npz"))
    seed = int(g["seed"])
    sd = hw.make_state_dict(cfg, 3)
    sd.update(hw.make_state_dict(cfg, seed, {**hw.clip_shapes(cfg.clip), **hw.llm_shapes(cfg)}))   # the generator's CLIP / Llama weights
    if mode == "bf16":
        hw.round_to_bf16_(sd)
    images = torch.from_numpy(np.random.default_rng(seed + 8000).standard_normal((3, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    assert abs(float(images.double().sum()) - float(g["images_checksum"])) < 1e-6
    dtype = torch.float32 if mode == "f32" else torch.bfloat16
    model = LisaMI355(cfg, sd, dtype=dtype, device=dev)
    ids = torch.from_numpy(g["input_ids"]).to(dev)
    with torch.no_grad():
        feats = model.encode_images(images.to(dev, dtype))
        img_pos = (ids == -200).int().argmax(1).to(torch.int32)
        x = ops.embed_splice(ids.clamp_min(-200).contiguous(), img_pos, model.llm.embed, feats.contiguous())
        cache = model.llm.new_cache(3, x.shape[1] + 1)
        hidden = model.llm.forward(x, cache)
        logits = model.llm.next_token_logits(hidden[:, -8:].reshape(-1, hidden.shape[-1]).contiguous()).view(3, 8, -1)
    ref_h, ref_l = torch.from_numpy(g["eval_hidden"]), torch.from_numpy(g["eval_logits_tail"])
    tol = 2e-4 if mode == "f32" else 3e-2
    eh = (hidden.float().cpu() - ref_h).abs().max().item() / ref_h.abs().max().item()
    el = (logits.float().cpu() - ref_l).abs().max().item() / ref_l.abs().max().item()
    print(f"{mode}: hidden rel {eh:.3e}, logits rel {el:.3e}")
    assert eh <= tol and el <= tol


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_clip_projector_splice_match_the_reference_own_glue(dev, mode):
    """Rows a4-a6 of the HIP path against what the REFERENCE'S OWN code returned (tests/golden/llava_glue_tiny.npz, written by
    oracle/make_golden.py::llava_glue_golden from clip_encoder.py:31-60 and llava_arch.py:93-347): CLIP tower + feature select +
    projector (LisaMI355.encode_images) and the embedding splice (haff_embed_splice) — fp32 mode within 1e-4, bf16 within the
    bf16 band; the rows that are pure gathers of the embedding table are exact in fp32."""
    import haff  # noqa: F401
    from haff import config as hcfg, ops, weights as hw
    from haff.lisa import LisaMI355
    cfg = hcfg.tiny()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llava_glue_tiny.npz"))
    seed = int(g["seed"])
    sub = {**hw.clip_shapes(cfg.clip), **{k: v for k, v in hw.llm_shapes(cfg).items()
                                          if k.startswith("model.mm_projector") or k == "model.embed_tokens.weight"}}
    sd = hw.make_state_dict(cfg, 3)
    sd.update(hw.make_state_dict(cfg, seed, sub))          # the generator's CLIP / projector / embedding weights
    if mode == "bf16":
        hw.round_to_bf16_(sd)
    images = torch.from_numpy(np.random.default_rng(seed + 5000).standard_normal((3, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    assert abs(float(images.double().sum()) - float(g["images_checksum"])) < 1e-6
    dtype = torch.float32 if mode == "f32" else torch.bfloat16
    model = LisaMI355(cfg, sd, dtype=dtype, device=dev)
    ref_f, ref_e = torch.from_numpy(g["image_features"]), torch.from_numpy(g["inputs_embeds"])
    with torch.no_grad():
        feats = model.encode_images(images.to(dev, dtype))
        ids = torch.from_numpy(g["input_ids"]).to(dev)
        img_pos = (ids == -200).int().argmax(1).to(torch.int32)
        emb = ops.embed_splice(ids.clamp_min(-200).contiguous(), img_pos, model.llm.embed, feats.contiguous())
    tol = 1e-4 if mode == "f32" else 2e-2
    assert feats.shape == ref_f.shape and (feats.float().cpu() - ref_f).abs().max().item() <= tol * ref_f.abs().max().item()
    assert emb.shape == ref_e.shape and (emb.float().cpu() - ref_e).abs().max().item() <= tol * ref_e.abs().max().item()
    if mode == "f32":
        txt = torch.ones(ref_e.shape[1], dtype=torch.bool)
        txt[2:2 + 256] = False                       # rows 2 .. 257 are the image features, the others gathers of the table
        assert torch.equal(emb.cpu()[:, txt], ref_e[:, txt])


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_last_layer_pruning_keeps_every_row_that_is_read(dev, mode):
    """Round 6: in evaluate() the last Llama layer of the prefill runs o_proj / MLP / final norm only on the rows that are read —
    each row's last real position (first-token logits) and the state in front of a [SEG] that is part of the PROMPT (LISA.py:457-465
    gathers hidden[j + 255] for every id position j + 1 that holds [SEG]). Ragged prompts, one prompt WITH a [SEG] in it, one whose
    first generated token is [SEG] (its state is the prefill's last row) and one without any: ids, masks and taxonomy equal the
    unpruned run's (fp32: to 1e-5 of the scale — the selected rows take the few-rows product kernels instead of the big tile, another
    summation order; bf16: inside the bf16 band) and the oracle's; the kept hidden rows are the ones seg_embeddings reads."""
    from haff.lisa import LisaMI355
    from oracle import lisa_oracle as O
    cfg, sd, images, images_clip, _, _ = _setup("tiny", mode, B=3)
    dtype = torch.float32 if mode == "f32" else torch.bfloat16
    model = LisaMI355(cfg, sd, dtype=dtype, device=dev)
    S = cfg.sam.img_size
    head = [cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]
    prompts = [head + [11, 12, cfg.seg_token_idx, 14, 15], head + [21, 22, 23], head + [31, 32, 33, 34, 35, 36, 37]]
    Lmax = max(len(p) for p in prompts)
    ids = torch.full((3, Lmax), cfg.pad_token_id, dtype=torch.long)
    mask = torch.zeros((3, Lmax), dtype=torch.bool)
    for b, p in enumerate(prompts):
        ids[b, :len(p)] = torch.tensor(p)
        mask[b, :len(p)] = True
    forced = torch.tensor([[7, 9, cfg.seg_token_idx, cfg.eos_token_id], [cfg.seg_token_idx, 8, cfg.eos_token_id, 0], [5, 6, 7, cfg.eos_token_id]])
    sizes = [(S, S)] * 3
    args = (images_clip.to(dev), images.to(dev), ids.to(dev), sizes, sizes)
    with torch.no_grad():
        assert model.prune_last_layer
        po, pl, pr, pt = model.evaluate(*args, max_new_tokens=4, forced_answer=forced, attention_mask=mask)
        model.prune_last_layer = False
        fo, fl, fr_, ft = model.evaluate(*args, max_new_tokens=4, forced_answer=forced, attention_mask=mask)
        model.prune_last_layer = True
        # the rows the pruned prefill keeps, against the full prefill's
        _, h_full = model.generate(args[0], args[2], 4, forced, mask)
        _, h_keep = model.generate(args[0], args[2], 4, forced, mask, needed_hidden_only=True)
    assert torch.equal(po, fo)
    assert [m.shape[0] for m in pl] == [2, 1, 0] and [m.shape[0] for m in fl] == [2, 1, 0]    # prompt [SEG] + generated [SEG]; one; none
    tol = 1e-5 if mode == "f32" else 2e-2
    for got, ref in zip(pl + pr, fl + fr_):
        if ref.numel():
            assert (got - ref).abs().max().item() <= tol * ref.abs().max().item()
    for got, ref in zip(pt, ft):
        if ref.numel():
            assert (got - ref).abs().max().item() <= (1e-5 if mode == "f32" else 2e-3)
    T = Lmax + 255
    nz = h_keep[:, :T].float().abs().sum(-1) > 0
    want = torch.zeros_like(nz)
    for b, p in enumerate(prompts):
        want[b, len(p) + 255 - 1] = True
        nz[b, len(p) + 255:] = False      # (behind a row's last real token: the states of ITS generated tokens, written by the decode steps)
    want[0, 5 + 255] = True          # id position 6 of row 0 is [SEG]: the state at id position 5
    assert torch.equal(nz.cpu(), want.cpu()), "the pruned prefill keeps exactly the rows evaluate() reads"
    e = (h_keep[:, :T][nz] - h_full[:, :T][nz]).float().abs().max().item()
    assert e <= tol * h_full[:, :T][nz].float().abs().max().item()
    with torch.no_grad():
        for b, p in enumerate(prompts[:2]):
            one = torch.tensor([p])
            ro, rl, rr, rt = O.lisa_evaluate(sd, cfg, images_clip[b:b + 1], images[b:b + 1], one, sizes[:1], sizes[:1],
                                             max_new_tokens=4, forced_answer=forced[b:b + 1], use_cache=True)
            assert rl[0].shape == pl[b].shape
            scale = rl[0].abs().max().item()
            assert (pl[b].cpu() - rl[0]).abs().max().item() <= (1e-3 if mode == "f32" else 1.5e-2 * scale)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_ragged_prompts_equal_batch1_runs(dev, mode):
    """Batched evaluate() over three prompts of DIFFERENT lengths (right-padded ids + attention mask, the padding rule of
    utils/dataset.py:90-93,144-150) == three independent batch-1 runs: output ids (left-aligned per row), [SEG] position
    per row (LISA.py:457-485), masks and taxonomy. fp32: <= 1e-3 on the logits (north_star tolerance); bf16: the rows go
    through differently shaped GEMM launches, so accumulation order differs — within the bf16 parity band."""
    from haff.lisa import LisaMI355
    from oracle import lisa_oracle as O
    cfg, sd, images, images_clip, _, _ = _setup("tiny", mode, B=3)
    dtype = torch.float32 if mode == "f32" else torch.bfloat16
    model = LisaMI355(cfg, sd, dtype=dtype, device=dev)
    S = cfg.sam.img_size
    rng = np.random.default_rng(11)
    head = [cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]
    prompts = [head + rng.integers(3, 300, size=n).tolist() for n in (3, 9, 6)]
    Lmax = max(len(p) for p in prompts)
    ids = torch.full((3, Lmax), cfg.pad_token_id, dtype=torch.long)
    mask = torch.zeros((3, Lmax), dtype=torch.bool)
    for b, p in enumerate(prompts):
        ids[b, :len(p)] = torch.tensor(p)
        mask[b, :len(p)] = True
    # the [SEG] token sits at a different generated index per row; row 1 stops early
    forced = torch.tensor([[7, cfg.seg_token_idx, 9, 11, cfg.eos_token_id],
                           [cfg.seg_token_idx, 8, cfg.eos_token_id, 0, 0],
                           [5, 6, 7, cfg.seg_token_idx, cfg.eos_token_id]])
    sizes = [(S, S)] * 3
    with torch.no_grad():
        bo, bl, br, bt = model.evaluate(images_clip.to(dev), images.to(dev), ids.to(dev), sizes, sizes, max_new_tokens=5,
                                        forced_answer=forced, attention_mask=mask)
    assert bo.shape == (3, Lmax + 5)
    for b, p in enumerate(prompts):
        one = torch.tensor([p])
        with torch.no_grad():
            o, l, r, t = model.evaluate(images_clip[b:b + 1].to(dev), images[b:b + 1].to(dev), one.to(dev), sizes[:1], sizes[:1],
                                        max_new_tokens=5, forced_answer=forced[b:b + 1])
            ro, rl, rr, rt = O.lisa_evaluate(sd, cfg, images_clip[b:b + 1], images[b:b + 1], one, sizes[:1], sizes[:1],
                                             max_new_tokens=5, forced_answer=forced[b:b + 1], use_cache=True)
        n = o.shape[1]
        assert torch.equal(bo[b, :n].cpu(), o[0].cpu()) and bool((bo[b, n:] == cfg.pad_token_id).all())
        assert torch.equal(o.cpu(), ro)
        for got, ref, orc in ((bl[b], l[0], rl[0]), (br[b], r[0], rr[0])):
            assert got.shape == ref.shape == orc.shape
            scale = orc.abs().max().item()
            e_b1 = (got - ref).abs().max().item()
            e_or = (got.cpu() - orc).abs().max().item()
            print(f"{mode} row{b}: vs batch-1 {e_b1:.2e}, vs oracle {e_or:.2e} (scale {scale:.2f})")
            if mode == "f32":
                assert e_b1 <= 1e-3 and e_or <= 1e-3
            else:
                assert e_b1 <= 4e-2 * scale and e_or <= 6e-2 * scale
        assert (bt[b] - t[0]).abs().max().item() <= (1e-4 if mode == "f32" else 3e-2)


def test_free_running_greedy_tokens_match_oracle(dev):
    """a8 without the forced answer: the argmax token ids of a free-running greedy decode (KV-cached, ragged batch)
    equal the oracle's, step by step, on the tiny and mid geometries in fp32 (ties are not expected on random logits)."""
    from haff.lisa import LisaMI355
    from oracle import lisa_oracle as O
    for cfg_name in ("tiny", "mid"):
        cfg, sd, images, images_clip, ids, _ = _setup(cfg_name, "f32", B=2)
        model = LisaMI355(cfg, sd, dtype=torch.float32, device=dev)
        with torch.no_grad():
            got, hid = model.generate(images_clip.to(dev), ids.to(dev), 6)
            ref, rh = O.lisa_generate(sd, cfg, images_clip, ids, 6, None, use_cache=True)
        assert torch.equal(got.cpu(), ref), (cfg_name, got.cpu(), ref)
        assert (hid.float().cpu() - rh).abs().max().item() <= 2e-4 * rh.abs().max().item()


def test_two_region_frame_with_aimed_hypernetwork_bias(dev):
    """VERDICT r3 item 7 — parity on a field with two plateaus: a frame of two flat regions, the last bias of
    output_hypernetworks_mlps.0 of each decoder shifted (from the ORACLE's own fp32 forward) so that the mask logits sit around
    +10 / -10 on the two regions (tools/parity_bimodal.py; Fisher's direction between the two clusters of the upscaled
    embedding). Both sides run the same modified weights. The construction is robust (12 seeds x 2 hands without a collapsed
    mask, profiles/r4_parity_bimodal_cpu.txt) but a RANDOM encoder separates the clusters by only ~3 scatter widths, so the field
    is not a trained checkpoint's: held here are the same bounds as on the Gaussian fields plus the band rule."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_bimodal as PB
    from haff.lisa import LisaMI355
    case = PB.build_case("tiny", 3)
    cfg, S = case["cfg"], case["cfg"].sam.img_size
    for side in ("left", "right"):
        d = case["diag"][side]
        assert 0.35 <= d["positive_frac"] <= 0.65 and 8.0 <= d["plateau_median_abs_logit"] <= 14.0, d   # two regions, plateaus near +-10
    model = LisaMI355(cfg, case["sd"], dtype=torch.bfloat16, device=dev)
    o_ids, left, right, tax = model.evaluate(case["images_clip"].to(dev), case["images"].to(dev), case["ids"].to(dev), [(S, S)], [(S, S)],
                                             max_new_tokens=4, forced_answer=case["forced"])
    r_ids, r_left, r_right, _ = case["oracle"]
    assert torch.equal(o_ids.cpu(), r_ids)
    for got, ref, name in ((left[0], r_left[0], "left"), (right[0], r_right[0], "right")):
        g = got.cpu()
        scale = ref.abs().max().item()
        err = (g - ref).abs().max().item()
        iou = _iou(g > 0, ref > 0)
        print(f"two-region {name}: max|err| {err / scale:.3e} of scale {scale:.1f}, IoU {iou:.5f}")
        assert err <= 1.5e-2 * scale and iou >= 0.985
        keep = ref.abs() >= 0.02 * scale
        assert torch.equal((g > 0)[keep], (ref > 0)[keep])
