"""Pins the oracle (oracle/lisa_oracle.py) against golden vectors captured from the reference's own SAM
modules and from transformers' Llama/CLIP (oracle/make_golden.py, run in the build container).
CPU-only; weights are rebuilt from (config, seed) — fixtures carry inputs/outputs only."""
import json
import os

import numpy as np
import pytest
import torch

import haff  # noqa: F401
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # tests/golden_cases.py
from haff import config as hcfg
from haff import weights as hw
from oracle import lisa_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
V = "model.visual_model"


def _load(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name + ".npz")).items()}


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _maxerr(a, b):
    return (a - _t(b)).abs().max().item()


@pytest.mark.parametrize("name,cfg", [("sam_tiny", hcfg.tiny()), ("sam_mid", hcfg.mid())])
def test_sam_against_reference_modules(name, cfg):
    g = _load(name)
    sd = hw.make_state_dict(cfg, int(g["seed"]), hw.sam_shapes(cfg.sam))
    taps = {}
    with torch.no_grad():
        emb = O.sam_image_encoder(sd, V + ".image_encoder", _t(g["images"]), cfg.sam, taps)
    for i in range(cfg.sam.depth):
        assert _maxerr(taps[f"block{i}"], g[f"tap_block{i}"]) < 2e-4, f"block {i}"
    assert _maxerr(emb, g["image_embeddings"]) < 2e-4
    grid = (cfg.sam.grid, cfg.sam.grid)
    pe = O.sam_dense_pe(sd, V + ".prompt_encoder", grid)
    assert _maxerr(pe, g["dense_pe"]) < 1e-5
    sparse, dense = O.sam_prompt_encoder_text(sd, V + ".prompt_encoder", _t(g["text_embeds"]), grid)
    assert _maxerr(sparse, g["sparse"]) == 0 and _maxerr(dense[:, :, 0, 0], g["dense_row"]) == 0
    e0 = _t(g["image_embeddings"])[0:1]
    with torch.no_grad():
        lo_l, iou_l, tax = O.sam_mask_decoder(sd, V + ".mask_decoder_left", e0, pe, sparse, dense, True)
        lo_r, iou_r = O.sam_mask_decoder(sd, V + ".mask_decoder_right", e0, pe, sparse, dense, False)
    assert _maxerr(lo_l, g["low_res_left"]) < 2e-4 and _maxerr(lo_r, g["low_res_right"]) < 2e-4
    assert _maxerr(iou_l, g["iou_left"]) < 1e-4 and _maxerr(tax, g["taxonomy"]) < 1e-5
    inp, orig = tuple(int(v) for v in g["input_size"]), tuple(int(v) for v in g["original_size"])
    post = O.sam_postprocess_masks(_t(g["low_res_left"]), cfg.sam.img_size, inp, orig)
    assert _maxerr(post, g["post_left"]) < 1e-5
    # sign (argmax) masks identical wherever the reference logit is not within float noise of zero
    ref = _t(g["low_res_left"])
    safe = ref.abs() > 1e-3
    assert torch.equal((lo_l > 0)[safe], (ref > 0)[safe])


def test_sam_vith_geometry_against_reference_modules():
    """The oracle's image encoder at the REAL ViT-H block geometry (dim 1280, 16 x 80 heads, 14x14 windows on the 64x64 grid
    with padding to 5x5 windows, rel-pos tables (27, 80) / (127, 80), one windowed + one global block, one 1024^2 frame)
    against the reference's own ImageEncoderViT (oracle/make_golden.py::sam_vith_golden; fixture: subsampled output +
    full-map channel sums + per-block statistics). This is the geometry tests/test_lisa_gpu.py's ViT-H-width tests and
    bench.py's cpu_baseline lean on; the tiny / mid goldens stop at grid 20, window 7."""
    import copy
    g = _load("sam_vith_depth2")
    cfg = copy.deepcopy(hcfg.haff_7b())
    cfg.sam.depth, cfg.sam.global_idx = int(g["depth"]), tuple(int(v) for v in g["global_idx"])
    s = cfg.sam
    shapes = {k: v for k, v in hw.sam_shapes(s).items() if ".image_encoder." in k}
    sd = hw.make_state_dict(cfg, int(g["seed"]), shapes)
    rng = np.random.default_rng(int(g["seed"]) + 1000)
    x = torch.from_numpy(rng.standard_normal((1, 3, s.img_size, s.img_size), dtype=np.float32))
    taps = {}
    with torch.no_grad():
        emb = O.sam_image_encoder(sd, V + ".image_encoder", x, s, taps)
    assert emb.shape == (1, 256, 64, 64)
    assert _maxerr(emb[:, :, ::2, ::2], g["emb_sub"]) < 2e-4
    sums = emb.double().sum((0, 2, 3))
    assert (sums - _t(g["emb_channel_sums"])).abs().max().item() < 2e-2          # 4096 terms of O(1) per channel
    assert abs(emb.double().abs().sum().item() - float(g["emb_abs_sum"])) < 1e-6 * float(g["emb_abs_sum"])
    for i in range(s.depth):
        t = taps[f"block{i}"]
        st = g[f"stat_block{i}"]
        assert abs(t.mean().item() - st[0]) < 1e-5 and abs(t.std().item() - st[1]) < 1e-5 and abs(t.abs().max().item() - st[2]) < 1e-3


def test_llama_against_transformers():
    cfg = hcfg.tiny()
    g = _load("llama_tiny")
    sd = hw.make_state_dict(cfg, int(g["seed"]), hw.llm_shapes(cfg))
    x = _t(g["inputs_embeds"])
    taps = {}
    with torch.no_grad():
        h = O.llama_forward(sd, x, cfg.llm, taps=taps)
        logits = torch.nn.functional.linear(h, sd["lm_head.weight"])
    assert _maxerr(taps["layer0"], g["layer0"]) < 1e-4
    assert _maxerr(h, g["hidden"]) < 1e-4 and _maxerr(logits, g["logits"]) < 2e-4
    # KV-cached schedule == full recompute
    cache = [None] * cfg.llm.layers
    with torch.no_grad():
        hs = [O.llama_forward(sd, x[:, :36], cfg.llm, cache)]
        for t in range(36, 40):
            hs.append(O.llama_forward(sd, x[:, t:t + 1], cfg.llm, cache))
    assert _maxerr(torch.cat(hs, 1), g["hidden_cached"]) < 1e-4
    assert _maxerr(torch.cat(hs, 1), g["hidden"]) < 1e-4


def test_clip_against_transformers():
    cfg = hcfg.tiny()
    g = _load("clip_tiny")
    sd = hw.make_state_dict(cfg, int(g["seed"]), hw.clip_shapes(cfg.clip))
    with torch.no_grad():
        f = O.clip_vision_features(sd, "model.vision_tower.vision_tower", _t(g["images"]), cfg.clip)
    assert f.shape == (2, 256, cfg.clip.hidden)
    assert _maxerr(f, g["features"]) < 1e-4


def _glue_case():
    cfg = hcfg.tiny()
    g = _load("llava_glue_tiny")
    seed = int(g["seed"])
    shapes = {**hw.clip_shapes(cfg.clip), **{k: v for k, v in hw.llm_shapes(cfg).items()
                                             if k.startswith("model.mm_projector") or k == "model.embed_tokens.weight"}}
    sd = hw.make_state_dict(cfg, seed, shapes)
    images = torch.from_numpy(np.random.default_rng(seed + 5000).standard_normal((3, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    assert abs(float(images.double().sum()) - float(g["images_checksum"])) < 1e-6     # the generator's images, reproduced
    return cfg, g, sd, images


def test_llava_glue_against_the_reference_own_methods():
    """Rows a4-a6 (round 6): the oracle's encode_images / splice_embeddings / splice_labels / splice_attention_mask against what the
    REFERENCE'S OWN code returned for the same weights and inputs — CLIPVisionTower.forward + feature_select (clip_encoder.py:31-60),
    LlavaMetaForCausalLM.encode_images and prepare_inputs_labels_for_multimodal (llava_arch.py:93-347), run by
    oracle/make_golden.py::llava_glue_golden. Until round 6 this glue was pinned by restatement only."""
    cfg, g, sd, images = _glue_case()
    with torch.no_grad():
        feats = O.encode_images(sd, cfg, images)
    assert feats.shape == (3, 256, cfg.llm.hidden) and _maxerr(feats, g["image_features"]) < 1e-4
    ids = torch.from_numpy(g["input_ids"])
    emb = O.splice_embeddings(sd, ids, torch.from_numpy(g["image_features"]))
    assert emb.shape == g["inputs_embeds"].shape and torch.equal(emb, torch.from_numpy(g["inputs_embeds"]))     # a gather + concat: exact
    am = O.splice_attention_mask(torch.ones_like(ids, dtype=torch.bool))
    assert torch.equal(am, torch.from_numpy(g["attention_mask_out"]))
    # training-shaped rows: right padding + labels
    ids_t, lab = torch.from_numpy(g["input_ids_train"]), torch.from_numpy(g["labels_train"])
    emb_t = O.splice_embeddings(sd, ids_t, torch.from_numpy(g["image_features"]))
    assert torch.equal(emb_t, torch.from_numpy(g["inputs_embeds_train"]))
    assert torch.equal(O.splice_labels(ids_t, lab), torch.from_numpy(g["labels_train_out"]))
    assert torch.equal(O.splice_attention_mask(torch.from_numpy(g["attention_mask_train"])), torch.from_numpy(g["attention_mask_train_out"]))
    # the [SEG] row rule on these ids (LISA.py:457-465): the state in front of the token, shifted by the 255 image rows
    m = O.seg_token_mask(ids, cfg.seg_token_idx)
    assert m.shape == (3, emb.shape[1] - 1) and m[1].nonzero().flatten().tolist() == [255 + ids.shape[1] - 4] and not m[0].any()


def test_lisa_evaluate_against_the_reference_own_method():
    """LISAForCausalLM.evaluate ITSELF (LISA.py:432-534 + get_visual_embs :157-168; its source run unchanged by
    oracle/make_golden.py::lisa_evaluate_golden on the reference's Sam classes, fed the ids / hidden states below) against the oracle's
    lisa_evaluate on the same ids / hidden states (injected through `memo`): the [SEG] row rule, text_hidden_fcs, the cumsum split,
    the per-sample decoders, postprocess to three different (resize, original) size pairs, empty masks for a sample without [SEG].
    Until round 6 this part of the oracle was pinned by restatement only."""
    cfg = hcfg.tiny()
    g = _load("lisa_evaluate_tiny")
    seed = int(g["seed"])
    sd = hw.make_state_dict(cfg, seed)
    S = cfg.sam.img_size
    rng = np.random.default_rng(seed + 6000)
    images = torch.from_numpy(rng.standard_normal((3, 3, S, S), dtype=np.float32))
    assert abs(float(images.double().sum()) - float(g["images_checksum"])) < 1e-6
    resize = [tuple(int(v) for v in r) for r in g["resize_list"]]
    orig = [tuple(int(v) for v in r) for r in g["original_size_list"]]
    out_ids, hidden = torch.from_numpy(g["output_ids"]), torch.from_numpy(g["hidden"])
    memo = {("gen", False, False): (out_ids, hidden)}
    with torch.no_grad():
        ids, left, right, tax = O.lisa_evaluate(sd, cfg, None, images, torch.from_numpy(g["input_ids"]), resize, orig,
                                                max_new_tokens=5, forced_answer=torch.from_numpy(g["forced"]), memo=memo)
    assert torch.equal(ids, out_ids)
    for i in range(3):
        for got, key in ((left[i], f"left{i}"), (right[i], f"right{i}"), (tax[i], f"tax{i}")):
            assert tuple(got.shape) == g[key].shape, (key, got.shape, g[key].shape)
            if got.numel():
                assert _maxerr(got, g[key]) < 2e-4, (key, _maxerr(got, g[key]))
    assert [m.shape[0] for m in left] == [2, 0, 1]
    # ... and the oracle's OWN generate reproduces the ids / hidden states the fixture was generated from (Llama / CLIP are pinned
    # against transformers in the tests above)
    images_clip = torch.from_numpy(rng.standard_normal((3, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    with torch.no_grad():
        o2, h2 = O.lisa_generate(sd, cfg, images_clip, torch.from_numpy(g["input_ids"]), 5, forced_answer=torch.from_numpy(g["forced"]), use_cache=True)
    assert torch.equal(o2, out_ids) and _maxerr(h2, g["hidden"]) < 1e-5


def test_lisa_model_forward_against_the_reference_own_method():
    """LISAForCausalLM.model_forward ITSELF (LISA.py:175-430, with the module's own dice_loss / sigmoid_ce_loss; run unchanged by
    oracle/make_golden.py::lisa_model_forward_golden on the reference's Sam classes, its `super().forward` served by the oracle's
    language-model functions) against the oracle's lisa_model_forward on the same batch: the six losses of a three-sample training
    batch to 1e-5, and the `inference=True` return (one image, two conversations: the `offset` regrouping) to 2e-4."""
    from golden_cases import model_forward_case
    cfg = hcfg.tiny()
    sd, train, infer, g = model_forward_case(cfg)
    with torch.no_grad():
        out = O.lisa_model_forward(sd, cfg, train)
        inf = O.lisa_model_forward(sd, cfg, infer)
    for k in ("loss", "ce_loss", "taxonomy_ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss"):
        want = float(g["train_" + k])
        assert abs(float(out[k]) - want) <= 1e-5 * max(1.0, abs(want)), (k, float(out[k]), want)
    for k in ("pred_masks_left", "pred_masks_right", "pred_taxonomies"):
        assert tuple(inf[k].shape) == g["inference_" + k].shape
        assert _maxerr(inf[k], g["inference_" + k]) < 2e-4, k
    assert inf["pred_masks_left"].shape[:2] == (1, 2)       # one image, the two [SEG] of its two conversations


def test_language_half_against_the_reference_own_forward():
    """LlavaLlamaForCausalLM.forward ITSELF (llava_llama.py:55-135, run unchanged by oracle/make_golden.py::llava_llama_forward_golden
    over the reference's own llava_arch / clip_encoder code and transformers' LlamaModel / CLIPVisionModel; nothing of this oracle ran
    inside it) against the oracle's chain encode_images -> splice_embeddings -> llama_forward -> lm_head -> shift-by-one CE: hidden states
    (training mode returns every layer's, the last one post-norm; eval mode the post-norm tensor), logits and the loss."""
    cfg = hcfg.tiny()
    g = _load("llava_llama_forward_tiny")
    seed = int(g["seed"])
    sd = hw.make_state_dict(cfg, seed, {**hw.clip_shapes(cfg.clip), **hw.llm_shapes(cfg)})
    images = torch.from_numpy(np.random.default_rng(seed + 8000).standard_normal((3, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    assert abs(float(images.double().sum()) - float(g["images_checksum"])) < 1e-6
    ids, labels = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["labels"])
    with torch.no_grad():
        hidden = O.llama_forward(sd, O.splice_embeddings(sd, ids, O.encode_images(sd, cfg, images)), cfg.llm)
        logits = torch.nn.functional.linear(hidden, sd["lm_head.weight"])
        lab = O.splice_labels(ids, labels)
        ce = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(), lab[:, 1:].reshape(-1), ignore_index=-100)
    assert int(g["train_n_hidden"]) == cfg.llm.layers + 1
    assert _maxerr(hidden, g["train_hidden_last"]) < 1e-4 and _maxerr(hidden, g["eval_hidden"]) < 1e-4
    assert _maxerr(logits[:, -8:], g["train_logits_tail"]) < 1e-4 and _maxerr(logits[:, -8:], g["eval_logits_tail"]) < 1e-4
    assert abs(float(logits.double().sum()) - float(g["train_logits_sum"])) <= 1e-5 * logits.numel() ** 0.5 * float(logits.std()) + 1e-2
    assert abs(float(ce) - float(g["train_loss"])) <= 1e-5


def _greedy_case():
    import copy
    cfg = hcfg.tiny()
    g = _load("greedy_generate_tiny")
    seed = int(g["seed"])
    sd = hw.make_state_dict(cfg, seed, {**hw.clip_shapes(cfg.clip), **hw.llm_shapes(cfg)})
    images = torch.from_numpy(np.random.default_rng(seed + 9000).standard_normal((3, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    assert abs(float(images.double().sum()) - float(g["images_checksum"])) < 1e-6
    cfg_eos = copy.deepcopy(cfg)
    cfg_eos.eos_token_id = int(g["eos_token_id"])
    return cfg, cfg_eos, g, sd, images


def test_greedy_loop_against_transformers_generate():
    """The greedy loop the reference delegates to transformers' `generate(num_beams=1)` (LISA.py:443-450) — argmax, append, pad a row
    after its EOS, stop when every row has finished — as transformers' own generate runs it on a tiny LlamaForCausalLM with the
    filler's weights (oracle/make_golden.py::greedy_generate_golden), against the oracle's lisa_generate in BOTH schedules (full
    recompute per token as the reference does, and KV-cached): free-running tokens, a row that stops early and is padded, and the
    early end of the loop when every row has finished."""
    cfg, cfg_eos, g, sd, images = _greedy_case()
    ids = torch.from_numpy(g["input_ids"])
    L = ids.shape[1]
    with torch.no_grad():
        for use_cache in (False, True):
            out, _ = O.lisa_generate(sd, cfg, images, ids, 8, use_cache=use_cache)
            assert out[:, L:].tolist() == g["free_tokens"].tolist(), use_cache
            out, _ = O.lisa_generate(sd, cfg_eos, images, ids, 8, use_cache=use_cache)
            assert out[:, L:].tolist() == g["tokens"].tolist(), use_cache
            out, _ = O.lisa_generate(sd, cfg_eos, images[:1], ids[:1], 8, use_cache=use_cache)
            assert out[:, L:].tolist() == g["tokens_row0_alone"].tolist() and out.shape[1] == L + 3


def test_seg_token_rule_and_losses_closed_form():
    """LISA.py:457-465 — position 255+j is selected iff token j+1 is [SEG]."""
    ids = torch.tensor([[1, 321, -200, 322, 7, 8, 320, 9, 2], [1, 321, -200, 322, 320, 5, 6, 320, 2]])
    m = O.seg_token_mask(ids, 320)
    assert m.shape == (2, 255 + 8)
    assert m[0].nonzero().flatten().tolist() == [255 + 5] and m[1].nonzero().flatten().tolist() == [255 + 3, 255 + 6]


def test_dice_and_bce_losses_closed_form():
    """LISA.py:16-59 against hand-computed values (float64 arithmetic of the same formulas, no torch):
    dice = sum_masks[1 - (2*sum(p/1000 * t) + 1e-6) / (sum(p/1000) + sum(t/1000) + 1e-6)] / (num_masks + 1e-8),
    bce = sum_masks[mean_pixels(softplus(x) - x*t)] / (num_masks + 1e-8). Cases: logits 0 (p = 1/2 everywhere), a perfect
    saturated prediction (loss -> 0), an empty target with a saturated-negative prediction (the eps/eps = 1 corner: loss 0),
    and random logits against the float64 formula."""
    H, W = 6, 10
    t = torch.zeros((2, H, W))
    t[0, :3] = 1.0                       # 30 positive pixels of 60
    t[1, 0, :4] = 1.0                    # 4 positive pixels
    zeros = torch.zeros((2, H, W))
    n = 2.0
    # p = 0.5: numerator = n_pos / 1000, denominator = (0.5 * HW + n_pos) / 1000
    exp_dice = sum(1 - (npos / 1000 + 1e-6) / ((0.5 * H * W + npos) / 1000 + 1e-6) for npos in (30, 4)) / (n + 1e-8)
    assert abs(O.dice_loss(zeros, t, n).item() - exp_dice) < 1e-6
    assert abs(O.sigmoid_ce_loss(zeros, t, n).item() - 2 * np.log(2.0) / (n + 1e-8)) < 1e-6
    sat = (t * 2 - 1) * 40.0             # sigmoid saturates to the target
    assert O.dice_loss(sat, t, n).item() < 1e-5 and O.sigmoid_ce_loss(sat, t, n).item() < 1e-6
    empty = torch.zeros((1, H, W))
    assert abs(O.dice_loss(torch.full((1, H, W), -40.0), empty, 1.0).item()) < 1e-6      # (0 + eps) / (0 + eps) = 1
    assert abs(O.dice_loss(torch.full((1, H, W), 40.0), empty, 1.0).item() - (1 - 1e-6 / (H * W / 1000 + 1e-6))) < 1e-6
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, H, W), generator=g) * 3
    xd, td = x.double().numpy().reshape(2, -1), t.double().numpy().reshape(2, -1)
    pd = 1.0 / (1.0 + np.exp(-xd))
    dice = (1 - (2 * (pd / 1000 * td).sum(-1) + 1e-6) / ((pd / 1000).sum(-1) + (td / 1000).sum(-1) + 1e-6)).sum() / (n + 1e-8)
    bce = (np.logaddexp(0.0, xd) - xd * td).mean(-1).sum() / (n + 1e-8)
    assert abs(O.dice_loss(x, t, n).item() - dice) < 1e-5 and abs(O.sigmoid_ce_loss(x, t, n).item() - bce) < 1e-5


def test_host_helpers_match_reference():
    from haff import prompt as P
    with open(os.path.join(GOLD, "host_helpers.json")) as f:
        g = json.load(f)

    class StubTok:
        bos_token_id = 1

        def __call__(self, text):
            class R:
                pass
            r = R()
            r.input_ids = [1] + [3 + (ord(ch) % 300) for ch in text]
            return r
    for p, ids in zip(g["prompts"], g["ids"]):
        assert P.tokenizer_image_token(p, StubTok()) == ids
    conv = P.conv_llava_v1()
    conv.append_message(conv.roles[0], "<im_start><image><im_end>\nWhere would you hold the mug?")
    conv.append_message(conv.roles[1], "")
    assert conv.get_prompt() == g["conv_llava_v1_prompt"]
    # both templates --conv_type offers (round 6), in every shape the path builds, against conversation_lib.conv_templates itself
    assert sorted(P.conv_templates) == sorted(g["conv_templates"]) == ["llava_llama_2", "llava_v1"]
    for name, ref in g["conv_templates"].items():
        t = P.get_conv(name)
        assert [list(t.roles), t.sep, t.sep2, t.system] == [ref["roles"], ref["sep"], ref["sep2"], ref["system"]]
        for tag, shape in ref["shapes"].items():
            c = P.get_conv(name)
            for q, a in shape["messages"]:
                c.append_message(c.roles[0], q)
                c.append_message(c.roles[1], a)
            assert c.get_prompt() == shape["prompt"], (name, tag)
            assert c.copy().get_prompt() == shape["prompt"]
    assert P.build_chat_prompt("Where would you hold the mug?", True, "llava_llama_2") == \
        g["conv_templates"]["llava_llama_2"]["shapes"]["open_turn"]["prompt"]
    assert P.build_chat_prompt("Where would you hold the mug?", True) == g["conv_llava_v1_prompt"]
    with pytest.raises(ValueError):
        P.get_conv("mpt")


def test_sam_host_preprocess_against_the_reference_own_functions():
    """Row a1's host helpers against the reference's OWN definitions (`preprocess`, inference.py:90-105; `ResizeLongestSide.
    get_preprocess_shape`, transforms.py:102-113; taken out of their files unchanged by oracle/make_golden.py::host_goldens): the
    resize shape for ten frame sizes at both target lengths — exact — and normalise + pad of a small uint8 frame — exact in fp32 —
    for the product's helpers (haff.preprocess) and the oracle's."""
    from haff import preprocess as PP
    with open(os.path.join(GOLD, "host_helpers.json")) as f:
        g = json.load(f)
    got = [list(PP.get_preprocess_shape(h, w, L)) for (h, w) in g["preprocess_shape_sizes"] for L in (1024, 224)]
    assert got == g["preprocess_shapes_1024_224"]
    frame = np.array(g["preprocess_small_frame"], dtype=np.uint8)
    want = torch.tensor(g["preprocess_small_out"], dtype=torch.float32)
    assert want.shape == (3, 32, 32)
    assert torch.equal(PP.sam_preprocess(frame, 32), want) and torch.equal(O.sam_preprocess(frame, 32), want)


def test_metric_definitions_against_the_reference_own_functions():
    """The metric's IoU (|A and B| / |A or B|, 0 for an empty union) and IoCM as the reference's OWN calculate_iou /
    calculate_iocm compute them (train_ds.py:761-800, evaluated by oracle/make_golden.py::host_goldens on seven mask pairs incl.
    empty ones) against the product's (train_ds.py validation, evaluation.py scorer), and AverageMeter's arithmetic / log format
    against utils/utils.py's class."""
    from haff import evaluation as EV, train_ds as TD
    with open(os.path.join(GOLD, "host_helpers.json")) as f:
        g = json.load(f)
    rng = np.random.default_rng(78)
    for case in g["iou_cases"]:
        pa, pb = case["p"]
        a, b = rng.random((24, 31)) < pa, rng.random((24, 31)) < pb
        for mod in (TD, EV):
            assert mod.calculate_iou(a, b) == pytest.approx(case["iou"], abs=1e-12), (mod.__name__, case)
            assert mod.calculate_iocm(a, b) == pytest.approx(case["iocm"], abs=1e-12), (mod.__name__, case)
    m = TD.AverageMeter("MaskLoss", ":.4f")
    for v, n_ in ((0.5, 1), (0.25, 3), (1.0 / 3.0, 2)):
        m.update(v, n_)
    ref = g["average_meter"]
    assert str(m) == ref["str"] and m.avg == pytest.approx(ref["avg"], abs=1e-15) and m.count == ref["count"]
    # the 2HANDS question / answer templates (utils/aff_dataset.py:27-46)
    from haff import aff_dataset as AD
    assert list(AD.SHORT_QUESTION_LIST) == g["aff_templates"]["short_question_list"]
    assert list(AD.ANSWER_LIST) == g["aff_templates"]["answer_list"]


@pytest.mark.parametrize("conv_type", ["llava_v1", "llava_llama_2"])
def test_collate_fn_against_the_reference_own_function(conv_type):
    """train_ds.collate_fn against the reference's OWN collate_fn (utils/dataset.py:30-169, evaluated by oracle/make_golden.py::
    host_goldens with its own conversation_lib / tokenizer_image_token in scope) on the same samples and the same stand-in tokenizer,
    under both --conv_type values: the <im_start><image><im_end> replacement, padded ids, the label mask of every round's instruction
    span, attention masks, offsets, the conversation strings and the dict's key set — all exact."""
    from golden_cases import StubSpTokenizer, collate_samples
    from haff import prompt as P, train_ds as TD
    with open(os.path.join(GOLD, "host_helpers.json")) as f:
        g = json.load(f)["collate_fn"][conv_type]
    batch = collate_samples(lambda: P.get_conv(conv_type))
    out = TD.collate_fn(batch, StubSpTokenizer(), model_max_length=StubSpTokenizer.model_max_length, use_mm_start_end=True, conv_type=conv_type)
    assert out["conversation_list"] == g["conversation_list"]
    assert out["input_ids"].tolist() == g["input_ids"] and out["labels"].tolist() == g["labels"]
    assert out["attention_masks"].int().tolist() == g["attention_masks"] and out["offset"].tolist() == g["offset"]
    assert out["taxonomies_list"].tolist() == g["taxonomies_list"] and [list(r) for r in out["resize_list"]] == g["resize_list"]
    assert bool(out["inference"]) == g["inference"] and sorted(out.keys()) == g["keys"]
    assert float(out["images"].double().sum()) == pytest.approx(g["images_sum"], abs=1e-9)
    assert any(v != -100 for v in g["labels"][0]) and g["labels"][0][0] == -100          # answers are labelled, BOS is not


def test_collate_label_mask_follows_conv_type():
    """utils/dataset.py:95-128 under both --conv_type values: BOS and every round's instruction span (up to and including the
    separator: " ASSISTANT: " for llava_v1, "[/INST] " for llava_llama_2, with the reference's -2 correction) are -100, the answer
    tokens keep their ids, padding is -100 — checked against an independent character-level reconstruction with a 1-char-1-token
    tokenizer (so spans can be located by string search)."""
    import torch
    from haff import prompt as P
    from haff import train_ds as TD

    class CharTok:
        bos_token_id, pad_token_id = 1, 0

        def __call__(self, text):
            class R:
                pass
            r = R()
            r.input_ids = [1] + [3 + ord(ch) for ch in text]
            return r

    def sample(conv, rounds):
        for q, a in rounds:
            conv.append_message(conv.roles[0], q)
            conv.append_message(conv.roles[1], a)
        z = torch.zeros(1, 8, 8)
        return (None, torch.zeros(3, 8, 8), torch.zeros(3, 8, 8), [conv.get_prompt()], z, z, [1.0, 0, 0, 0], {"left": z[0], "right": z[0]},
                (8, 8), None, None, False)
    for name in ("llava_v1", "llava_llama_2"):
        rounds_a = [("<image>\nwhere to hold the mug?", "Sure, [SEG]."), ("and the pan?", "It is [SEG].")]
        rounds_b = [("<image>\nshort", "[SEG].")]
        batch = [sample(P.get_conv(name), rounds_a), sample(P.get_conv(name), rounds_b)]
        out = TD.collate_fn(batch, CharTok(), model_max_length=100000, use_mm_start_end=False, conv_type=name)
        ids, labels = out["input_ids"], out["labels"]
        sep = " ASSISTANT: " if name == "llava_v1" else "[/INST] "
        for row, text in enumerate(out["conversation_list"]):
            # token t of the row <-> character: BOS, then one id per character, "<image>" collapsed into the single -200
            chars = [None]
            i = 0
            while i < len(text):
                if text.startswith("<image>", i):
                    chars.append("<image>")
                    i += 7
                else:
                    chars.append(text[i])
                    i += 1
            n = len(chars)
            assert int((ids[row] != 0).sum()) == n
            want = torch.full((ids.shape[1],), -100, dtype=torch.long)
            # answers: from the end of each separator to the end of the round (incl. sep2's characters); the reference's "- 2"
            # on the instruction length (written for sentencepiece, where BOS and the separator's trailing space do not count)
            # lets the mask stop two tokens early with a 1-char-1-token tokenizer: the last two separator characters stay labelled
            pos, tok = 0, 1
            for rou in text.split("</s>"):
                if rou == "":
                    break
                k = rou.index(sep) + len(sep)
                n_rou = len(rou.replace("<image>", "I"))
                n_ins = len(rou[:k].replace("<image>", "I"))
                # reference arithmetic: round_len = n_rou + 1 (BOS), instruction_len = n_ins + 1 - 2
                want[tok + n_ins - 1:tok + n_rou + 1] = ids[row][tok + n_ins - 1:tok + n_rou + 1]
                tok += n_rou + 1
            want[tok:] = -100
            assert torch.equal(labels[row], want), (name, row)
            assert (labels[row][:1] == -100).all() and (labels[row][n:] == -100).all()


def test_bf16_points_per_stack_and_memo():
    """Test infrastructure of round 5 (bench.parity_full_frame's attribution): lisa_evaluate(points=, memo=) switches the
    bf16-points mode per stack and re-uses stage results. points=None is the exact forward bit for bit (memo or not); all three
    stacks == `with bf16_points():` bit for bit; a single stack's points change the result, by less than all three together do in
    the stack that dominates; and a memo filled by one call serves the next without recomputing the shared stage."""
    import numpy as np
    import haff  # noqa: F401
    from haff import config as hcfg, weights as hw
    from oracle import lisa_oracle as O
    cfg = hcfg.tiny()
    sd = hw.round_to_bf16_(hw.make_state_dict(cfg, 3))
    rng = np.random.default_rng(3)
    S = cfg.sam.img_size
    images = torch.from_numpy(rng.standard_normal((1, 3, S, S), dtype=np.float32))
    clip = torch.from_numpy(rng.standard_normal((1, 3, 224, 224), dtype=np.float32))
    ids = torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx, 11, 12, 13, 14]])
    forced = torch.tensor([[7, cfg.seg_token_idx, 9, cfg.eos_token_id]])
    kw = dict(max_new_tokens=4, forced_answer=forced, use_cache=True)
    sz = [(S, S)]
    with torch.no_grad():
        exact = O.lisa_evaluate(sd, cfg, clip, images, ids, sz, sz, **kw)
        memo = {}
        again = O.lisa_evaluate(sd, cfg, clip, images, ids, sz, sz, memo=memo, points=(), **kw)
        assert torch.equal(exact[1][0], again[1][0]) and torch.equal(exact[2][0], again[2][0]) and len(memo) == 2
        with O.bf16_points():
            allp = O.lisa_evaluate(sd, cfg, clip, images, ids, sz, sz, **kw)
        allp2 = O.lisa_evaluate(sd, cfg, clip, images, ids, sz, sz, memo=memo, points=("sam", "clip", "llama"), **kw)
        assert torch.equal(allp[1][0], allp2[1][0]) and torch.equal(allp[2][0], allp2[2][0])
        calls = {"sam": 0}
        real = O.sam_image_encoder

        def counting(*a, **k):
            calls["sam"] += 1
            return real(*a, **k)
        O.sam_image_encoder = counting
        try:
            sam_only = O.lisa_evaluate(sd, cfg, clip, images, ids, sz, sz, memo=memo, points=("sam",), **kw)
            llama_only = O.lisa_evaluate(sd, cfg, clip, images, ids, sz, sz, memo=memo, points=("llama",), **kw)
        finally:
            O.sam_image_encoder = real
        assert calls["sam"] == 0, "both image embeddings were in the memo already (exact and bf16-points)"
    d = lambda a: (a[1][0] - exact[1][0]).abs().max().item()   # noqa: E731
    assert 0 < d(sam_only) and 0 < d(llama_only) and d(allp) > 0
    assert max(d(sam_only), d(llama_only)) <= 2.0 * d(allp) + 1e-6
