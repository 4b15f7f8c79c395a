"""CPU-side checks: the C-ABI library builds/loads here (hipcc cross-compiles gfx950 without a GPU) and exports
exactly the entry points include/haff_hip.h declares; host-side helpers behave; nothing computes on a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import haff
from haff import config as hcfg
from haff import flops, preprocess, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "haff_hip.h")).read()
    return sorted(set(re.findall(r"^int (haff_\w+)\(", text, flags=re.M)))


def test_header_matches_python_prototypes():
    assert _declared() == sorted(haff.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(haff.LIB_PATH):
        haff.build_library()
    lib = ctypes.CDLL(haff.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    haff.load_library()


def test_stream_caps_are_per_stream_and_thread_safe():
    """haff_gemm_stream_cap keeps ONE table keyed by stream (no process-wide setting: include/haff_hip.h, conventions): a cap set for
    one stream is invisible to another, 256 releases the entry, invalid values are queries, the 33rd simultaneously capped stream is
    refused, and host threads hammering their own streams never see each other's values. Touches no GPU (nothing is launched)."""
    import threading
    lib = haff.load_library()
    cap = lambda s, c: int(lib.haff_gemm_stream_cap(ctypes.c_void_p(s), c))   # noqa: E731
    A, B = 0x1000, 0x2000
    assert cap(A, 0) == 256 and cap(B, 0) == 256 and cap(0, 0) == 256
    assert cap(A, 224) == 256 and cap(B, 128) == 256
    assert cap(A, 0) == 224 and cap(B, 0) == 128 and cap(0, 0) == 256        # the null stream is a stream of its own
    for bad in (7, 100, 260, -8):
        assert cap(A, bad) == 224                                             # not a multiple of 8 in 8..256: a query
    assert cap(A, 192) == 224 and cap(B, 0) == 128
    assert cap(A, 256) == 192 and cap(A, 0) == 256 and cap(B, 0) == 128       # released
    assert cap(B, 256) == 128
    keys = [0x10000 + 64 * i for i in range(32)]
    assert all(cap(k, 64) == 256 for k in keys)
    assert cap(0x99999, 64) == -2                                             # table full: refused, nothing changed
    assert all(cap(k, 256) == 64 for k in keys) and cap(0x99999, 0) == 256
    bad = []

    def worker(key, mine):
        for _ in range(2000):
            if cap(key, mine) not in (256, mine):
                bad.append(key)
            if cap(key, 0) != mine:
                bad.append(key)
            if cap(key, 256) != mine:
                bad.append(key)
    ts = [threading.Thread(target=worker, args=(0x5000 + 8 * i, 8 * (i + 1))) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


def test_decode_chain_geometry_checks_run_on_the_host():
    """haff_decode_chain_supported / _sync_words are host arithmetic (no launch): the chained decode step takes <= 8 rows, head dim 128,
    hidden and ffn multiples of 256 (its K halves are whole 32-deep k-steps per wave), <= 48 layers; the counter buffer holds 16 lines
    per (layer, stage), one error line, one ticket per K-split tile. haff_decode_chain_bf16 refuses bad arguments before touching the GPU."""
    lib = haff.load_library()
    sup = lambda *a: int(lib.haff_decode_chain_supported(*a))   # noqa: E731
    assert sup(1, 4096, 11008, 32, 32) == 3 * 256 + 32 + 512 + 688 + 512        # 7B, one row: workgroups per layer
    assert sup(8, 5120, 13824, 40, 40) == 3 * 320 + 8 * 40 + 640 + 864 + 640    # 13B, 8 rows
    assert sup(9, 4096, 11008, 32, 32) == 0 and sup(0, 4096, 11008, 32, 32) == 0
    assert sup(8, 4096, 11008, 64, 32) == 0 and sup(8, 4096, 11136, 32, 32) == 0 and sup(8, 4224, 11008, 33, 32) == 0
    assert sup(8, 4096, 11008, 32, 49) == 0 and sup(8, 16384, 11008, 128, 4) == 0
    words = int(lib.haff_decode_chain_sync_words(32, 4096))
    assert words == 32 * 5 * 16 * 32 + 32 + 32 * 2 * 256 and int(lib.haff_decode_chain_sync_words(0, 4096)) == 0
    assert int(lib.haff_decode_chain_bf16(None, 32, 1, 4096, 11008, 32, *([None] * 8), 1e-5, None, None, 299, 0.088, None, 0, None)) == -1
    assert int(lib.haff_decode_chain_status(None, 32, None)) == -1


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from haff import ops
    from haff.lisa import LisaMI355
    with pytest.raises(RuntimeError):
        LisaMI355(hcfg.tiny(), {}, device="cuda:0")
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(8, 8), torch.zeros(8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "2handedafforder_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle"


def test_weight_filler_is_deterministic_and_complete():
    cfg = hcfg.tiny()
    a = weights.make_state_dict(cfg, 7)
    b = weights.make_state_dict(cfg, 7)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    c = weights.make_state_dict(cfg, 8)
    assert not torch.equal(a["lm_head.weight"], c["lm_head.weight"])
    sub = weights.make_state_dict(cfg, 7, weights.clip_shapes(cfg.clip))
    assert set(sub) == set(weights.clip_shapes(cfg.clip))
    assert a["model.visual_model.image_encoder.blocks.0.attn.rel_pos_h"].shape == (13, 32)
    assert a["model.visual_model.image_encoder.blocks.1.attn.rel_pos_h"].shape == (27, 32)


def test_flop_model_matches_survey():
    assert abs(flops.frame_flops(hcfg.haff_7b())["total"] / 10.01e12 - 1) < 5e-3
    assert abs(flops.frame_flops(hcfg.haff_13b())["total"] / 13.73e12 - 1) < 5e-3
    assert abs(flops.frame_flops(hcfg.haff_7b())["sam_encoder"] / 5.961e12 - 1) < 1e-3


def test_preprocess_contracts():
    rng = np.random.default_rng(0)
    fr = rng.integers(0, 256, size=(50, 64, 3), dtype=np.uint8)
    x = preprocess.sam_preprocess(torch.from_numpy(fr), 64)
    assert x.shape == (3, 64, 64) and torch.all(x[:, 50:] == 0)
    ref = (torch.from_numpy(fr).permute(2, 0, 1).float() - torch.tensor(preprocess.SAM_MEAN).view(3, 1, 1)) / torch.tensor(preprocess.SAM_STD).view(3, 1, 1)
    assert torch.equal(x[:, :50], ref)
    assert preprocess.get_preprocess_shape(480, 640, 1024) == (768, 1024)
    assert preprocess.clip_preprocess(torch.from_numpy(fr)).shape == (3, 224, 224)


def test_sigmoid_thresholds_as_exact_fp32_logit_thresholds():
    """a15: `sigmoid(mask) > th` (fp32 sigmoid, float32(th): inference.py:294-301) == `mask > x*(th)` for the ONE fp32
    constant postprocess.py hard-codes per threshold. Re-derive each constant by bisection against torch.sigmoid, then
    check the equivalence on every float within 64 ulps of x*, on a dense sweep and on random logits."""
    from haff import postprocess as P
    for th in P.THRESHOLDS:
        xs = P.sigmoid_logit_threshold(th)
        assert xs == P.derive_sigmoid_logit_threshold(th), th
        th32 = torch.tensor(th, dtype=torch.float32)
        near = [torch.tensor(xs, dtype=torch.float32)]
        for _ in range(64):
            near.append(torch.nextafter(near[-1], torch.tensor(float("inf"))))
        lo = torch.tensor(xs, dtype=torch.float32)
        for _ in range(64):
            lo = torch.nextafter(lo, torch.tensor(float("-inf")))
            near.append(lo)
        x = torch.cat([torch.stack(near), torch.linspace(-30, 30, 200001), torch.randn(1 << 20, generator=torch.Generator().manual_seed(1)) * 3])
        assert torch.equal(torch.sigmoid(x) > th32, x > xs), th
    assert P.sigmoid_logit_threshold(0.5) > 0.0   # sigmoid_f32 rounds to exactly 0.5 just above zero: not the chat rule


def test_sentencepiece_tokenizer_wrapper(tmp_path):
    """checkpoint.SentencePieceTokenizer (the reference's slow Llama tokenizer + 3 added tokens, inference.py:115-127)
    on a tiny sentencepiece model trained here: id layout, BOS handling, added-token splitting, the image-token
    splice of llava/mm_utils.py:19-44 and a decode round trip."""
    import sentencepiece as spm
    from haff import prompt as hprompt
    from haff.checkpoint import SentencePieceTokenizer
    corpus = tmp_path / "c.txt"
    corpus.write_text("\n".join(["where would someone grasp the cup to pour water", "cut the bread with the knife",
                                  "A chat between a curious human and an artificial intelligence assistant.",
                                  "USER: can you segment the affordance ASSISTANT: Sure, it is ."] * 20))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "tok"), vocab_size=320,
                                   model_type="bpe", bos_id=1, eos_id=2, unk_id=0, pad_id=-1, byte_fallback=True,
                                   character_coverage=1.0, minloglevel=2)
    tok = SentencePieceTokenizer(str(tmp_path / "tok.model"))
    sp = spm.SentencePieceProcessor(model_file=str(tmp_path / "tok.model"))
    n = sp.get_piece_size()
    assert (tok.special["[SEG]"], tok.special["<im_start>"], tok.special["<im_end>"]) == (n, n + 1, n + 2)
    assert len(tok) == n + 3 and tok.bos_token_id == 1 and tok.pad_token_id == tok.unk_token_id == 0
    ids = tok("cut the bread [SEG] with<im_end>").input_ids
    assert ids[0] == 1 and ids.count(n) == 1 and ids[-1] == n + 2
    assert ids == [1] + sp.encode("cut the bread ") + [n] + sp.encode(" with") + [n + 2]
    assert tok("cut", add_special_tokens=False).input_ids == sp.encode("cut")
    text = "USER: <im_start><image><im_end>\ncut the bread ASSISTANT: Sure, it is [SEG]."
    got = hprompt.tokenizer_image_token(text, tok)
    assert got.count(-200) == 1 and got[0] == 1 and got.count(1) == 1   # one BOS, one image sentinel
    i = got.index(-200)
    assert got[i - 1] == n + 1 and got[i + 1] == n + 2
    assert tok.decode(tok("cut the bread").input_ids, skip_special_tokens=True).strip() == "cut the bread"
    assert "[SEG]" in tok.decode(ids)


def test_evaluation_harness(tmp_path):
    """haff.evaluation on a hand-made benchmark/prediction tree: closed-form IoU / IoCM / Hausdorff values, the
    missing-hand rule, the threshold sweep (calculate_iou.py:26-41,97-114,243-261,321-343)."""
    from PIL import Image
    from haff import evaluation as ev
    a = np.zeros((10, 10), bool); a[2:6, 2:6] = True          # 16 px
    b = np.zeros((10, 10), bool); b[4:8, 4:8] = True          # 16 px, 4 px overlap
    assert ev.calculate_iou(a, b) == 4 / 28 and ev.calculate_iocm(a, b) == 4 / 16
    assert ev.calculate_iou(a, np.zeros((0, 0))) is None and ev.calculate_iou(np.zeros_like(a), np.zeros_like(a)) == 0.0
    dhd, hd = ev.calculate_hausdorff(a, b)
    assert abs(dhd - np.sqrt(8)) < 1e-9 and abs(hd - np.sqrt(8)) < 1e-9      # corner (7,7) to nearest point (5,5)
    assert ev.calculate_hausdorff(a, np.zeros_like(a)) == (np.sqrt(200), np.sqrt(200))
    assert ev.calculate_hausdorff(np.zeros_like(a), a) == (0.0, 0.0)

    def png(path, m):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        Image.fromarray(m.astype(np.uint8) * 255).save(path)
    bench, comp = tmp_path / "bench", tmp_path / "pred"
    png(str(bench / "vid" / "0001" / "inpainting.png"), np.zeros((10, 10), bool))
    png(str(bench / "vid" / "0001" / "aff_left.png"), a)
    png(str(bench / "vid" / "0001" / "aff_right.png"), b)
    for th, shrink in (("0.3", 0), ("0.7", 1)):
        pl = np.zeros((10, 10), bool); pl[2 + shrink:6, 2 + shrink:6] = True
        png(str(comp / th / "vid" / "0001" / "aff_left.png"), pl)          # right hand missing -> counted as empty
    res = ev.evaluate_folders(str(bench), str(comp), calc_map=True, is_cropped=True, verbose=False)
    r3, r7 = res["per_threshold"]
    union = a | b
    assert r3["count"] == 1 and abs(r3["iou"] - 16 / union.sum()) < 1e-9 and r3["iocm"] == 1.0
    assert abs(r7["iou"] - 9 / union.sum()) < 1e-9 and r7["iocm"] == 1.0
    assert res["best"]["threshold"] == "0.3" and res["mean_average_precision"] == 1.0


def test_merge_lora_and_reload(tmp_path):
    """merge_lora.py (merge_lora_weights_and_save_hf_model.py:146-155): W += (alpha/r) B A on q/v_proj, trained tensors
    override the base, vision_tower keys dropped, sharded safetensors + index + config read back by checkpoint.py."""
    from haff import checkpoint, merge_lora
    cfg = hcfg.tiny()
    base = weights.make_state_dict(cfg, 3)
    g = torch.Generator().manual_seed(4)
    r, alpha, H = 4, 8, cfg.llm.hidden
    trained = {"model.embed_tokens.weight": torch.randn(base["model.embed_tokens.weight"].shape, generator=g),
               "lm_head.weight": torch.randn(base["lm_head.weight"].shape, generator=g)}
    for i in range(cfg.llm.layers):
        for n in ("q_proj", "v_proj"):
            trained[f"model.layers.{i}.self_attn.{n}.lora_A"] = torch.randn((r, H), generator=g) * 0.1
            trained[f"model.layers.{i}.self_attn.{n}.lora_B"] = torch.randn((H, r), generator=g) * 0.1
    merged = merge_lora.merge_state_dict(base, trained, r, alpha, torch.float32)
    assert not any("lora_" in k or "vision_tower" in k for k in merged)
    k = "model.layers.1.self_attn.v_proj"
    want = base[k + ".weight"] + (alpha / r) * trained[k + ".lora_B"] @ trained[k + ".lora_A"]
    assert torch.allclose(merged[k + ".weight"], want, atol=1e-6)
    assert torch.equal(merged["model.layers.1.self_attn.k_proj.weight"], base["model.layers.1.self_attn.k_proj.weight"])
    assert torch.equal(merged["lm_head.weight"], trained["lm_head.weight"])
    files = merge_lora.save_pretrained(merged, str(tmp_path / "out"), merge_lora.hf_config(cfg, torch.float32),
                                       max_shard_bytes=4 << 20)
    assert len(files) > 1                                   # sharding exercised
    back = checkpoint.load_hf_dir(str(tmp_path / "out"))
    assert set(back) == set(merged) and all(torch.equal(back[kk], merged[kk]) for kk in merged)
    idx = json.load(open(tmp_path / "out" / "model.safetensors.index.json"))
    assert set(idx["weight_map"]) == set(merged)
    with pytest.raises(ValueError):
        merge_lora.merge_state_dict(base, trained, 8, alpha)


def test_aff_records_dataset_feeds_collate():
    """aff_dataset.AffRecordsDataset (aff_dataset.py:198-280 on HF-layout records) -> train_ds.collate_fn: contour masks,
    templates, preprocessing shapes, taxonomy vectors, [SEG] in the supervised span."""
    from haff import aff_dataset, checkpoint, train_ds
    cfg = hcfg.tiny()
    rng = np.random.default_rng(0)
    sq = [[10, 10], [30, 10], [30, 25], [10, 25]]
    recs = [{"narration": b"Cut The Bread", "inpainted": rng.integers(0, 255, (48, 64, 3), dtype=np.uint8), "taxonomy": [0, 0, 1, 0],
             "masks": {"aff_left": [sq], "aff_right": [], "original_size": (48, 64)}},
            {"text": "open bottle", "image": rng.integers(0, 255, (48, 64, 3), dtype=np.uint8), "taxonomy": 1,
             "masks": {"aff_left": [], "aff_right": [[[40, 5], [60, 5], [50, 40]]], "original_size": (48, 64)}}]
    m = aff_dataset.recreate_mask_from_contours([sq], (48, 64))
    assert m.shape == (48, 64) and m[10:26, 10:31].all() and m.sum() == 16 * 21 and m.dtype == np.uint8
    ds = aff_dataset.AffRecordsDataset(recs, cfg, samples_per_epoch=7, seed=3)
    assert len(ds) == 7
    items = [ds[i] for i in range(4)]
    S = cfg.sam.img_size
    for it in items:
        _, image, clip, convs, ml, mr, tax, label, resize, questions, classes, inference = it
        assert image.shape == (3, S, S) and clip.shape == (3, cfg.clip.image, cfg.clip.image)
        assert ml.shape == (1, 48, 64) and mr.shape == (1, 48, 64) and len(tax) == 4 and abs(sum(tax) - 1) < 1e-9
        assert max(resize) == S and "[SEG]" in convs[0] and "<image>" in convs[0] and classes[0].lower() in questions[0]
        assert set(torch.unique(label["left"]).tolist()) <= {0, 255} and inference is False
    tok = checkpoint.ByteTokenizer(cfg)
    batch = train_ds.collate_fn(items, tok)
    assert batch["images"].shape == (4, 3, S, S) and batch["taxonomies_list"].shape == (4, 4)
    assert batch["offset"].tolist() == [0, 1, 2, 3, 4] and (batch["labels"] != -100).any()
