"""Row f1 against files written by the toolchains the reference uses (fixtures made by oracle/make_golden_files.py in the build
container): transformers' save_pretrained for the Llama and CLIP directories, the reference's own Sam module for the SAM
key / shape / content manifest, transformers' LlamaTokenizer + sentencepiece for the token ids. The loaders had only ever read
this repo's own writers before (round-2 finding: circular)."""
import hashlib
import json
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 41


def _cfg_sd():
    import haff  # noqa: F401
    from haff import config as hcfg, weights as hw
    cfg = hcfg.tiny()
    return cfg, hw.make_state_dict(cfg, SEED)


def test_llama_directory_written_by_transformers_loads_bit_exact():
    """Sharded safetensors + index + config.json from LlamaForCausalLM.save_pretrained (what
    merge_lora_weights_and_save_hf_model.py:146-155 calls): every tensor comes back under the reference's key and equals the
    seeded tensor that was loaded into the HF module; config.json (HF's field names) parses to the geometry."""
    from haff import checkpoint
    cfg, sd = _cfg_sd()
    d = os.path.join(GOLD, "hf_llama_tiny")
    files = sorted(os.listdir(d))
    assert "model.safetensors.index.json" in files and sum(f.endswith(".safetensors") for f in files) >= 2   # really sharded
    got = checkpoint.load_hf_dir(d)
    want = {k: v for k, v in sd.items() if k.startswith("model.layers.") or k in ("model.norm.weight", "model.embed_tokens.weight", "lm_head.weight")}
    assert set(want) <= set(got), sorted(set(want) - set(got))[:4]
    assert all("rotary" in k for k in set(got) - set(want)), sorted(set(got) - set(want))[:4]
    for k, v in want.items():
        assert got[k].dtype == torch.float32 and torch.equal(got[k], v), k
    c = checkpoint.config_from_dir(d)
    l = c.llm
    assert (l.hidden, l.layers, l.heads, l.ffn, l.rms_eps, l.rope_theta) == (cfg.llm.hidden, cfg.llm.layers, cfg.llm.heads,
                                                                             cfg.llm.ffn, cfg.llm.rms_eps, cfg.llm.rope_theta)
    # HF wrote vocab_size = 323 (the fixture's embedding already carries the three added rows: 323 % 1000 != 3, so the modulo
    # heuristic alone would add three more) — the caller's tokenizer length decides (train_ds.py), see load_state_dict(cfg=)
    assert json.load(open(os.path.join(d, "config.json")))["vocab_size"] == cfg.llm.vocab
    assert (c.bos_token_id, c.eos_token_id, c.pad_token_id) == (1, 2, 0)


@pytest.mark.parametrize("layout", ["hf_clip_tiny", "hf_clip_tiny_v5_bare"])
def test_clip_directory_written_by_transformers_loads_bit_exact(layout, tmp_path):
    """CLIPModel.save_pretrained (hub layout: the vision tower under vision_model.*, a text tower beside it) and a bare
    CLIPVisionModel as transformers 5.x writes it (no prefix): the vision tower lands under the reference's
    model.vision_tower.vision_tower.vision_model.* keys, bit-equal; nothing else leaks in."""
    from haff import checkpoint
    cfg, sd = _cfg_sd()
    n0 = len(checkpoint.load_hf_dir(os.path.join(GOLD, "hf_llama_tiny")))   # the base the CLIP directory is attached to
    with pytest.raises(KeyError):   # a plain Llama directory has no visual_model / text_hidden_fcs / projector: refused loudly
        checkpoint.load_state_dict(os.path.join(GOLD, "hf_llama_tiny"), os.path.join(GOLD, layout), cfg=cfg)
    orig = checkpoint.hw.all_shapes
    checkpoint.hw.all_shapes = lambda c: {}          # skip the completeness check: this test looks at the CLIP mapping only
    try:
        got = checkpoint.load_state_dict(os.path.join(GOLD, "hf_llama_tiny"), os.path.join(GOLD, layout), cfg=cfg)
    finally:
        checkpoint.hw.all_shapes = orig
    pfx = "model.vision_tower.vision_tower."
    want = {k: v for k, v in sd.items() if k.startswith(pfx)}
    clip_keys = {k for k in got if k.startswith(pfx)}
    assert set(want) <= clip_keys
    assert all("post_layernorm" in k for k in clip_keys - set(want)), sorted(clip_keys - set(want))[:4]
    for k, v in want.items():
        assert torch.equal(got[k], v), k
    assert not any("text_model" in k or "projection" in k for k in got)
    assert len(got) == n0 + len(clip_keys)


def test_sam_inventory_matches_the_reference_module():
    """sam_ref_manifest.json = state_dict() of the reference's own Sam (build_sam.py arguments, tiny geometry) after
    load_state_dict of the seeded tensors: every key this repo carries exists there with the same shape and the same bytes; the
    reference has 15 more (point / box / mask prompt embeddings, which the text-prompt path never touches); and a SAM .pth in
    the ORIGINAL layout (one `mask_decoder.*`) is duplicated left / right by the loader as build_sam.py:125-136 does."""
    from haff import checkpoint
    cfg, sd = _cfg_sd()
    man = json.load(open(os.path.join(GOLD, "sam_ref_manifest.json")))
    V = "model.visual_model."
    ours = {k[len(V):]: v for k, v in sd.items() if k.startswith(V)}
    ref = man["tensors"]
    assert set(ours) <= set(ref), sorted(set(ours) - set(ref))[:4]
    extra = set(ref) - set(ours)
    assert extra == set(man["keys_not_in_seeded_inventory"]) and len(extra) == 15
    assert all(("point_embeddings" in k or "not_a_point" in k or "mask_downscaling" in k) for k in extra), sorted(extra)
    for k, v in ours.items():
        assert list(v.shape) == ref[k]["shape"], k
        assert hashlib.sha1(v.contiguous().numpy().tobytes()).hexdigest() == ref[k]["sha1"], k


def test_sam_pth_in_the_original_layout_is_duplicated_left_and_right(tmp_path):
    from haff import checkpoint
    cfg, sd = _cfg_sd()
    V = "model.visual_model."
    pth = {}
    for k, v in sd.items():
        if k.startswith(V + "image_encoder.") or k.startswith(V + "prompt_encoder."):
            pth[k[len(V):]] = v
        elif k.startswith(V + "mask_decoder_left.") and "taxonomy_embed" not in k:
            pth["mask_decoder." + k[len(V + "mask_decoder_left."):]] = v
    torch.save(pth, str(tmp_path / "sam.pth"))          # torch.save of a key -> tensor dict: what sam_vit_h_4b8939.pth is
    orig = checkpoint.hw.all_shapes
    checkpoint.hw.all_shapes = lambda c: {}
    try:
        got = checkpoint.load_state_dict(os.path.join(GOLD, "hf_llama_tiny"), None, str(tmp_path / "sam.pth"), cfg=cfg)
    finally:
        checkpoint.hw.all_shapes = orig
    for k, v in sd.items():
        if k.startswith(V + "mask_decoder_left.") and "taxonomy_embed" not in k:
            assert torch.equal(got[k], v) and torch.equal(got[k.replace("_left", "_right")], v), k
        elif k.startswith(V + "image_encoder."):
            assert torch.equal(got[k], v), k


def test_tokenizer_against_transformers_and_sentencepiece():
    """tokenizer_ids.json: prompts tokenised in the build container by transformers 5.15's LlamaTokenizer over the fixture's
    sentencepiece model (its `tokenizers` backend: 5.15 has no slow class) and by sentencepiece itself. On plain text the two
    agree and so must SentencePieceTokenizer (BOS prepended once); the added tokens get the ids transformers assigned
    (appended after the sentencepiece vocabulary in train_ds.py:142-149's order). Where the 5.15 class departs from
    sentencepiece (leading spaces, text after an added token) the wrapper follows the 4.31 slow class's documented legacy
    behaviour (every chunk encoded by sentencepiece on its own) — restated from publication, unpinned, and checked here only
    against sentencepiece."""
    import sentencepiece as spm
    from haff import checkpoint
    d = os.path.join(GOLD, "tokenizer_tiny")
    rec = json.load(open(os.path.join(d, "tokenizer_ids.json")))
    assert rec["backend"] == "TokenizersBackend"          # i.e. NOT a sentencepiece-backed slow tokenizer: say so loudly
    tk = checkpoint.SentencePieceTokenizer(os.path.join(d, "tokenizer.model"))
    sp = spm.SentencePieceProcessor(model_file=os.path.join(d, "tokenizer.model"))
    assert len(tk) == rec["len"] and tk.special == rec["added_token_ids"]
    n_agree = 0
    for p in rec["prompts"]:
        ids = tk(p["text"]).input_ids
        assert ids[0] == tk.bos_token_id == 1
        if p["hf_ids_no_bos"] == p["spm_ids"]:
            n_agree += 1
            assert ids[1:] == p["hf_ids_no_bos"], p["text"]
            assert tk(p["text"], add_special_tokens=False).input_ids == p["hf_ids_no_bos"]
        elif not any(t in p["text"] for t in tk.special):
            assert ids[1:] == p["spm_ids"] == sp.encode(p["text"]), p["text"]
    assert n_agree >= 7
    ids = tk("Sure, [SEG] .").input_ids
    assert ids.count(320) == 1 and tk.decode(ids, skip_special_tokens=True).replace(" ", "") == "Sure,[SEG]."
    ids = tk("<im_start><image><im_end>\nhold the pan").input_ids
    assert ids[1] == 321 and 322 in ids


def test_added_token_ids_come_from_the_tokenizer_files(tmp_path):
    """from_pretrained's special-token rule: added_tokens.json first, else base + 3 rows behind the sentencepiece
    vocabulary, else (no tokenizer files) config.json's vocab_size or vocab_size + 3 rows; a plain base or a padded vocabulary is
    refused instead of reading ordinary / padding rows as [SEG]. The reference reads the ids from the tokenizer
    (inference.py:115-131, train_ds.py:142-149)."""
    import shutil
    import pytest
    import haff  # noqa: F401
    from haff import checkpoint
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"vocab_size": 32000, "hidden_size": 4096}))
    # no tokenizer files: the synthetic directories of this repo
    assert checkpoint.resolve_added_tokens(str(d), 32003) == {"[SEG]": 32000, "<im_start>": 32001, "<im_end>": 32002}
    # rows == vocab_size: only where something says the three rows are counted in it — this repo's marker in config.json
    # (merge_lora.py) or a caller that passed seg_token_idx; a plain base-Llama directory is refused
    with pytest.raises(ValueError, match="cannot tell"):
        checkpoint.resolve_added_tokens(str(d), 32000)
    assert checkpoint.resolve_added_tokens(str(d), 32000, layout_asserted=True)["[SEG]"] == 31997
    (d / "config.json").write_text(json.dumps({"vocab_size": 32000, "hidden_size": 4096, "haff_vocab_includes_added_tokens": True}))
    assert checkpoint.resolve_added_tokens(str(d), 32000)["[SEG]"] == 31997
    (d / "config.json").write_text(json.dumps({"vocab_size": 32000, "hidden_size": 4096}))
    with pytest.raises(ValueError):
        checkpoint.resolve_added_tokens(str(d), 32064)
    # a sentencepiece model: base + 3 or nothing
    tok = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tokenizer_tiny", "tokenizer.model")
    shutil.copy(tok, d / "tokenizer.model")
    base = checkpoint.sentencepiece_vocab_size(str(d))
    assert base and base > 100
    assert checkpoint.resolve_added_tokens(str(d), base + 3) == {"[SEG]": base, "<im_start>": base + 1, "<im_end>": base + 2}
    for rows in (base, base + 64, 32003):
        with pytest.raises(ValueError, match="cannot tell"):
            checkpoint.resolve_added_tokens(str(d), rows)
    # added_tokens.json is authoritative, wherever the rows sit
    (d / "added_tokens.json").write_text(json.dumps({"<im_end>": base + 5, "<im_start>": base + 4, "[SEG]": base + 1, "<extra>": base}))
    assert checkpoint.resolve_added_tokens(str(d), base + 64) == {"[SEG]": base + 1, "<im_start>": base + 4, "<im_end>": base + 5}
    with pytest.raises(ValueError, match="do not fit"):
        checkpoint.resolve_added_tokens(str(d), base + 5)
