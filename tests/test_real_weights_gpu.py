"""f1 end to end, with real FILES in the reference's layouts (reduced geometry, weights from the seeded filler):

  plain LLaVA base dir (sharded safetensors + config.json + sentencepiece tokenizer.model, vocab WITHOUT the added tokens,
  no visual_model / text_hidden_fcs)  +  CLIP dir (`vision_model.*` keys)  +  SAM .pth (`image_encoder.*`,
  `prompt_encoder.*`, `mask_decoder.*`, build_sam.py:125-136)  +  2HANDS records
      -> train_ds.py (real flags: --version --vision-tower --vision_pretrained; train_ds.py:135-244)
      -> merge_lora.py (merge_lora_weights_and_save_hf_model.py:91-155)
      -> LisaMI355.from_pretrained(merged, vision_tower=...) (inference.py:158-168) and the inference CLI on it."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_inputs(tmp_path, cfg, sd):
    import sentencepiece as spm
    from safetensors.torch import save_file
    from haff import merge_lora
    base, clip, sam = tmp_path / "llava_base", tmp_path / "clip", tmp_path / "sam.pth"
    base.mkdir()
    clip.mkdir()
    n_base = cfg.llm.vocab - 3
    corpus = tmp_path / "corpus.txt"
    words = ("where would someone grasp the cup to pour water cut bread with knife open drawer bottle hold pan stir pot "
             "please segment region perform action image can you show me interact objects following task sure it is result").split()
    rng = np.random.default_rng(0)
    filler = ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), size=rng.integers(3, 9))) for _ in range(400)]
    corpus.write_text("\n".join(" ".join(rng.choice(words + filler, size=12)) for _ in range(600)))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(base / "tokenizer"), vocab_size=n_base,
                                   model_type="bpe", bos_id=1, eos_id=2, unk_id=0, pad_id=-1, character_coverage=1.0,
                                   minloglevel=2)
    assert spm.SentencePieceProcessor(model_file=str(base / "tokenizer.model")).get_piece_size() == n_base
    llm = {k: v.clone() for k, v in sd.items() if k.startswith("model.layers.") or k in ("model.norm.weight",) or "mm_projector" in k}
    llm["model.embed_tokens.weight"] = sd["model.embed_tokens.weight"][:n_base].clone()
    llm["lm_head.weight"] = sd["lm_head.weight"][:n_base].clone()
    hf = merge_lora.hf_config(cfg, torch.float32)
    hf["vocab_size"] = n_base
    hf["haff_vocab_includes_added_tokens"] = False
    merge_lora.save_pretrained({k: v.contiguous() for k, v in llm.items()}, str(base), hf, max_shard_bytes=1 << 20)
    pfx = "model.vision_tower.vision_tower."
    save_file({k[len(pfx):]: v.contiguous() for k, v in sd.items() if k.startswith(pfx)}, str(clip / "model.safetensors"))
    V = "model.visual_model."
    sam_sd = {}
    for k, v in sd.items():
        if k.startswith(V + "image_encoder.") or k.startswith(V + "prompt_encoder."):
            sam_sd[k[len(V):]] = v.clone()
        elif k.startswith(V + "mask_decoder_left.") and "taxonomy_embed" not in k:
            sam_sd["mask_decoder." + k[len(V + "mask_decoder_left."):]] = v.clone()
    torch.save(sam_sd, str(sam))
    rng = np.random.default_rng(0)
    recs = []
    for i in range(4):
        x0, y0 = 10 + 5 * i, 8 + 3 * i
        recs.append({"narration": ["cut the bread", "open the drawer", "pour water", "hold the pan"][i],
                     "inpainted": rng.integers(0, 255, (96, 128, 3), dtype=np.uint8), "taxonomy": [0, 0, 1, 0],
                     "masks": {"aff_left": [[[x0, y0], [x0 + 30, y0], [x0 + 30, y0 + 20], [x0, y0 + 20]]],
                               "aff_right": [[[80, 40], [110, 40], [95, 70]]], "original_size": (96, 128)}})
    torch.save(recs, str(tmp_path / "records.pt"))
    return base, clip, sam, tmp_path / "records.pt"


def test_train_merge_serve_on_checkpoint_files(dev, tmp_path, capsys, monkeypatch):
    import haff  # noqa: F401
    from haff import checkpoint, config as hcfg, inference, lisa, merge_lora, train_ds, weights as hw
    cfg = hcfg.tiny()
    sd = hw.make_state_dict(cfg, 9)
    base, clip, sam, records = _write_inputs(tmp_path, cfg, sd)
    # the plain base alone is NOT loadable for inference (no visual_model / text_hidden_fcs) ...
    with pytest.raises(KeyError):
        checkpoint.load_state_dict(str(base), str(clip))
    # ... and completes to the full trainable inventory the way the reference's fine-tune entrypoint builds it
    full = checkpoint.load_state_dict(str(base), str(clip), str(sam), for_training=True, seed=5)
    got_cfg = checkpoint.config_from_dir(str(base))
    assert got_cfg.llm.vocab == cfg.llm.vocab and got_cfg.seg_token_idx == cfg.seg_token_idx and got_cfg.sam.grid == cfg.sam.grid
    assert full["model.embed_tokens.weight"].shape[0] == cfg.llm.vocab
    assert torch.equal(full["model.visual_model.mask_decoder_right.iou_token.weight"], sd["model.visual_model.mask_decoder_left.iou_token.weight"])
    argv = ["--version", str(base), "--vision-tower", str(clip), "--vision_pretrained", str(sam), "--sam_records", str(records),
            "--epochs", "1", "--steps_per_epoch", "2", "--grad_accumulation_steps", "2", "--batch_size", "2", "--precision", "bf16",
            "--log_base_dir", str(tmp_path / "runs"), "--exp_name", "t", "--val_samples", "2", "--lr", "0.0003",
            "--image_size", str(cfg.sam.img_size), "--model_max_length", "3000", "--lora_r", "8", "--lora_alpha", "16"]
    train_ds.main(argv)
    out = capsys.readouterr().out
    assert "Epoch: [0][2/2]" in out and "IoU:" in out and "saved checkpoint" in out
    with pytest.raises(SystemExit):   # hub ids are refused loudly instead of being ignored
        train_ds.main(["--version", "liuhaotian/llava-v1.5-13b"])
    merged = tmp_path / "merged"
    merge_lora.main(["--version", str(base), "--weight", str(tmp_path / "runs" / "t" / "ckpt_model" / "latest.pt"),
                     "--save_path", str(merged), "--precision", "bf16", "--lora_r", "8", "--lora_alpha", "16",
                     "--vision_pretrained", str(sam)])
    mcfg = json.load(open(merged / "config.json"))
    assert mcfg["train_mask_decoder"] is True and mcfg["out_dim"] == 256 and mcfg["vocab_size"] == cfg.llm.vocab
    assert os.path.exists(merged / "tokenizer.model")
    model = lisa.LisaMI355.from_pretrained(str(merged), vision_tower=str(clip), torch_dtype=torch.bfloat16, device=dev)
    assert model.cfg.llm.vocab == cfg.llm.vocab and model.cfg.sam.img_size == cfg.sam.img_size
    # the merged checkpoint serves: the reference's inference CLI flags on it, sentencepiece prompt and all
    bench = tmp_path / "bench" / "kitchen" / "clip0"
    bench.mkdir(parents=True)
    from PIL import Image
    Image.fromarray(np.random.default_rng(1).integers(0, 256, size=(150, 224, 3), dtype=np.uint8)).save(bench / "inpainting.png")
    (bench / "annotation.json").write_text(json.dumps({"narration": "cut the bread"}))
    orig = lisa.LisaMI355.evaluate

    def forced(self, *a, **kw):
        kw["forced_answer"] = torch.tensor([[5, self.cfg.seg_token_idx, self.cfg.eos_token_id]])
        kw["max_new_tokens"] = 3
        return orig(self, *a, **kw)
    monkeypatch.setattr(lisa.LisaMI355, "evaluate", forced)
    inference.main(["--version", str(merged), "--vision-tower", str(clip), "--benchmark-dir", str(tmp_path / "bench"),
                    "--vis_save_path", str(tmp_path / "vis"), "--image_size", str(cfg.sam.img_size)])
    written = [th for th in (0.1, 0.2, 0.3, 0.5, 0.7) for side in ("left", "right")
               if os.path.exists(f"{tmp_path / 'vis'}{th}/kitchen/clip0/aff_{side}.png")]
    assert len(written) in (5, 10)


def test_from_pretrained_on_files_written_by_transformers_matches_the_oracle(dev, tmp_path):
    """f1 without the circle: the Llama shards + index + config.json and the CLIP directory under tests/golden/ were written by
    transformers' own save_pretrained (oracle/make_golden_files.py, build container); the tensors no third-party writer exists
    for (mm_projector, text_hidden_fcs, visual_model.* — a LISAForCausalLM cannot be constructed offline) join them as one
    more safetensors shard written with the safetensors library, listed in HF's own index file. LisaMI355.from_pretrained on
    that directory must serve exactly what the CPU oracle computes from the seeded state dict: fp32 logits within 1e-3, the
    same masks, the same tokens."""
    import shutil
    from safetensors.torch import save_file
    import haff  # noqa: F401
    from haff import config as hcfg, lisa, weights as hw
    from oracle import lisa_oracle as O
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cfg = hcfg.tiny()
    sd = hw.make_state_dict(cfg, 41)
    d = tmp_path / "merged"
    shutil.copytree(os.path.join(gold, "hf_llama_tiny"), d)
    extra = {k: v.contiguous() for k, v in sd.items()
             if k.startswith("model.visual_model.") or k.startswith("model.text_hidden_fcs.") or k.startswith("model.mm_projector.")}
    save_file(extra, str(d / "model-extra.safetensors"))
    idx = json.load(open(d / "model.safetensors.index.json"))
    idx["weight_map"].update({k: "model-extra.safetensors" for k in extra})
    json.dump(idx, open(d / "model.safetensors.index.json", "w"))
    c = json.load(open(d / "config.json"))
    c["haff_geometry"] = {"name": cfg.name, "sam": {k: getattr(cfg.sam, k) for k in ("img_size", "patch", "embed_dim", "depth", "heads", "window", "global_idx")},
                          "clip": {k: getattr(cfg.clip, k) for k in ("image", "patch", "hidden", "layers", "heads", "mlp")}}
    json.dump(c, open(d / "config.json", "w"))       # (vocab_size stays what transformers wrote: 323, no modulo-3 hint)
    # without tokenizer files nothing says that the 323 rows include the three added tokens: refused ...
    with pytest.raises(ValueError, match="cannot tell"):
        lisa.LisaMI355.from_pretrained(str(d), vision_tower=os.path.join(gold, "hf_clip_tiny"), torch_dtype=torch.float32, device=dev)
    # ... the reference's merge script saves the tokenizer beside the weights (merge_lora_weights_and_save_hf_model.py:155):
    # added_tokens.json, as HF's tokenizer.save_pretrained writes it, is the authority
    json.dump({"[SEG]": cfg.seg_token_idx, "<im_start>": cfg.im_start_idx, "<im_end>": cfg.im_end_idx}, open(d / "added_tokens.json", "w"))
    model = lisa.LisaMI355.from_pretrained(str(d), vision_tower=os.path.join(gold, "hf_clip_tiny"), torch_dtype=torch.float32, device=dev)
    assert model.cfg.llm.vocab == cfg.llm.vocab and model.cfg.seg_token_idx == cfg.seg_token_idx
    rng = np.random.default_rng(3)
    S = cfg.sam.img_size
    images = torch.from_numpy(rng.standard_normal((2, 3, S, S), dtype=np.float32))
    clip = torch.from_numpy(rng.standard_normal((2, 3, 224, 224), dtype=np.float32))
    ids = torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx, 9, 8, 7, 6]]).expand(2, -1).contiguous()
    forced = torch.tensor([[5, cfg.seg_token_idx, cfg.eos_token_id]]).expand(2, -1).contiguous()
    sizes, orig = [(S, S)] * 2, [(S, S), (150, 200)]
    with torch.no_grad():
        r_ids, r_l, r_r, r_t = O.lisa_evaluate(sd, cfg, clip, images, ids, sizes, orig, max_new_tokens=3, forced_answer=forced, use_cache=True)
    o_ids, left, right, tax = model.evaluate(clip.to(dev), images.to(dev), ids.to(dev), sizes, orig, max_new_tokens=3, forced_answer=forced)
    assert torch.equal(o_ids.cpu(), r_ids)
    for got, ref in list(zip(left, r_l)) + list(zip(right, r_r)):
        g = got.cpu()
        assert (g - ref).abs().max().item() <= 1e-3
        safe = ref.abs() > 1e-3
        assert torch.equal((g > 0)[safe], (ref > 0)[safe])
    for got, ref in zip(tax, r_t):
        assert (got.cpu() - ref).abs().max().item() <= 1e-4
