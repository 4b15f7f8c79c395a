"""Generates the FILE fixtures of row f1 under tests/golden/ — run ONLY in the build container (needs /root/reference and
transformers):

    python oracle/make_golden_files.py

Why: `checkpoint.load_state_dict` / `config_from_dir` / `SentencePieceTokenizer` had only ever read files this repo's own
writers produced. These fixtures are written by the toolchains the reference uses:

  tests/golden/hf_llama_tiny/   LlamaForCausalLM(tiny).save_pretrained(max_shard_size=...) — transformers' own writer:
                                config.json, generation_config.json, sharded model-0000x-of-0000y.safetensors + index
                                (the reference saves its merged model with the same call,
                                merge_lora_weights_and_save_hf_model.py:146-155)
  tests/golden/hf_clip_tiny/    CLIPModel(tiny vision tower + a 1-layer text tower).save_pretrained — the hub layout
                                (vision_model.*, text_model.*, projections) that CLIPVisionModel.from_pretrained reads the tower
                                from (clip_encoder.py:21-29); hf_clip_tiny_v5_bare/: a bare CLIPVisionModel as transformers 5.x
                                writes it (keys WITHOUT the "vision_model." prefix)
  tests/golden/sam_ref_manifest.json
                                the reference's OWN Sam module (build_sam.py:59-117 arguments at the tiny geometry) holding the
                                seeded weights: state_dict() keys, shapes and SHA-1 of every tensor's bytes (the tensors
                                themselves are 44 MB: too large to commit; weights.py rebuilds them from the seed)
  tests/golden/tokenizer_tiny/  a small sentencepiece BPE model (byte fallback, Llama-style) + tokenizer_ids.json: prompts
                                tokenised by transformers 5.15's LlamaTokenizer over that model and by sentencepiece itself.
                                NOTE: transformers 5.15 ships NO slow (sentencepiece-backed) LlamaTokenizer any more — its
                                LlamaTokenizer is a `tokenizers` BPE conversion of the model file; the reference pins 4.31's slow
                                class (inference.py:115-127, use_fast=False). The fixture therefore pins what both agree on
                                (plain text chunks) and records where the 5.15 class differs (leading spaces, text after an added
                                token), cases for which the 4.31 legacy behaviour stays restated-from-publication only.

All weights come from 2handedafforder_amd/weights.py's seeded filler (seed 41), loaded INTO the third-party / reference
modules before they write; the tests rebuild the same tensors from the seed and demand bit-equality after the round trip.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import haff  # noqa: E402,F401
from haff import config as hcfg  # noqa: E402
from haff import weights as hw  # noqa: E402
from make_golden import build_ref_sam, load_ref_modeling  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 41


def sha(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def main():
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM
    cfg = hcfg.tiny()
    sd = hw.make_state_dict(cfg, SEED)

    # ---- Llama: transformers' save_pretrained, sharded --------------------------------------------------------------
    l = cfg.llm
    hf = LlamaForCausalLM(LlamaConfig(vocab_size=l.vocab, hidden_size=l.hidden, intermediate_size=l.ffn, num_hidden_layers=l.layers,
                                      num_attention_heads=l.heads, num_key_value_heads=l.heads, rms_norm_eps=l.rms_eps,
                                      rope_theta=l.rope_theta, max_position_embeddings=512, tie_word_embeddings=False,
                                      bos_token_id=1, eos_token_id=2, pad_token_id=0))
    want = {k: v for k, v in sd.items() if k.startswith("model.layers.") or k in ("model.norm.weight", "model.embed_tokens.weight", "lm_head.weight")}
    missing, unexpected = hf.load_state_dict(want, strict=False)
    assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
    d = os.path.join(OUT, "hf_llama_tiny")
    shutil.rmtree(d, ignore_errors=True)
    hf.save_pretrained(d, max_shard_size="120KB", safe_serialization=True)
    print("hf_llama_tiny:", sorted(os.listdir(d)))

    # ---- CLIP: the hub layout is a full CLIPModel (vision_model.* + text_model.* + projections), from which
    # CLIPVisionModel.from_pretrained (clip_encoder.py:25) takes the vision tower; a tiny text tower rides along to keep it so
    from transformers import CLIPConfig, CLIPModel, CLIPTextConfig
    c = cfg.clip
    vcfg = CLIPVisionConfig(hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers, num_attention_heads=c.heads,
                            image_size=c.image, patch_size=c.patch, layer_norm_eps=c.eps, hidden_act="quick_gelu", projection_dim=32)
    tcfg = CLIPTextConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                          max_position_embeddings=16, projection_dim=32)
    clip = CLIPModel(CLIPConfig(text_config=tcfg.to_dict(), vision_config=vcfg.to_dict(), projection_dim=32))
    pfx = "model.vision_tower.vision_tower."
    missing, unexpected = clip.load_state_dict({k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)}, strict=False)
    assert not unexpected, unexpected
    assert all(not m.startswith("vision_model.") or "post_layernorm" in m for m in missing), missing
    d = os.path.join(OUT, "hf_clip_tiny")
    shutil.rmtree(d, ignore_errors=True)
    clip.save_pretrained(d, safe_serialization=True)
    print("hf_clip_tiny:", sorted(os.listdir(d)))
    # transformers 5.x writes a bare CLIPVisionModel WITHOUT the "vision_model." prefix (4.31 and the hub files carry it): a second,
    # one-layer directory in that layout keeps the loader honest about both
    vm = CLIPVisionModel(vcfg)
    vm.load_state_dict({k[len(pfx) + len("vision_model."):]: v for k, v in sd.items() if k.startswith(pfx + "vision_model.")}, strict=False)
    d = os.path.join(OUT, "hf_clip_tiny_v5_bare")
    shutil.rmtree(d, ignore_errors=True)
    vm.save_pretrained(d, safe_serialization=True)
    print("hf_clip_tiny_v5_bare:", sorted(os.listdir(d)))

    # ---- SAM: the reference's own module as the key / shape / content authority ---------------------------------------------
    ref = load_ref_modeling()
    sam = build_ref_sam(ref, cfg.sam)
    V = "model.visual_model."
    missing, unexpected = sam.load_state_dict({k[len(V):]: v for k, v in sd.items() if k.startswith(V)}, strict=False)
    assert not unexpected, unexpected
    man = {"seed": SEED, "geometry": "tiny", "keys_not_in_seeded_inventory": sorted(missing), "tensors": {}}
    for k, v in sam.state_dict().items():
        if k in missing:
            man["tensors"][k] = {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""), "sha1": None}
        else:
            man["tensors"][k] = {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""), "sha1": sha(v)}
    json.dump(man, open(os.path.join(OUT, "sam_ref_manifest.json"), "w"), indent=0)
    print("sam_ref_manifest:", len(man["tensors"]), "tensors,", len(missing), "outside the seeded inventory")

    # ---- tokenizer ------------------------------------------------------------------------------------------------------------
    import sentencepiece as spm
    from transformers import LlamaTokenizer
    d = os.path.join(OUT, "tokenizer_tiny")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    words = ("where would someone grasp the cup to pour water cut bread with knife open drawer bottle hold pan stir pot please segment "
             "region perform action image can you show me interact objects following task sure it is result a chat between curious "
             "human and an artificial intelligence assistant gives helpful detailed polite answers user questions").split()
    rng = np.random.default_rng(0)
    corpus = os.path.join(d, "corpus.txt")
    with open(corpus, "w") as f:
        f.write("\n".join(" ".join(rng.choice(words, size=14)).capitalize() + rng.choice([".", "?", "!", ":"]) for _ in range(800)))
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(d, "tokenizer"), vocab_size=l.vocab - 3, model_type="bpe",
                                   byte_fallback=True, character_coverage=1.0, unk_id=0, bos_id=1, eos_id=2, pad_id=-1,
                                   normalization_rule_name="identity", add_dummy_prefix=True, remove_extra_whitespaces=False,
                                   split_digits=True, allow_whitespace_only_pieces=True, minloglevel=2)
    os.remove(corpus)
    os.remove(os.path.join(d, "tokenizer.vocab"))
    sp = spm.SentencePieceProcessor(model_file=os.path.join(d, "tokenizer.model"))
    assert sp.get_piece_size() == l.vocab - 3
    tk = LlamaTokenizer.from_pretrained(d)
    tk.add_tokens("[SEG]")                                                  # train_ds.py:142-149, in that order
    tk.add_tokens(["<im_start>", "<im_end>"], special_tokens=True)
    prompts = ["A chat between a curious human and an artificial intelligence assistant.",
               "Where would someone grasp the cup to pour water?", "Can you segment the region to cut bread with the knife?",
               "Sure, it is", "USER: please show me. ASSISTANT:", "stir the pot 12 times!", "naive café",
               # cases where the 5.15 tokenizers-backend class is NOT the 4.31 slow class (recorded, not pinned):
               " leading space", "Sure, [SEG] .", "<im_start><image><im_end>\nhold the pan"]
    rec = {"transformers_version": transformers.__version__, "tokenizer_class": type(tk).__name__,
           "backend": type(tk).__mro__[1].__name__, "added_token_ids": {t: tk.convert_tokens_to_ids(t) for t in ("[SEG]", "<im_start>", "<im_end>")},
           "len": len(tk), "prompts": []}
    for s in prompts:
        rec["prompts"].append({"text": s, "hf_ids_no_bos": tk(s, add_special_tokens=False).input_ids, "spm_ids": sp.encode(s)})
    open(os.path.join(d, "tokenizer_ids.json"), "w").write(json.dumps(rec, separators=(",", ":")).replace('{"text"', '\n{"text"'))
    print("tokenizer_tiny:", sorted(os.listdir(d)), rec["added_token_ids"], rec["backend"])


if __name__ == "__main__":
    main()
