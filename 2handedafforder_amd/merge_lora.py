"""Fold a fine-tune checkpoint back into a plain HF-style checkpoint — the step between train_ds.py and inference.py.

Mirrors `2Haff/merge_lora_weights_and_save_hf_model.py:91-155`: rebuild the LoRA model, load the trained tensors,
`merge_and_unload()` (peft: W += (alpha / r) * B @ A on every adapted Linear, here q_proj and v_proj of each Llama layer,
train_ds.py:192-230), drop every `vision_tower` key (the CLIP tower is loaded separately, clip_encoder.py:21-29) and
`save_pretrained` (sharded weights + an index json + config.json). The trained tensors are the ones
`train_model.LisaTrainable.state_dict()` holds: the LoRA pairs plus the fully trained `embed_tokens`, `lm_head`,
`text_hidden_fcs` and both mask decoders (train_ds.py:233-244).

The merge arithmetic is fp32 on the host and rounded once to the save dtype, as peft does on the module's dtype-cast
weights; `checkpoint.load_state_dict` reads the result back.
"""
import argparse
import json
import os
from collections import OrderedDict

import torch


def merge_state_dict(base_sd, trained, lora_r, lora_alpha, dtype=torch.bfloat16):
    """base_sd: reference-keyed full state dict; trained: LisaTrainable.state_dict(). Returns the merged state dict
    (no `.lora_*` keys, no `vision_tower` keys), tensors on CPU in `dtype`."""
    scale = float(lora_alpha) / float(lora_r)
    out = OrderedDict()
    for k, v in base_sd.items():
        if "vision_tower" in k:
            continue
        out[k] = v.detach().to("cpu")
    for k, v in trained.items():
        if k.endswith(".lora_A") or k.endswith(".lora_B"):
            continue
        # tensors the fine-tune CREATED on top of a plain LLaVA base (text_hidden_fcs, the mask decoders with their taxonomy
        # head, the resized embed_tokens / lm_head) have no counterpart, or a smaller one, in the base: the trained one wins
        out[k] = v.detach().to("cpu")
    for k in [k for k in trained if k.endswith(".lora_A")]:
        mod = k[: -len(".lora_A")]
        a, b = trained[k].detach().float().cpu(), trained[mod + ".lora_B"].detach().float().cpu()
        if a.shape[0] != lora_r or b.shape[1] != lora_r:
            raise ValueError(f"{mod}: adapter rank {a.shape[0]} != --lora_r {lora_r}")
        w = out[mod + ".weight"].float()
        out[mod + ".weight"] = w + scale * (b @ a)
    return OrderedDict((k, v.to(dtype).contiguous()) for k, v in out.items())


def save_pretrained(state_dict, save_path, config=None, max_shard_bytes=10 << 30):
    """HF layout: model-0000x-of-0000n.safetensors + model.safetensors.index.json (+ config.json)."""
    from safetensors.torch import save_file
    os.makedirs(save_path, exist_ok=True)
    shards, cur, cur_bytes = [], OrderedDict(), 0
    for k, v in state_dict.items():
        nb = v.numel() * v.element_size()
        if cur and cur_bytes + nb > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = OrderedDict(), 0
        cur[k] = v
        cur_bytes += nb
    if cur:
        shards.append(cur)
    weight_map, total = {}, 0
    for i, sh in enumerate(shards):
        name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(dict(sh), os.path.join(save_path, name), metadata={"format": "pt"})
        for k, v in sh.items():
            weight_map[k] = name
            total += v.numel() * v.element_size()
    with open(os.path.join(save_path, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)
    if config is not None:
        with open(os.path.join(save_path, "config.json"), "w") as f:
            json.dump(config, f, indent=1)
    return [os.path.join(save_path, n) for n in sorted(set(weight_map.values()))]


def hf_config(cfg, dtype):
    """The config.json fields checkpoint.config_from_dir (and the reference's from_pretrained) read back."""
    l = cfg.llm
    return {"architectures": ["LISAForCausalLM"], "model_type": "llava", "hidden_size": l.hidden,
            "num_hidden_layers": l.layers, "num_attention_heads": l.heads, "intermediate_size": l.ffn,
            "vocab_size": l.vocab, "rms_norm_eps": l.rms_eps, "rope_theta": l.rope_theta,
            "mm_vision_select_layer": cfg.clip.select_layer, "mm_use_im_start_end": True,
            # the two keys LISAForCausalLM.__init__ branches on when the merged checkpoint is loaded again (LISA.py:129-141)
            "train_mask_decoder": True, "out_dim": cfg.out_dim, "haff_vocab_includes_added_tokens": True,
            "haff_geometry": {"name": cfg.name, "sam": dict(vars(cfg.sam)), "clip": dict(vars(cfg.clip))},
            "bos_token_id": cfg.bos_token_id, "eos_token_id": cfg.eos_token_id, "pad_token_id": cfg.pad_token_id,
            "torch_dtype": {torch.bfloat16: "bfloat16", torch.float32: "float32", torch.float16: "float16"}[dtype]}


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="merge lora weights and save model with hf format")
    p.add_argument("--version", required=True, help="base checkpoint directory (HF layout)")
    p.add_argument("--weight", required=True, help="fine-tune checkpoint: train_ds.py's latest.pt or a bare state dict")
    p.add_argument("--save_path", default="./lisa_model", type=str)
    p.add_argument("--precision", default="bf16", choices=["fp32", "bf16", "fp16"])
    p.add_argument("--lora_r", default=8, type=int)
    p.add_argument("--lora_alpha", default=16, type=int)
    p.add_argument("--vision_pretrained", default="", type=str, help="SAM checkpoint (sam_vit_h_4b8939.pth) when the base has no visual_model")
    return p.parse_args(argv)


def main(argv=None):
    from . import checkpoint
    args = parse_args(argv)
    dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.precision]
    cfg = checkpoint.config_from_dir(args.version)
    base = checkpoint.load_hf_dir(args.version)
    blob = torch.load(args.weight, map_location="cpu", weights_only=False)
    trained = blob["params"] if isinstance(blob, dict) and "params" in blob else blob
    if args.vision_pretrained:   # frozen SAM encoder / prompt encoder of the fine-tune run (the decoders come from `trained`)
        for k, v in checkpoint._load_file(args.vision_pretrained).items():
            if not k.startswith("mask_decoder."):
                base.setdefault("model.visual_model." + k, v)
    merged = merge_state_dict(base, trained, args.lora_r, args.lora_alpha, dtype)
    files = save_pretrained(merged, args.save_path, hf_config(cfg, dtype))
    tok = os.path.join(args.version, "tokenizer.model")
    if os.path.isfile(tok):
        import shutil
        shutil.copy(tok, os.path.join(args.save_path, "tokenizer.model"))
    print(f"merged {sum(k.endswith('.lora_A') for k in trained)} LoRA pairs; wrote {len(files)} shard(s) to {args.save_path}")


if __name__ == "__main__":
    main()
