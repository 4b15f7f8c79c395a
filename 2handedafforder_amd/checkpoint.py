"""Weight loading for the CLIs: the merged HF checkpoint written by the reference's
merge_lora_weights_and_save_hf_model.py:149-155 (sharded pytorch_model-*.bin or *.safetensors with an index json;
keys of weights.all_shapes, `vision_tower` excluded) + the CLIP tower loaded separately (clip_encoder.py:21-29) +
optionally SAM's sam_vit_h_4b8939.pth with mask_decoder.* duplicated into _left/_right (build_sam.py:125-136).
With no files on disk (this repo's offline setting) `synthetic_state_dict` supplies seeded random weights.
"""
import glob
import json
import os

import torch

from . import config as hcfg
from . import weights as hw


def config_from_dir(path):
    """7B vs 13B geometry from the checkpoint's config.json (hidden_size / num_hidden_layers)."""
    with open(os.path.join(path, "config.json")) as f:
        c = json.load(f)
    cfg = hcfg.haff_13b() if int(c.get("hidden_size", 4096)) == 5120 else hcfg.haff_7b()
    cfg.llm.vocab = int(c.get("vocab_size", cfg.llm.vocab))
    cfg.llm.rms_eps = float(c.get("rms_norm_eps", cfg.llm.rms_eps))
    cfg.clip.select_layer = int(c.get("mm_vision_select_layer", cfg.clip.select_layer))
    return cfg


def _load_file(fn):
    if fn.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(fn)
    return torch.load(fn, map_location="cpu", weights_only=True)


def load_hf_dir(path):
    """All tensors of a (possibly sharded) HF checkpoint directory."""
    files = sorted(glob.glob(os.path.join(path, "*.safetensors"))) or sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not files:
        raise FileNotFoundError(f"no checkpoint shards under {path}")
    sd = {}
    for fn in files:
        sd.update(_load_file(fn))
    return sd


def load_state_dict(version_dir, clip_dir=None, sam_ckpt=None):
    """Reference-keyed state dict for LisaMI355 from on-disk checkpoints."""
    sd = load_hf_dir(version_dir)
    if clip_dir is not None:
        for k, v in load_hf_dir(clip_dir).items():
            if k.startswith("vision_model."):
                sd["model.vision_tower.vision_tower." + k] = v
    if sam_ckpt is not None:
        for k, v in _load_file(sam_ckpt).items():
            if k.startswith("mask_decoder."):
                sd["model.visual_model." + k.replace("mask_decoder", "mask_decoder_left", 1)] = v
                sd["model.visual_model." + k.replace("mask_decoder", "mask_decoder_right", 1)] = v
            else:
                sd.setdefault("model.visual_model." + k, v)
    missing = [k for k in hw.all_shapes(config_from_dir(version_dir)) if k not in sd and "post_layernorm" not in k]
    if missing:
        raise KeyError(f"{len(missing)} tensors missing from the checkpoint, e.g. {missing[:4]}")
    return sd


def synthetic_state_dict(cfg, seed, device=None, dtype=torch.bfloat16):
    if device is not None and torch.device(device).type == "cuda":
        return hw.make_state_dict_device(cfg, seed, device, dtype)
    return hw.make_state_dict(cfg, seed)


class ByteTokenizer:
    """Offline stand-in for the sentencepiece tokenizer (no tokenizer.model exists in this setting): UTF-8 bytes
    shifted past the special ids, BOS prepended; the three added tokens of train_ds.py:142-149 map to the ids the
    config names. Only for synthetic-weight runs of the CLIs."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.bos_token_id, self.eos_token_id, self.unk_token_id = cfg.bos_token_id, cfg.eos_token_id, cfg.pad_token_id
        self.pad_token_id = cfg.pad_token_id
        self.special = {"[SEG]": cfg.seg_token_idx, "<im_start>": cfg.im_start_idx, "<im_end>": cfg.im_end_idx}

    def __call__(self, text, add_special_tokens=True):
        ids = [self.bos_token_id] if add_special_tokens else []
        i = 0
        while i < len(text):
            for tok, tid in self.special.items():
                if text.startswith(tok, i):
                    ids.append(tid)
                    i += len(tok)
                    break
            else:
                ids.extend(3 + (b % (min(self.cfg.llm.vocab, self.cfg.seg_token_idx) - 3)) for b in text[i].encode("utf-8"))
                i += 1

        class _R:
            pass
        r = _R()
        r.input_ids = ids
        return r

    def decode(self, ids, skip_special_tokens=False):
        inv = {v: k for k, v in self.special.items()}
        out = []
        for t in [int(x) for x in ids]:
            if t in inv:
                out.append(inv[t])
            elif t in (self.bos_token_id, self.eos_token_id, self.pad_token_id):
                out.append("" if skip_special_tokens else {self.bos_token_id: "<s>", self.eos_token_id: "</s>"}.get(t, "<unk>"))
            elif t >= 3:
                out.append(chr((t - 3) % 256) if 32 <= (t - 3) % 256 < 127 else "?")
        return "".join(out)
