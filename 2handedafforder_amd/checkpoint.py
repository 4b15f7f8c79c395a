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
    """Model geometry from the checkpoint's config.json: the Llama fields HF writes (hidden_size, num_hidden_layers,
    num_attention_heads, intermediate_size, vocab_size, rms_norm_eps, rope_theta), LLaVA's mm_vision_select_layer and
    LISA's out_dim. SAM ViT-H and CLIP-L/14 are what the reference hard-wires (build_sam.py:15-23, inference.py:34-36);
    a non-default vision geometry (this repo's reduced test models) travels in the optional "haff_geometry" key."""
    with open(os.path.join(path, "config.json")) as f:
        c = json.load(f)
    cfg = hcfg.haff_13b() if int(c.get("hidden_size", 4096)) == 5120 else hcfg.haff_7b()
    l = cfg.llm
    l.hidden = int(c.get("hidden_size", l.hidden))
    l.layers = int(c.get("num_hidden_layers", l.layers))
    l.heads = int(c.get("num_attention_heads", l.heads))
    l.ffn = int(c.get("intermediate_size", l.ffn))
    l.rope_theta = float(c.get("rope_theta", l.rope_theta))
    l.rms_eps = float(c.get("rms_norm_eps", l.rms_eps))
    # a plain LLaVA base says 32000; the fine-tune adds [SEG], <im_start>, <im_end> (train_ds.py:142-149) -> +3
    n_base = int(c.get("vocab_size", l.vocab))
    l.vocab = n_base if c.get("haff_vocab_includes_added_tokens", n_base % 1000 == 3) else n_base + 3
    cfg.seg_token_idx, cfg.im_start_idx, cfg.im_end_idx = l.vocab - 3, l.vocab - 2, l.vocab - 1
    cfg.out_dim = int(c.get("out_dim", cfg.out_dim))
    cfg.clip.select_layer = int(c.get("mm_vision_select_layer", cfg.clip.select_layer))
    for k in ("bos_token_id", "eos_token_id", "pad_token_id"):
        if c.get(k) is not None:
            setattr(cfg, k, int(c[k]))
    geo = c.get("haff_geometry") or {}
    for name, sub in (("sam", cfg.sam), ("clip", cfg.clip)):
        for k, v in (geo.get(name) or {}).items():
            setattr(sub, k, tuple(v) if isinstance(v, list) else v)
    if "name" in geo:
        cfg.name = geo["name"]
    return cfg


def added_token_ids(path):
    """{"[SEG]": id, "<im_start>": id, "<im_end>": id} from the directory's added_tokens.json (what HF's save_pretrained writes
    for tokens added with add_tokens, train_ds.py:142-149), or None when the file or one of the three is missing."""
    fn = os.path.join(path, "added_tokens.json")
    if not os.path.exists(fn):
        return None
    with open(fn) as f:
        d = json.load(f)
    want = ("[SEG]", "<im_start>", "<im_end>")
    return {k: int(d[k]) for k in want} if all(k in d for k in want) else None


def sentencepiece_vocab_size(path):
    """Pieces in the directory's tokenizer.model (the base Llama vocabulary: 32000), or None without the file."""
    fn = os.path.join(path, "tokenizer.model")
    if not os.path.exists(fn):
        return None
    try:
        import sentencepiece as spm
    except ImportError as e:   # name the dependency: the file is there, the package that reads it is not
        raise ImportError(f"{fn} is a sentencepiece model: reading the base vocabulary size needs the `sentencepiece` package "
                          "(pip install sentencepiece), or put an added_tokens.json beside it") from e
    sp = spm.SentencePieceProcessor()
    sp.Load(fn)
    return int(sp.GetPieceSize())


def resolve_added_tokens(path, rows, layout_asserted=False):
    """Ids of [SEG] / <im_start> / <im_end> for a checkpoint directory whose embedding has `rows` rows (LisaMI355.from_pretrained
    documents the order of authority); ValueError when they cannot be told. Without tokenizer files, "the last three rows" is
    taken only where something SAYS the three rows are there: rows == config.json's vocab_size + 3, or rows == vocab_size in a
    directory this repo wrote (config.json carries haff_vocab_includes_added_tokens: merge_lora.py) or whose caller passed an
    explicit seg_token_idx (layout_asserted) — a plain base-Llama directory without tokenizer files is refused."""
    ids = added_token_ids(path)
    if ids is None:
        base = sentencepiece_vocab_size(path)
        with open(os.path.join(path, "config.json")) as f:
            cj = json.load(f)
        n_cfg = int(cj.get("vocab_size", rows))
        counted = bool(cj.get("haff_vocab_includes_added_tokens", False)) or layout_asserted
        if (base is not None and rows == base + 3) or (base is None and (rows == n_cfg + 3 or (rows == n_cfg and counted))):
            ids = {"[SEG]": rows - 3, "<im_start>": rows - 2, "<im_end>": rows - 1}
        else:
            raise ValueError(f"{path}: embed_tokens has {rows} rows, config.json says vocab_size {n_cfg}, the tokenizer's base "
                             f"vocabulary is {base}: cannot tell which rows are [SEG] / <im_start> / <im_end> (expected base + 3 "
                             "rows, or an added_tokens.json)")
    if max(ids.values()) >= rows:
        raise ValueError(f"added token ids {ids} do not fit the {rows} embedding rows")
    return ids


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


def load_state_dict(version_dir, clip_dir=None, sam_ckpt=None, for_training=False, seed=0, cfg=None):
    """Reference-keyed state dict for LisaMI355 from on-disk checkpoints. for_training: a plain LLaVA base is completed
    the way the reference's fine-tune entrypoint does it (complete_for_training). cfg: the geometry of the model about to
    be built, when the caller has adjusted what config.json says (train_ds.py: --out_dim, a vocabulary grown to
    len(tokenizer)); default: config_from_dir(version_dir)."""
    if cfg is None:
        cfg = config_from_dir(version_dir)
    sd = load_hf_dir(version_dir)
    if clip_dir is not None:
        clip_sd = load_hf_dir(clip_dir)
        # hub / transformers 4.x files carry the tower as "vision_model.*" (inside a full CLIPModel: text_model.* and the
        # projections are ignored, as CLIPVisionModel.from_pretrained does, clip_encoder.py:25); a bare CLIPVisionModel saved by
        # transformers 5.x has the same tensors without that prefix
        bare = not any(k.startswith("vision_model.") for k in clip_sd)
        for k, v in clip_sd.items():
            if k.startswith("vision_model."):
                sd["model.vision_tower.vision_tower." + k] = v
            elif bare and k.split(".")[0] in ("embeddings", "pre_layrnorm", "encoder", "post_layernorm"):
                sd["model.vision_tower.vision_tower.vision_model." + k] = v
    if sam_ckpt is not None:
        for k, v in _load_file(sam_ckpt).items():
            if k.startswith("mask_decoder."):
                sd["model.visual_model." + k.replace("mask_decoder", "mask_decoder_left", 1)] = v
                sd["model.visual_model." + k.replace("mask_decoder", "mask_decoder_right", 1)] = v
            else:
                sd.setdefault("model.visual_model." + k, v)
    if for_training:
        complete_for_training(sd, cfg, seed)
    missing = [k for k in hw.all_shapes(cfg) if k not in sd and "post_layernorm" not in k]
    if missing:
        raise KeyError(f"{len(missing)} tensors missing from the checkpoint, e.g. {missing[:4]}")
    return sd


def complete_for_training(sd, cfg, seed=0):
    """What the reference builds on top of a plain LLaVA base before fine-tuning (train_ds.py:167-244,
    LISA.py:79-104): `resize_token_embeddings(len(tokenizer))` — embed_tokens / lm_head grow to the vocabulary with the
    added [SEG], <im_start>, <im_end> rows (new rows ~ N(0, initializer_range = 0.02), the HF _init_weights rule);
    `initialize_lisa_modules` — text_hidden_fcs freshly initialised (nn.Linear default: U(-1/sqrt(in), 1/sqrt(in)) for
    weight and bias) and SAM built from --vision_pretrained with mask_decoder.* duplicated left/right
    (build_sam.py:125-136; done by load_state_dict) plus the taxonomy head, which no SAM checkpoint has (nn.Linear default).
    Only tensors ABSENT from `sd` are created (a merged 2HAff checkpoint passes through unchanged). Seeded: every rank builds
    identical tensors."""
    g = torch.Generator().manual_seed(seed)
    shapes = hw.all_shapes(cfg)

    created = []
    for k in ("model.embed_tokens.weight", "lm_head.weight"):
        have, want = sd[k].shape[0], shapes[k][0]
        if have < want:
            extra = torch.randn((want - have, sd[k].shape[1]), generator=g) * 0.02
            sd[k] = torch.cat([sd[k].float(), extra], 0).to(sd[k].dtype)
            created.append(f"{k}[{have}:{want}]")
        elif have > want:   # resize_token_embeddings(len(tokenizer)) also shrinks a padded vocabulary
            sd[k] = sd[k][:want].contiguous()
    for k, shape in shapes.items():
        if k in sd:
            continue
        if "text_hidden_fcs" in k or "taxonomy_embed" in k:
            fan_in = shapes[k.replace(".bias", ".weight")][1]
            bound = 1.0 / fan_in ** 0.5
            sd[k] = (torch.rand(shape, generator=g) * 2 - 1) * bound
            created.append(k)
    return created


def synthetic_state_dict(cfg, seed, device=None, dtype=torch.bfloat16):
    if device is not None and torch.device(device).type == "cuda":
        return hw.make_state_dict_device(cfg, seed, device, dtype)
    return hw.make_state_dict(cfg, seed)


class SentencePieceTokenizer:
    """The reference's tokenizer: `AutoTokenizer.from_pretrained(version, use_fast=False)` (inference.py:115-127,
    train_ds.py:131-149) = transformers' slow LlamaTokenizer over `tokenizer.model`, plus the added tokens `[SEG]`,
    `<im_start>`, `<im_end>` appended after the sentencepiece vocabulary in that order (ids 32000..32002 for Llama-2).

    Restated from the published behaviour of the 4.31 slow tokenizer in its default (legacy) mode: the text is split on
    the added tokens, every remaining chunk is encoded by sentencepiece on its own (so each chunk gets the model's
    dummy-prefix space), BOS is prepended once, no EOS. Parity with transformers 4.31 is UNPINNED here: that version is
    not installed and the 5.x LlamaTokenizer is a different (fast) implementation; the tests pin the wrapper against
    sentencepiece itself and the id layout the config expects."""

    def __init__(self, model_file, cfg=None, added_tokens=("[SEG]", "<im_start>", "<im_end>")):
        import sentencepiece as spm
        self.sp = spm.SentencePieceProcessor(model_file=model_file)
        n = self.sp.get_piece_size()
        self.bos_token_id, self.eos_token_id, self.unk_token_id = self.sp.bos_id(), self.sp.eos_id(), self.sp.unk_id()
        self.pad_token_id = self.unk_token_id            # tokenizer.pad_token = tokenizer.unk_token (inference.py:121)
        self.special = {tok: n + i for i, tok in enumerate(added_tokens)}
        self.vocab_size = n + len(added_tokens)
        if cfg is not None:  # the ids the model was built for (train_ds.py:143-149) must match this vocabulary
            want = {"[SEG]": cfg.seg_token_idx, "<im_start>": cfg.im_start_idx, "<im_end>": cfg.im_end_idx}
            for tok, tid in want.items():
                if tok in self.special and self.special[tok] != tid:
                    raise ValueError(f"{tok}: tokenizer id {self.special[tok]} != model config id {tid}")

    def __len__(self):
        return self.vocab_size

    def _split(self, text):
        """[(is_added_token, piece)] in order, splitting on the added tokens (longest match first)."""
        toks = sorted(self.special, key=len, reverse=True)
        out, buf, i = [], [], 0
        while i < len(text):
            for t in toks:
                if text.startswith(t, i):
                    if buf:
                        out.append((False, "".join(buf)))
                        buf = []
                    out.append((True, t))
                    i += len(t)
                    break
            else:
                buf.append(text[i])
                i += 1
        if buf:
            out.append((False, "".join(buf)))
        return out

    def __call__(self, text, add_special_tokens=True):
        ids = [self.bos_token_id] if add_special_tokens else []
        for is_tok, piece in self._split(text):
            if is_tok:
                ids.append(self.special[piece])
            else:
                ids.extend(self.sp.encode(piece))

        class _R:
            pass
        r = _R()
        r.input_ids = ids
        return r

    def decode(self, ids, skip_special_tokens=False):
        inv = {v: k for k, v in self.special.items()}
        out, run = [], []

        def flush():
            if run:
                out.append(self.sp.decode(run))
                run.clear()
        for t in [int(x) for x in ids]:
            if t in inv:
                flush()
                out.append(inv[t])
            elif t in (self.bos_token_id, self.eos_token_id) or t < 0:
                flush()
                if not skip_special_tokens and t >= 0:
                    out.append("<s>" if t == self.bos_token_id else "</s>")
            elif t < self.sp.get_piece_size():
                run.append(t)
        flush()
        return "".join(out)


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
