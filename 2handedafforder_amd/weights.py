"""Parameter inventory of the 2Haff stack under the REFERENCE's state-dict key names, plus a deterministic
synthetic filler (there are no checkpoints offline).

Key layout = what `merge_lora_weights_and_save_hf_model.py:149-155` saves and `inference.py:158-168` loads
(temp_log.txt:23-60 confirms the prefixes): model.layers.*, model.embed_tokens, model.norm, lm_head,
model.mm_projector, model.vision_tower.vision_tower.* (CLIP, loaded separately by clip_encoder.py:21-29),
model.visual_model.{image_encoder,prompt_encoder,mask_decoder_left,mask_decoder_right}.*, model.text_hidden_fcs.0.{0,2}.

The filler iterates sorted(keys) with numpy PCG64 — stable across platforms — so the build container (where the
reference modules are importable and golden outputs are captured) and the GPU box rebuild identical weights
from (config, seed) alone; fixtures therefore carry no weights.
"""
from collections import OrderedDict

import numpy as np
import torch

V = "model.visual_model"
CLIP = "model.vision_tower.vision_tower.vision_model"


def _linear(shapes, name, out_f, in_f, bias=True):
    shapes[name + ".weight"] = (out_f, in_f)
    if bias:
        shapes[name + ".bias"] = (out_f,)


def _norm(shapes, name, c, bias=True):
    shapes[name + ".weight"] = (c,)
    if bias:
        shapes[name + ".bias"] = (c,)


def sam_shapes(cfg, shapes=None):
    """ImageEncoderViT + PromptEncoder(text path) + the two MaskDecoders (build_sam.py:59-117)."""
    shapes = OrderedDict() if shapes is None else shapes
    s = cfg
    E = V + ".image_encoder"
    g = s.grid
    shapes[E + ".patch_embed.proj.weight"] = (s.embed_dim, 3, s.patch, s.patch)
    shapes[E + ".patch_embed.proj.bias"] = (s.embed_dim,)
    shapes[E + ".pos_embed"] = (1, g, g, s.embed_dim)
    hd = s.embed_dim // s.heads
    for i in range(s.depth):
        B = f"{E}.blocks.{i}"
        rp = g if i in s.global_idx else s.window
        _norm(shapes, B + ".norm1", s.embed_dim)
        _linear(shapes, B + ".attn.qkv", 3 * s.embed_dim, s.embed_dim)
        _linear(shapes, B + ".attn.proj", s.embed_dim, s.embed_dim)
        shapes[B + ".attn.rel_pos_h"] = (2 * rp - 1, hd)
        shapes[B + ".attn.rel_pos_w"] = (2 * rp - 1, hd)
        _norm(shapes, B + ".norm2", s.embed_dim)
        _linear(shapes, B + ".mlp.lin1", s.mlp_ratio * s.embed_dim, s.embed_dim)
        _linear(shapes, B + ".mlp.lin2", s.embed_dim, s.mlp_ratio * s.embed_dim)
    shapes[E + ".neck.0.weight"] = (s.out_chans, s.embed_dim, 1, 1)
    _norm(shapes, E + ".neck.1", s.out_chans)
    shapes[E + ".neck.2.weight"] = (s.out_chans, s.out_chans, 3, 3)
    _norm(shapes, E + ".neck.3", s.out_chans)
    P = V + ".prompt_encoder"
    C = s.out_chans
    shapes[P + ".pe_layer.positional_encoding_gaussian_matrix"] = (2, C // 2)
    shapes[P + ".no_mask_embed.weight"] = (1, C)
    for side in ("left", "right"):
        D = f"{V}.mask_decoder_{side}"
        shapes[D + ".iou_token.weight"] = (1, C)
        shapes[D + ".mask_tokens.weight"] = (4, C)
        T = D + ".transformer"

        def attn(name, internal):
            _linear(shapes, name + ".q_proj", internal, C)
            _linear(shapes, name + ".k_proj", internal, C)
            _linear(shapes, name + ".v_proj", internal, C)
            _linear(shapes, name + ".out_proj", C, internal)
        for l in range(2):
            L = f"{T}.layers.{l}"
            attn(L + ".self_attn", C)
            _norm(shapes, L + ".norm1", C)
            attn(L + ".cross_attn_token_to_image", C // 2)
            _norm(shapes, L + ".norm2", C)
            _linear(shapes, L + ".mlp.lin1", 2048, C)
            _linear(shapes, L + ".mlp.lin2", C, 2048)
            _norm(shapes, L + ".norm3", C)
            _norm(shapes, L + ".norm4", C)
            attn(L + ".cross_attn_image_to_token", C // 2)
        attn(T + ".final_attn_token_to_image", C // 2)
        _norm(shapes, T + ".norm_final_attn", C)
        shapes[D + ".output_upscaling.0.weight"] = (C, C // 4, 2, 2)
        shapes[D + ".output_upscaling.0.bias"] = (C // 4,)
        _norm(shapes, D + ".output_upscaling.1", C // 4)
        shapes[D + ".output_upscaling.3.weight"] = (C // 4, C // 8, 2, 2)
        shapes[D + ".output_upscaling.3.bias"] = (C // 8,)
        for i in range(4):
            M = f"{D}.output_hypernetworks_mlps.{i}"
            _linear(shapes, M + ".layers.0", C, C)
            _linear(shapes, M + ".layers.1", C, C)
            _linear(shapes, M + ".layers.2", C // 8, C)
        M = D + ".iou_prediction_head"
        _linear(shapes, M + ".layers.0", 256, C)
        _linear(shapes, M + ".layers.1", 256, 256)
        _linear(shapes, M + ".layers.2", 4, 256)
        if side == "left":
            M = D + ".taxonomy_embed"
            _linear(shapes, M + ".layers.0", 4 * C, 4 * C)
            _linear(shapes, M + ".layers.1", 4 * C, 4 * C)
            _linear(shapes, M + ".layers.2", 4, 4 * C)
    return shapes


def clip_shapes(cfg, shapes=None):
    shapes = OrderedDict() if shapes is None else shapes
    c = cfg
    shapes[CLIP + ".embeddings.class_embedding"] = (c.hidden,)
    shapes[CLIP + ".embeddings.patch_embedding.weight"] = (c.hidden, 3, c.patch, c.patch)
    shapes[CLIP + ".embeddings.position_embedding.weight"] = (c.n_patches + 1, c.hidden)
    _norm(shapes, CLIP + ".pre_layrnorm", c.hidden)
    for i in range(c.layers):
        L = f"{CLIP}.encoder.layers.{i}"
        _norm(shapes, L + ".layer_norm1", c.hidden)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _linear(shapes, f"{L}.self_attn.{n}", c.hidden, c.hidden)
        _norm(shapes, L + ".layer_norm2", c.hidden)
        _linear(shapes, L + ".mlp.fc1", c.mlp, c.hidden)
        _linear(shapes, L + ".mlp.fc2", c.hidden, c.mlp)
    _norm(shapes, CLIP + ".post_layernorm", c.hidden)
    return shapes


def llm_shapes(cfg, shapes=None):
    shapes = OrderedDict() if shapes is None else shapes
    l = cfg.llm
    shapes["model.embed_tokens.weight"] = (l.vocab, l.hidden)
    for i in range(l.layers):
        L = f"model.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            _linear(shapes, f"{L}.self_attn.{n}", l.hidden, l.hidden, bias=False)
        _linear(shapes, L + ".mlp.gate_proj", l.ffn, l.hidden, bias=False)
        _linear(shapes, L + ".mlp.up_proj", l.ffn, l.hidden, bias=False)
        _linear(shapes, L + ".mlp.down_proj", l.hidden, l.ffn, bias=False)
        _norm(shapes, L + ".input_layernorm", l.hidden, bias=False)
        _norm(shapes, L + ".post_attention_layernorm", l.hidden, bias=False)
    _norm(shapes, "model.norm", l.hidden, bias=False)
    shapes["lm_head.weight"] = (l.vocab, l.hidden)
    _linear(shapes, "model.mm_projector", l.hidden, cfg.clip.hidden)
    _linear(shapes, "model.text_hidden_fcs.0.0", l.hidden, l.hidden)
    _linear(shapes, "model.text_hidden_fcs.0.2", cfg.out_dim, l.hidden)
    return shapes


def all_shapes(cfg):
    shapes = OrderedDict()
    sam_shapes(cfg.sam, shapes)
    clip_shapes(cfg.clip, shapes)
    llm_shapes(cfg, shapes)
    return shapes


def _std_for(key, shape):
    """Per-tensor sigma: ~1/sqrt(fan_in) for matrices so activations stay O(1) through deep stacks."""
    if len(shape) == 1 and key.endswith(".weight"):
        return None  # every 1-D ".weight" is a norm gain: 1 + 0.1*N(0,1)
    if key.endswith(".bias"):
        return 0.02
    if "rel_pos" in key:
        return 0.1
    if key.endswith("pos_embed") or "position_embedding" in key or "class_embedding" in key:
        return 0.1
    if "positional_encoding_gaussian_matrix" in key:
        return 1.0
    if key.endswith("embed_tokens.weight"):
        return 1.0
    if key.endswith("_token.weight") or key.endswith("_tokens.weight") or key.endswith("no_mask_embed.weight"):
        return 1.0
    if "output_upscaling" in key and key.endswith("weight"):
        return 1.0 / np.sqrt(shape[0])  # ConvTranspose2d: fan-in is dim 0
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return 1.0 / np.sqrt(fan_in)


def make_state_dict(cfg, seed, shapes=None, dtype=torch.float32):
    """Deterministic synthetic weights for `shapes` (default: the whole stack) as CPU tensors."""
    shapes = all_shapes(cfg) if shapes is None else shapes
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for key in sorted(shapes):
        shape = shapes[key]
        z = rng.standard_normal(size=shape, dtype=np.float32)
        std = _std_for(key, shape)
        arr = 1.0 + 0.1 * z if std is None else std * z
        sd[key] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(dtype)
    return sd


def round_to_bf16_(sd):
    """Round every tensor to bf16-representable values in place (kept as fp32): throughput-mode weights are
    stored in bf16, and any fp32 checker must see exactly those values for parity checks."""
    for k in sd:
        sd[k] = sd[k].to(torch.bfloat16).to(torch.float32)
    return sd


def make_state_dict_device(cfg, seed, device, dtype=torch.bfloat16, shapes=None):
    """Same inventory and per-tensor sigmas as make_state_dict, generated directly in HBM with torch's device
    generator (full-size 7B/13B models: no host copy, seconds instead of minutes). Not bit-identical to the
    PCG64 filler — full-size runs are checked by size-independent properties, not against CPU goldens."""
    shapes = all_shapes(cfg) if shapes is None else shapes
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd = OrderedDict()
    for key in sorted(shapes):
        shape = shapes[key]
        z = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        std = _std_for(key, shape)
        z = z.mul_(0.1).add_(1.0) if std is None else z.mul_(float(std))
        sd[key] = z.to(dtype)
        del z
    return sd
