"""ORACLE — test infrastructure only. Never imported by the product path (2handedafforder_amd/).

A plain fp32 CPU restatement (torch eager, functional, driven by a reference-keyed state dict) of the
2HandedAfforder per-frame affordance path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it, and only as the checker / reported baseline.

Pinning (see oracle/make_golden.py and tests/test_oracle_golden.py):
  * the SAM half is checked against outputs of the reference's OWN modules
    (/root/reference/2Haff/model/segment_anything/modeling/*.py, imported by path in the build container);
  * Llama / CLIP arithmetic lives in third-party `transformers` (pinned ==4.31.0 by 2Haff/requirements.txt:20,
    absent from /root/reference); it is checked against the container's transformers 5.x LlamaModel /
    CLIPVisionModel, which implement the same published math;
  * the LISA / LLaVA glue: `import model.LISA` fails here (transformers 4.31 internals, .cuda() calls), but since round 6 the
    reference's OWN definitions are what the fixtures come from — clip_encoder.py / llava_arch.py imported by file path
    (encode_images, prepare_inputs_labels_for_multimodal), LISAForCausalLM.evaluate / get_visual_embs / model_forward and
    dice_loss / sigmoid_ce_loss taken out of LISA.py's syntax tree unchanged and run on the reference's Sam classes with the
    third-party language-model calls served by this file's Llama / CLIP functions (oracle/make_golden.py: llava_glue_golden,
    lisa_evaluate_golden, lisa_model_forward_golden), and LlavaLlamaForCausalLM.forward (llava_llama.py:55-135) the same way over
    transformers' LlamaModel with no function of this file inside the call (llava_llama_forward_golden); the greedy loop against
    transformers' own generate (greedy_generate_golden). Restated-only now: prepare_inputs_for_generation (llava_llama.py:137-163).

Every function cites the reference lines it follows (paths relative to /root/reference/2Haff/).
"""
import math

import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200  # utils/utils.py:8
N_IMG_PAD = 255           # LISA.py:461 — 256 CLIP patch embeddings replace one sentinel

# ---- bf16-points mode (test infrastructure for the bf16 throughput path) ------------------------------------------
# The reference's own bf16 run (`--precision bf16`, model.bfloat16()) rounds every op's output to bf16. The HIP path rounds
# at FEWER points — once per fused kernel: a Linear's output after its fused bias / activation / residual, a norm's output,
# the attention probabilities handed to the P.V product, the attention output — and accumulates in fp32 everywhere. With
# `with bf16_points():` this oracle rounds at exactly those points (fp32 arithmetic in between; the SAM neck's last conv,
# the prompt encoder, both mask decoders, text_hidden_fcs and postprocess stay fp32 like LisaMI355(fp32_tail=True)), so the
# bf16 HIP path can be held to a few 1e-3 of the logit scale instead of the 6e-2 the distance to the exact fp32 forward
# needs. Default OFF: every golden fixture and every fp32 comparison sees the exact fp32 restatement.
_BF16_POINTS = False


class bf16_points:
    def __enter__(self):
        global _BF16_POINTS
        self._old, _BF16_POINTS = _BF16_POINTS, True
        return self

    def __exit__(self, *exc):
        global _BF16_POINTS
        _BF16_POINTS = self._old
        return False


def _r(x):
    """One kernel boundary of the bf16 path: round to bf16 (no-op in the exact mode)."""
    return x.to(torch.bfloat16).to(torch.float32) if _BF16_POINTS else x


def _lin(sd, name, x):
    """nn.Linear with optional bias."""
    b = sd.get(name + ".bias")
    return F.linear(x, sd[name + ".weight"], b)


def _ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _ln2d(sd, name, x, eps=1e-6):
    """LayerNorm2d over the channel dim of NCHW (segment_anything/modeling/common.py:31-43)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[name + ".weight"][:, None, None] * x + sd[name + ".bias"][:, None, None]


def _mlp(sd, name, x, n_layers):
    """mask_decoder.py:183-205 MLP: ReLU between layers, none after the last."""
    for i in range(n_layers):
        x = _lin(sd, f"{name}.layers.{i}", x)
        if i < n_layers - 1:
            x = F.relu(x)
    return x


# ----------------------------------------------------------------------------------------------------------
# SAM image encoder (segment_anything/modeling/image_encoder.py)
# ----------------------------------------------------------------------------------------------------------
def get_rel_pos(q_size, k_size, rel_pos):
    """image_encoder.py:321-351."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear")
        r = r.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        r = rel_pos
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return r[rel.long()]


def sam_attention(sd, pfx, x, num_heads):
    """image_encoder.py:235-260 with add_decomposed_rel_pos :354-392 (bias from the UNSCALED q)."""
    B, H, W, C = x.shape
    hd = C // num_heads
    qkv = _r(_lin(sd, pfx + ".qkv", x)).reshape(B, H * W, 3, num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * num_heads, H * W, -1).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    Rh = _r(get_rel_pos(H, H, sd[pfx + ".rel_pos_h"]))   # (the tables are bf16 MFMA operands in the kernels)
    Rw = _r(get_rel_pos(W, W, sd[pfx + ".rel_pos_w"]))
    r_q = q.reshape(B * num_heads, H, W, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
    attn = (attn.view(-1, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(-1, H * W, H * W)
    attn = _r(attn.softmax(dim=-1))
    x = _r(attn @ v).view(B, num_heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return _lin(sd, pfx + ".proj", x)   # (+ bias + shortcut before the one rounding: sam_block)


def sam_block(sd, pfx, x, num_heads, window):
    """image_encoder.py:177-193; window partition pads with zeros AFTER norm1 (:179-183, :276-288)."""
    shortcut = x
    x = _r(_ln(sd, pfx + ".norm1", x, 1e-6))
    if window > 0:
        B, H, W, C = x.shape
        ph, pw = (window - H % window) % window, (window - W % window) % window
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        x = x.view(B, Hp // window, window, Wp // window, window, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, window, window, C)
    x = sam_attention(sd, pfx + ".attn", x, num_heads)
    if window > 0:
        x = x.view(B, Hp // window, Wp // window, window, window, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
        x = x[:, :H, :W, :]
    x = _r(shortcut + x)
    h = _r(_ln(sd, pfx + ".norm2", x, 1e-6))
    h = _lin(sd, pfx + ".mlp.lin2", _r(F.gelu(_lin(sd, pfx + ".mlp.lin1", h))))  # common.py:13-26
    return _r(x + h)


def sam_image_encoder(sd, pfx, x, cfg, taps=None):
    """ImageEncoderViT.forward (image_encoder.py:110-125). x [B,3,S,S] -> [B,out_chans,S/16,S/16]."""
    x = _r(F.conv2d(_r(x), sd[pfx + ".patch_embed.proj.weight"], sd[pfx + ".patch_embed.proj.bias"], stride=cfg.patch))
    x = _r(x.permute(0, 2, 3, 1) + _r(sd[pfx + ".pos_embed"]))
    if taps is not None:
        taps["patch_embed"] = x.clone()
    for i in range(cfg.depth):
        win = 0 if i in cfg.global_idx else cfg.window
        x = sam_block(sd, f"{pfx}.blocks.{i}", x, cfg.heads, win)
        if taps is not None:
            taps[f"block{i}"] = x.clone()
    x = x.permute(0, 3, 1, 2)
    x = _r(F.conv2d(x, sd[pfx + ".neck.0.weight"]))
    x = _r(_ln2d(sd, pfx + ".neck.1", x))
    x = F.conv2d(x, sd[pfx + ".neck.2.weight"], padding=1)   # fp32 from here on (the decoder tail's input)
    x = _ln2d(sd, pfx + ".neck.3", x)
    return x


# ----------------------------------------------------------------------------------------------------------
# SAM prompt encoder text path + dense PE (segment_anything/modeling/prompt_encoder.py)
# ----------------------------------------------------------------------------------------------------------
def sam_dense_pe(sd, pfx, grid_hw):
    """PromptEncoder.get_dense_pe -> PositionEmbeddingRandom.forward (prompt_encoder.py:67-76,203-229)."""
    h, w = grid_hw
    G = sd[pfx + ".pe_layer.positional_encoding_gaussian_matrix"]
    grid = torch.ones((h, w), dtype=G.dtype)
    y = (grid.cumsum(0) - 0.5) / h
    x = (grid.cumsum(1) - 0.5) / w
    c = 2 * torch.stack([x, y], dim=-1) - 1
    c = 2 * math.pi * (c @ G)
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1).permute(2, 0, 1).unsqueeze(0)


def sam_prompt_encoder_text(sd, pfx, text_embeds, grid_hw):
    """PromptEncoder.forward(points=None, boxes=None, masks=None, text_embeds=) (prompt_encoder.py:140-186)."""
    bs = text_embeds.shape[0]
    sparse = torch.cat([torch.empty((bs, 0, text_embeds.shape[-1])), text_embeds], dim=1)
    dense = sd[pfx + ".no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(bs, -1, grid_hw[0], grid_hw[1])
    return sparse, dense


# ----------------------------------------------------------------------------------------------------------
# SAM mask decoder + two-way transformer (mask_decoder.py, transformer.py)
# ----------------------------------------------------------------------------------------------------------
def _dec_attention(sd, pfx, q, k, v, num_heads):
    """transformer.py:185-242."""
    q, k, v = _lin(sd, pfx + ".q_proj", q), _lin(sd, pfx + ".k_proj", k), _lin(sd, pfx + ".v_proj", v)

    def sep(x):
        b, n, c = x.shape
        return x.reshape(b, n, num_heads, c // num_heads).transpose(1, 2)
    q, k, v = sep(q), sep(k), sep(v)
    attn = (q @ k.permute(0, 1, 3, 2)) / math.sqrt(q.shape[-1])
    out = torch.softmax(attn, dim=-1) @ v
    b, h, n, c = out.shape
    return _lin(sd, pfx + ".out_proj", out.transpose(1, 2).reshape(b, n, h * c))


def sam_two_way_transformer(sd, pfx, image_embedding, image_pe, point_embedding, depth=2, num_heads=8, taps=None):
    """TwoWayTransformer.forward (transformer.py:62-106) and TwoWayAttentionBlock.forward (:151-182)."""
    keys = image_embedding.flatten(2).permute(0, 2, 1)
    key_pe = image_pe.flatten(2).permute(0, 2, 1)
    queries, query_pe = point_embedding, point_embedding
    for i in range(depth):
        L = f"{pfx}.layers.{i}"
        if i == 0:  # skip_first_layer_pe
            queries = _dec_attention(sd, L + ".self_attn", queries, queries, queries, num_heads)
        else:
            q = queries + query_pe
            queries = queries + _dec_attention(sd, L + ".self_attn", q, q, queries, num_heads)
        queries = _ln(sd, L + ".norm1", queries, 1e-5)
        q, k = queries + query_pe, keys + key_pe
        queries = queries + _dec_attention(sd, L + ".cross_attn_token_to_image", q, k, keys, num_heads)
        queries = _ln(sd, L + ".norm2", queries, 1e-5)
        queries = queries + _lin(sd, L + ".mlp.lin2", F.relu(_lin(sd, L + ".mlp.lin1", queries)))
        queries = _ln(sd, L + ".norm3", queries, 1e-5)
        q, k = queries + query_pe, keys + key_pe
        keys = keys + _dec_attention(sd, L + ".cross_attn_image_to_token", k, q, queries, num_heads)
        keys = _ln(sd, L + ".norm4", keys, 1e-5)
        if taps is not None:
            taps[f"layer{i}.queries"] = queries.clone()
            taps[f"layer{i}.keys"] = keys.clone()
    q, k = queries + query_pe, keys + key_pe
    queries = queries + _dec_attention(sd, pfx + ".final_attn_token_to_image", q, k, keys, num_heads)
    queries = _ln(sd, pfx + ".norm_final_attn", queries, 1e-5)
    return queries, keys


def sam_mask_decoder(sd, pfx, image_embeddings, image_pe, sparse, dense, taxonomy_on, taps=None):
    """MaskDecoder.forward / predict_masks with multimask_output=False (mask_decoder.py:79-178)."""
    n = sparse.size(0)
    out_tok = torch.cat([sd[pfx + ".iou_token.weight"], sd[pfx + ".mask_tokens.weight"]], dim=0)
    tokens = torch.cat((out_tok.unsqueeze(0).expand(n, -1, -1), sparse), dim=1)
    src = torch.repeat_interleave(image_embeddings, n, dim=0) + dense
    pos = torch.repeat_interleave(image_pe, n, dim=0)
    b, c, h, w = src.shape
    hs, src = sam_two_way_transformer(sd, pfx + ".transformer", src, pos, tokens, taps=taps)
    iou_tok, mask_toks = hs[:, 0, :], hs[:, 1:5, :]
    src = src.transpose(1, 2).view(b, c, h, w)
    up = F.conv_transpose2d(src, sd[pfx + ".output_upscaling.0.weight"], sd[pfx + ".output_upscaling.0.bias"], stride=2)
    up = F.gelu(_ln2d(sd, pfx + ".output_upscaling.1", up))
    up = F.gelu(F.conv_transpose2d(up, sd[pfx + ".output_upscaling.3.weight"], sd[pfx + ".output_upscaling.3.bias"], stride=2))
    if taps is not None:
        taps["upscaled"] = up.clone()
    hyper = torch.stack([_mlp(sd, f"{pfx}.output_hypernetworks_mlps.{i}", mask_toks[:, i, :], 3) for i in range(4)], dim=1)
    if taps is not None:
        taps["hyper"] = hyper.clone()
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, 4, h, w)
    iou = _mlp(sd, pfx + ".iou_prediction_head", iou_tok, 3)
    masks, iou = masks[:, 0:1], iou[:, 0:1]
    if taxonomy_on:
        tax = F.softmax(_mlp(sd, pfx + ".taxonomy_embed", mask_toks.flatten(1), 3), dim=-1)
        return masks, iou, tax
    return masks, iou


def sam_postprocess_masks(masks, img_size, input_size, original_size):
    """Sam.postprocess_masks (sam.py:155-189)."""
    masks = F.interpolate(masks.float(), (img_size, img_size), mode="bilinear", align_corners=False)
    masks = masks[..., : input_size[0], : input_size[1]]
    return F.interpolate(masks, tuple(original_size), mode="bilinear", align_corners=False)


# ----------------------------------------------------------------------------------------------------------
# CLIP vision tower (third-party transformers CLIPVisionModel; call site multimodal_encoder/clip_encoder.py:41-60)
# ----------------------------------------------------------------------------------------------------------
def clip_vision_features(sd, pfx, x, cfg):
    """hidden_states[select_layer][:, 1:] of CLIPVisionModel (feature_select, clip_encoder.py:31-39)."""
    vm = pfx + ".vision_model"
    p = _r(F.conv2d(_r(x), sd[vm + ".embeddings.patch_embedding.weight"], stride=cfg.patch)).flatten(2).transpose(1, 2)
    cls = sd[vm + ".embeddings.class_embedding"].expand(x.shape[0], 1, -1)
    h = _r(torch.cat([cls, p], dim=1) + sd[vm + ".embeddings.position_embedding.weight"][None])
    h = _r(_ln(sd, vm + ".pre_layrnorm", h, cfg.eps))
    n_run = cfg.layers + 1 + cfg.select_layer if cfg.select_layer < 0 else cfg.select_layer
    hd = cfg.hidden // cfg.heads
    for i in range(n_run):
        L = f"{vm}.encoder.layers.{i}"
        r = h
        y = _r(_ln(sd, L + ".layer_norm1", h, cfg.eps))
        B, N, C = y.shape
        q = (_r(_lin(sd, L + ".self_attn.q_proj", y)) * hd ** -0.5).view(B, N, cfg.heads, hd).transpose(1, 2)
        k = _r(_lin(sd, L + ".self_attn.k_proj", y)).view(B, N, cfg.heads, hd).transpose(1, 2)
        v = _r(_lin(sd, L + ".self_attn.v_proj", y)).view(B, N, cfg.heads, hd).transpose(1, 2)
        a = _r(_r(torch.softmax(q @ k.transpose(-1, -2), dim=-1)) @ v)
        h = _r(r + _lin(sd, L + ".self_attn.out_proj", a.transpose(1, 2).reshape(B, N, C)))
        y = _lin(sd, L + ".mlp.fc1", _r(_ln(sd, L + ".layer_norm2", h, cfg.eps)))
        y = _r(y * torch.sigmoid(1.702 * y))  # quick_gelu (fused into fc1's epilogue: one rounding)
        h = _r(h + _lin(sd, L + ".mlp.fc2", y))
    return h[:, 1:]


# ----------------------------------------------------------------------------------------------------------
# Llama decoder (third-party transformers LlamaModel; call site language_model/llava_llama.py:93-105)
# ----------------------------------------------------------------------------------------------------------
def _rms(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def _rope(pos, d, theta):
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    f = pos.float()[:, None] * inv[None, :]
    e = torch.cat([f, f], dim=-1)
    return e.cos(), e.sin()


def _rot_half(x):
    return torch.cat([-x[..., x.shape[-1] // 2:], x[..., : x.shape[-1] // 2]], dim=-1)


def llama_forward(sd, x, cfg, cache=None, taps=None):
    """x [B,T,H] input embeddings; cache: optional list of (k,v) per layer, extended in place.
    Returns post-final-norm hidden [B,T,H] (what llava_llama.py:124-135 hands back in eval) ."""
    B, T, H = x.shape
    hd = H // cfg.heads
    past = 0 if cache is None or cache[0] is None else cache[0][0].shape[2]
    pos = torch.arange(past, past + T)
    cos, sin = _rope(pos, hd, cfg.rope_theta)
    for i in range(cfg.layers):
        L = f"model.layers.{i}"
        h = _r(_rms(x, sd[L + ".input_layernorm.weight"], cfg.rms_eps))
        q = _r(F.linear(h, sd[L + ".self_attn.q_proj.weight"])).view(B, T, cfg.heads, hd).transpose(1, 2)
        k = _r(F.linear(h, sd[L + ".self_attn.k_proj.weight"])).view(B, T, cfg.heads, hd).transpose(1, 2)
        v = _r(F.linear(h, sd[L + ".self_attn.v_proj.weight"])).view(B, T, cfg.heads, hd).transpose(1, 2)
        q = _r(q * cos + _rot_half(q) * sin)   # (the RoPE kernel rewrites q and the cached k in bf16)
        k = _r(k * cos + _rot_half(k) * sin)
        if cache is not None:
            if cache[i] is not None:
                k = torch.cat([cache[i][0], k], dim=2)
                v = torch.cat([cache[i][1], v], dim=2)
            cache[i] = (k, v)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        Tk = k.shape[2]
        mask = torch.arange(Tk)[None, :] > (torch.arange(T)[:, None] + (Tk - T))
        s = s.masked_fill(mask, float("-inf"))
        a = _r(_r(torch.softmax(s.float(), dim=-1)) @ v)
        x = _r(x + F.linear(a.transpose(1, 2).reshape(B, T, H), sd[L + ".self_attn.o_proj.weight"]))
        h = _r(_rms(x, sd[L + ".post_attention_layernorm.weight"], cfg.rms_eps))
        g = _r(F.silu(F.linear(h, sd[L + ".mlp.gate_proj.weight"])) * F.linear(h, sd[L + ".mlp.up_proj.weight"]))
        x = _r(x + F.linear(g, sd[L + ".mlp.down_proj.weight"]))
        if taps is not None:
            taps[f"layer{i}"] = x.clone()
    return _r(_rms(x, sd["model.norm.weight"], cfg.rms_eps))


# ----------------------------------------------------------------------------------------------------------
# LLaVA glue (model/llava/model/llava_arch.py)
# ----------------------------------------------------------------------------------------------------------
def encode_images(sd, cfg, images_clip):
    """encode_images (llava_arch.py:93-96): CLIP patch features -> Linear projector (:35)."""
    f = clip_vision_features(sd, "model.vision_tower.vision_tower", images_clip, cfg.clip)
    return _r(_lin(sd, "model.mm_projector", f))


def splice_embeddings(sd, input_ids, image_features):
    """prepare_inputs_labels_for_multimodal, mm_use_im_start_end branch (llava_arch.py:185-208,235-256):
    [embed(ids[:p]) ; image features ; embed(ids[p+1:p+2]) ; embed(ids[p+2:])]."""
    E = sd["model.embed_tokens.weight"]
    rows = []
    for b in range(input_ids.shape[0]):
        ids = input_ids[b]
        p = int(torch.where(ids == IMAGE_TOKEN_INDEX)[0][0])
        rows.append(torch.cat([E[ids[:p]], image_features[b], E[ids[p + 1: p + 2]], E[ids[p + 2:]]], dim=0))
    return torch.stack(rows, dim=0)


def splice_labels(input_ids, labels, n_img=N_IMG_PAD + 1):
    """The labels of prepare_inputs_labels_for_multimodal, mm_use_im_start_end branch (llava_arch.py:195-208,247-249): IGNORE_INDEX
    over the n_img spliced image rows, every other label where its id went. Pinned against the reference's own method
    (tests/golden/llava_glue_tiny.npz)."""
    lab = []
    for b in range(input_ids.shape[0]):
        p = int(torch.where(input_ids[b] == IMAGE_TOKEN_INDEX)[0][0])
        lab.append(torch.cat([labels[b, :p], torch.full((n_img,), -100, dtype=labels.dtype), labels[b, p + 1:]]))
    return torch.stack(lab)


def splice_attention_mask(attention_mask, n_img=N_IMG_PAD + 1):
    """... and its attention mask (llava_arch.py:332-345, equal-length rows): n_img - 1 True columns on the LEFT of the caller's mask."""
    pad = torch.ones((attention_mask.shape[0], n_img - 1), dtype=attention_mask.dtype)
    return torch.cat([pad, attention_mask], dim=1)


def seg_token_mask(output_ids, seg_token_idx, n_pad=N_IMG_PAD):
    """LISA.py:457-465: mask[:, 255 + j] is set iff token j+1 is [SEG]."""
    m = output_ids[:, 1:] == seg_token_idx
    return torch.cat([torch.zeros((m.shape[0], n_pad), dtype=torch.bool), m], dim=1)


def text_hidden_fcs(sd, h):
    """LISA.py:95-101: Linear -> ReLU -> Linear -> Dropout(0)."""
    return _lin(sd, "model.text_hidden_fcs.0.2", F.relu(_lin(sd, "model.text_hidden_fcs.0.0", h)))


# ----------------------------------------------------------------------------------------------------------
# LISAForCausalLM.evaluate (model/LISA.py:432-534)
# ----------------------------------------------------------------------------------------------------------
def _stack_ctx(points, stack):
    """bf16 points for ONE stack of the path ("sam" / "clip" / "llama"): points=None leaves the mode as the caller set it."""
    import contextlib
    return bf16_points() if points is not None and stack in points else contextlib.nullcontext()


def lisa_generate(sd, cfg, images_clip, input_ids, max_new_tokens, forced_answer=None, use_cache=False, points=None):
    """Greedy generate (LISA.py:443-450). use_cache=False follows the reference exactly: config.use_cache=False
    (LISA.py:115) so EVERY step re-runs CLIP + projector + the full sequence (llava_llama.py:82-102).
    use_cache=True is the numerically equivalent KV-cached schedule. forced_answer [B,n] overrides the
    appended tokens (argmax is still computed) — synthetic random-init models never emit [SEG]/EOS.
    points (test infrastructure): a subset of {"clip", "llama"} switches the bf16-points mode on for that stack only.
    Returns (output_ids [B,L+N], hidden [B,T+N-1,H] of the last step)."""
    B = input_ids.shape[0]
    out_ids = input_ids.clone()
    finished = torch.zeros(B, dtype=torch.bool)
    hidden_all, cache = None, None
    img = None
    for step in range(max_new_tokens):
        if not use_cache or step == 0:
            with _stack_ctx(points, "clip"):
                img = encode_images(sd, cfg, images_clip)
        with _stack_ctx(points, "llama"):
            if not use_cache:
                hidden_all = llama_forward(sd, splice_embeddings(sd, out_ids, img), cfg.llm)
            elif step == 0:
                cache = [None] * cfg.llm.layers
                hidden_all = llama_forward(sd, splice_embeddings(sd, out_ids, img), cfg.llm, cache)
            else:
                h_new = llama_forward(sd, sd["model.embed_tokens.weight"][out_ids[:, -1:]], cfg.llm, cache)
                hidden_all = torch.cat([hidden_all, h_new], dim=1)
        logits = F.linear(hidden_all[:, -1], sd["lm_head.weight"])
        nxt = logits.argmax(-1)
        if forced_answer is not None:
            nxt = forced_answer[:, step].clone()
        nxt = torch.where(finished, torch.full_like(nxt, cfg.pad_token_id), nxt)
        out_ids = torch.cat([out_ids, nxt[:, None]], dim=1)
        finished = finished | (nxt == cfg.eos_token_id)
        if bool(finished.all()):
            break
    return out_ids, hidden_all


def lisa_evaluate(sd, cfg, images_clip, images, input_ids, resize_list, original_size_list, max_new_tokens=32,
                  forced_answer=None, use_cache=False, taps=None, points=None, memo=None):
    """LISAForCausalLM.evaluate (LISA.py:432-534) -> (output_ids, pred_masks_left, pred_masks_right, taxonomies).
    Test infrastructure on top of the reference's call: `points` (a subset of {"sam", "clip", "llama"}) runs ONE OR MORE stacks
    in the bf16-points mode and the rest exactly (attribution of the bf16 path's distance to the stacks), and `memo` (a dict the
    caller keeps across calls on the SAME inputs) re-uses the generate() result per (clip, llama) setting and the image
    embedding per sam setting, so that an attribution at full depth costs one stage per variant instead of one frame."""
    on = (lambda s: points is not None and s in points)
    kg, ks = ("gen", on("clip"), on("llama")), ("sam", on("sam"))
    if memo is not None and kg in memo:
        output_ids, hidden = memo[kg]
    else:
        output_ids, hidden = lisa_generate(sd, cfg, images_clip, input_ids, max_new_tokens, forced_answer, use_cache, points)
        if memo is not None:
            memo[kg] = (output_ids, hidden)
    mask = seg_token_mask(output_ids, cfg.seg_token_idx)
    last = text_hidden_fcs(sd, hidden)
    pred = last[mask]
    counts = mask.int().sum(-1)
    offs = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(-1)], dim=0)
    pred_embeddings = [pred[offs[i]: offs[i + 1]] for i in range(len(offs) - 1)]
    V = "model.visual_model"
    if memo is not None and ks in memo:
        emb = memo[ks]
    else:
        with _stack_ctx(points, "sam"):
            emb = torch.cat([sam_image_encoder(sd, V + ".image_encoder", images[i: i + 1], cfg.sam) for i in range(images.shape[0])], 0)
        if memo is not None:
            memo[ks] = emb
    grid = (cfg.sam.img_size // cfg.sam.patch,) * 2
    pe = sam_dense_pe(sd, V + ".prompt_encoder", grid)
    if taps is not None:
        taps.update({"hidden": hidden, "pred_embeddings": pred_embeddings, "image_embeddings": emb})
    left, right, tax = [], [], []
    for i in range(len(pred_embeddings)):
        sparse, dense = sam_prompt_encoder_text(sd, V + ".prompt_encoder", pred_embeddings[i].unsqueeze(1), grid)
        lo_l, _, t = sam_mask_decoder(sd, V + ".mask_decoder_left", emb[i: i + 1], pe, sparse, dense, True)
        left.append(sam_postprocess_masks(lo_l, cfg.sam.img_size, resize_list[i], original_size_list[i])[:, 0])
        tax.append(t)
        lo_r, _ = sam_mask_decoder(sd, V + ".mask_decoder_right", emb[i: i + 1], pe, sparse, dense, False)
        right.append(sam_postprocess_masks(lo_r, cfg.sam.img_size, resize_list[i], original_size_list[i])[:, 0])
    return output_ids, left, right, tax


# ----------------------------------------------------------------------------------------------------------
# host pre/post-processing contracts
# ----------------------------------------------------------------------------------------------------------
SAM_MEAN = (123.675, 116.28, 103.53)  # inference.py:93-94
SAM_STD = (58.395, 57.12, 57.375)


def sam_preprocess(frame_u8_hwc, img_size):
    """inference.preprocess (inference.py:91-105) for a frame already resized so its long side == img_size."""
    x = torch.from_numpy(frame_u8_hwc).permute(2, 0, 1).contiguous().float() if not torch.is_tensor(frame_u8_hwc) \
        else frame_u8_hwc.permute(2, 0, 1).contiguous().float()
    x = (x - torch.tensor(SAM_MEAN).view(-1, 1, 1)) / torch.tensor(SAM_STD).view(-1, 1, 1)
    h, w = x.shape[-2:]
    return F.pad(x, (0, img_size - w, 0, img_size - h))


def inference_output_planes(pred_masks_left, pred_masks_right, taxonomies, th_list=(0.1, 0.2, 0.3, 0.5, 0.7)):
    """Output gating + thresholds of inference.py:276-334: {(side, th): uint8 [H0,W0] of 0/255} = the arrays handed to
    cv2.imwrite (a hand whose gate is closed writes nothing -> absent key). taxonomy = taxonomies[0] (:276), argmax over
    the flattened tensor (:278,305); sigmoid in fp32 on the fp32 mask, numpy compare against the python float th
    (float32 under both numpy casting regimes)."""
    import numpy as np
    out = {}
    taxonomy = taxonomies[0]
    if taxonomy.numel() == 0:
        return out
    t = int(torch.argmax(taxonomy))
    for side, masks, blank in (("left", pred_masks_left, 1), ("right", pred_masks_right, 0)):
        if t == blank:
            continue
        for pred_mask in masks:
            if pred_mask.shape[0] == 0:
                continue
            prob = torch.sigmoid(pred_mask).detach().cpu().numpy()[0]
            for th in th_list:
                th_pred = np.zeros_like(prob)
                th_pred[prob > th] = 255
                out[(side, th)] = th_pred.astype(np.uint8)
    return out


def chat_output_planes(pred_mask_left, pred_mask_right, taxonomy):
    """chat.py:226-253 for one prompt group: (mask_left * 100, mask_right * 100) as uint8; `mask > 0`, argmax == 1 blanks
    the left hand, == 0 the right (both files are always written)."""
    import numpy as np
    left = pred_mask_left.detach().cpu().numpy()[0] > 0
    if int(torch.argmax(taxonomy)) == 1:
        left = np.zeros_like(left)
    right = pred_mask_right.detach().cpu().numpy()[0] > 0
    if int(torch.argmax(taxonomy)) == 0:
        right = np.zeros_like(right)
    return (left * 100).astype(np.uint8), (right * 100).astype(np.uint8)


def mask_iou(a, b):
    """train_ds.py:761-776 == ActAffordance/scripts/evaluation/calculate_iou.py:26-41."""
    inter = (a & b).sum().item()
    union = (a | b).sum().item()
    return inter / union if union > 0 else 0.0


# ----------------------------------------------------------------------------------------------------------
# Training forward + losses: LISAForCausalLM.model_forward (model/LISA.py:175-430), trainable set of
# train_ds.py:192-244 (LoRA r/alpha on q_proj,v_proj + embed_tokens, lm_head, text_hidden_fcs, both mask decoders)
# ----------------------------------------------------------------------------------------------------------
def dice_loss(inputs, targets, num_masks, scale=1000, eps=1e-6):
    """LISA.py:16-39."""
    inputs = inputs.sigmoid().flatten(1, 2)
    targets = targets.flatten(1, 2)
    numerator = 2 * (inputs / scale * targets).sum(-1)
    denominator = (inputs / scale).sum(-1) + (targets / scale).sum(-1)
    loss = 1 - (numerator + eps) / (denominator + eps)
    return loss.sum() / (num_masks + 1e-8)


def sigmoid_ce_loss(inputs, targets, num_masks):
    """LISA.py:42-59."""
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.flatten(1, 2).mean(1).sum() / (num_masks + 1e-8)


def with_lora(sd, lora, cfg, alpha=16.0):
    """Effective q/v weights W + (alpha/r) B A (peft LoRA, train_ds.py:217-231; dropout omitted = eval / p=0)."""
    if not lora:
        return sd
    out = dict(sd)
    for i in range(cfg.llm.layers):
        for n in ("q_proj", "v_proj"):
            k = f"model.layers.{i}.self_attn.{n}"
            A, B = lora[k + ".lora_A"], lora[k + ".lora_B"]
            out[k + ".weight"] = sd[k + ".weight"] + (alpha / A.shape[0]) * (B @ A)
    return out


def lisa_model_forward(sd, cfg, batch, lora=None, lora_alpha=16.0, ce_loss_weight=1.0, dice_loss_weight=0.5,
                       bce_loss_weight=2.0):
    """model_forward (LISA.py:175-430). `batch` has the keys of collate_fn (utils/dataset.py:152-169)."""
    sdw = with_lora(sd, lora, cfg, lora_alpha)
    images, images_clip = batch["images"], batch["images_clip"]
    input_ids, labels, offset = batch["input_ids"], batch["labels"], batch["offset"]
    inference = batch.get("inference", False)
    V = "model.visual_model"
    with torch.no_grad():
        emb = torch.cat([sam_image_encoder(sd, V + ".image_encoder", images[i: i + 1], cfg.sam) for i in range(images.shape[0])], 0)
    bsz = emb.shape[0]
    assert bsz == len(offset) - 1
    m = input_ids[:, 1:] == cfg.seg_token_idx
    m = torch.cat([m, torch.zeros((m.shape[0], 1), dtype=torch.bool)], dim=1)
    m = torch.cat([torch.zeros((m.shape[0], N_IMG_PAD), dtype=torch.bool), m], dim=1)
    # images_clip expanded per conversation (LISA.py:235-245)
    clip_rep = torch.cat([images_clip[i: i + 1].expand(int(offset[i + 1] - offset[i]), -1, -1, -1) for i in range(bsz)], 0)
    with torch.no_grad():
        img = encode_images(sd, cfg, clip_rep)  # vision tower + projector are frozen (train_ds.py:183-186)
    x = splice_embeddings(sdw, input_ids, img)
    hidden = llama_forward(sdw, x, cfg.llm)
    logits = F.linear(hidden, sdw["lm_head.weight"])
    lab = splice_labels(input_ids, labels)
    ce = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(), lab[:, 1:].reshape(-1), ignore_index=-100)
    last = text_hidden_fcs(sdw, hidden)
    pred = last[m]
    counts = m.int().sum(-1)
    seg_off = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(-1)], 0)[offset]
    pred_embeddings = [pred[seg_off[i]: seg_off[i + 1]] for i in range(len(seg_off) - 1)]
    grid = (cfg.sam.grid, cfg.sam.grid)
    pe = sam_dense_pe(sd, V + ".prompt_encoder", grid)
    pl, pr, pt = [], [], []
    for i in range(bsz):
        sparse, dense = sam_prompt_encoder_text(sdw, V + ".prompt_encoder", pred_embeddings[i].unsqueeze(1), grid)
        lo_l, _, t = sam_mask_decoder(sdw, V + ".mask_decoder_left", emb[i: i + 1], pe, sparse, dense, True)
        pt.append(t)
        pl.append(sam_postprocess_masks(lo_l, cfg.sam.img_size, batch["resize_list"][i], batch["label_list"][i]["left"].shape)[:, 0])
        lo_r, _ = sam_mask_decoder(sdw, V + ".mask_decoder_right", emb[i: i + 1], pe, sparse, dense, False)
        pr.append(sam_postprocess_masks(lo_r, cfg.sam.img_size, batch["resize_list"][i], batch["label_list"][i]["right"].shape)[:, 0])
    gt_l = torch.stack(batch["masks_list_left"], 0)
    gt_r = torch.stack(batch["masks_list_right"], 0)
    pl, pr, pt = torch.stack(pl, 0), torch.stack(pr, 0), torch.stack(pt)
    gt_tax = batch["taxonomies_list"]
    if inference:
        return {"pred_masks_left": pl, "pred_masks_right": pr, "pred_taxonomies": pt, "gt_masks_left": gt_l,
                "gt_masks_right": gt_r, "gt_taxonomies": gt_tax}
    w_left, w_right, w_both = gt_tax[:, 0], gt_tax[:, 1], gt_tax[:, 2] + gt_tax[:, 3]
    pl = (w_left.view(-1, 1, 1, 1) + w_both.view(-1, 1, 1, 1)) * pl
    pr = (w_right.view(-1, 1, 1, 1) + w_both.view(-1, 1, 1, 1)) * pr
    bce_l = bce_r = dice_l = dice_r = tax_ce = 0
    num_masks = 0
    for i in range(len(pl)):
        n = gt_l[i].shape[0]
        bce_l = bce_l + sigmoid_ce_loss(pl[i], gt_l[i], n) * n
        dice_l = dice_l + dice_loss(pl[i], gt_l[i], n) * n
        bce_r = bce_r + sigmoid_ce_loss(pr[i], gt_r[i], n) * n
        dice_r = dice_r + dice_loss(pr[i], gt_r[i], n) * n
        num_masks += n
        tax_ce = tax_ce + F.cross_entropy(pt[i], gt_tax[i].unsqueeze(0).float())
    tax_ce = tax_ce / len(pl)
    mask_bce = bce_loss_weight * bce_l / (num_masks + 1e-8) + bce_loss_weight * bce_r / (num_masks + 1e-8)
    mask_dice = dice_loss_weight * dice_l / (num_masks + 1e-8) + dice_loss_weight * dice_r / (num_masks + 1e-8)
    ce = ce * ce_loss_weight
    mask_loss = mask_bce + mask_dice
    return {"loss": ce + mask_loss + tax_ce, "ce_loss": ce, "taxonomy_ce_loss": tax_ce, "mask_bce_loss": mask_bce,
            "mask_dice_loss": mask_dice, "mask_loss": mask_loss}
