"""Generates tests/golden/*.npz — run ONLY in the build container (needs /root/reference and transformers).

    python oracle/make_golden.py

What is captured (inputs + expected outputs + intermediate taps, NO weights: weights are rebuilt from
(config, seed) by 2handedafforder_amd/weights.py, whose keys are the reference's state-dict keys):

  * from the REFERENCE'S OWN modules, imported by file path from
    /root/reference/2Haff/model/segment_anything/modeling/ (pure torch; the package __init__ of
    segment_anything is bypassed because it pulls torchvision): ImageEncoderViT, PromptEncoder (text path +
    get_dense_pe), MaskDecoder left (taxonomy_on) / right, Sam.postprocess_masks.
  * from the container's `transformers` (the reference's Llama/CLIP arithmetic is that third-party library,
    pinned ==4.31.0 in 2Haff/requirements.txt:20 and absent from /root/reference; 5.x implements the same
    published math for Llama-1/2-style configs): LlamaModel + lm_head, CLIPVisionModel hidden_states[-2][:,1:].

Nothing from the reference is copied into the repo; fixtures are data only.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import haff  # noqa: E402,F401
from haff import config as hcfg  # noqa: E402
from haff import weights as hw  # noqa: E402

REF_MODELING = "/root/reference/2Haff/model/segment_anything/modeling"
OUT = os.path.join(ROOT, "tests", "golden")


def load_ref_modeling():
    spec = importlib.util.spec_from_file_location(
        "ref_sam_modeling", os.path.join(REF_MODELING, "__init__.py"), submodule_search_locations=[REF_MODELING])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_sam_modeling"] = mod
    spec.loader.exec_module(mod)
    return mod


def build_ref_sam(ref, s):
    """Same constructor arguments as build_sam.py:_build_sam (:59-117), with parametric sizes."""
    from functools import partial
    g = s.grid
    return ref.Sam(
        image_encoder=ref.ImageEncoderViT(
            depth=s.depth, embed_dim=s.embed_dim, img_size=s.img_size, mlp_ratio=s.mlp_ratio,
            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=s.heads, patch_size=s.patch, qkv_bias=True,
            use_rel_pos=True, global_attn_indexes=list(s.global_idx), window_size=s.window, out_chans=s.out_chans),
        prompt_encoder=ref.PromptEncoder(embed_dim=s.out_chans, image_embedding_size=(g, g),
                                         input_image_size=(s.img_size, s.img_size), mask_in_chans=16),
        mask_decoder_left=ref.MaskDecoder(
            num_multimask_outputs=3,
            transformer=ref.TwoWayTransformer(depth=2, embedding_dim=s.out_chans, mlp_dim=2048, num_heads=8),
            transformer_dim=s.out_chans, iou_head_depth=3, iou_head_hidden_dim=256, taxonomy_on=True),
        mask_decoder_right=ref.MaskDecoder(
            num_multimask_outputs=3,
            transformer=ref.TwoWayTransformer(depth=2, embedding_dim=s.out_chans, mlp_dim=2048, num_heads=8),
            transformer_dim=s.out_chans, iou_head_depth=3, iou_head_hidden_dim=256, taxonomy_on=False),
    ).eval()


def sam_goldens(ref, name, cfg, seed, n_prompts, input_size, original_size):
    s = cfg.sam
    shapes = hw.sam_shapes(s)
    sd = hw.make_state_dict(cfg, seed, shapes)
    sam = build_ref_sam(ref, s)
    missing, unexpected = sam.load_state_dict({k[len("model.visual_model."):]: v for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    # the only reference parameters our inventory omits are prompt types LISA never uses
    assert all(("point_embeddings" in m or "not_a_point" in m or "mask_downscaling" in m) for m in missing), missing
    rng = np.random.default_rng(seed + 1000)
    x = torch.from_numpy(rng.standard_normal((2, 3, s.img_size, s.img_size), dtype=np.float32))
    text = torch.from_numpy(rng.standard_normal((n_prompts, 1, s.out_chans), dtype=np.float32))
    taps = {}
    hooks = []
    for i, blk in enumerate(sam.image_encoder.blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"block{i}", o.detach().numpy())))
    hooks.append(sam.image_encoder.patch_embed.register_forward_hook(
        lambda m, a, o: taps.__setitem__("patch_proj", o.detach().numpy())))
    with torch.no_grad():
        emb = sam.image_encoder(x)
        sparse, dense = sam.prompt_encoder(points=None, boxes=None, masks=None, text_embeds=text)
        pe = sam.prompt_encoder.get_dense_pe()
        lo_l, iou_l, tax = sam.mask_decoder_left(image_embeddings=emb[0:1], image_pe=pe, sparse_prompt_embeddings=sparse,
                                                 dense_prompt_embeddings=dense, multimask_output=False)
        lo_r, iou_r = sam.mask_decoder_right(image_embeddings=emb[0:1], image_pe=pe, sparse_prompt_embeddings=sparse,
                                             dense_prompt_embeddings=dense, multimask_output=False)
        post_l = sam.postprocess_masks(lo_l, input_size=input_size, original_size=original_size)
        post_r = sam.postprocess_masks(lo_r, input_size=input_size, original_size=original_size)
    for h in hooks:
        h.remove()
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), seed=seed, images=x.numpy(), text_embeds=text.numpy(),
        image_embeddings=emb.numpy(), dense_pe=pe.numpy(), sparse=sparse.detach().numpy(), dense_row=dense[:, :, 0, 0].detach().numpy(),
        low_res_left=lo_l.numpy(), iou_left=iou_l.numpy(), taxonomy=tax.numpy(), low_res_right=lo_r.numpy(),
        iou_right=iou_r.numpy(), post_left=post_l.numpy(), post_right=post_r.numpy(),
        input_size=np.array(input_size), original_size=np.array(original_size),
        **{"tap_" + k: v for k, v in taps.items()})
    print(name, "emb", tuple(emb.shape), "low_res", tuple(lo_l.shape), "post", tuple(post_l.shape),
          "logit std %.3f" % lo_l.std().item())


def sam_vith_golden(ref, name, seed):
    """The reference's ImageEncoderViT at the REAL ViT-H block geometry (dim 1280, 16 heads of 80, mlp 5120, 14x14 windows on
    the 64x64 grid of a 1024^2 frame -> 5x5 padded windows, rel-pos tables of (27, 80) / (127, 80)), depth cut to 2 (one
    windowed + one global block), ONE frame. Output-only, and subsampled to stay small: every second position of every
    second row of all 256 channels (1 MiB) + per-channel float64 sums over the full map + per-block (mean, std, |max|).
    The input frame is not stored: the test regenerates it from the seed."""
    import copy
    cfg = copy.deepcopy(hcfg.haff_7b())
    cfg.sam.depth, cfg.sam.global_idx = 2, (1,)
    s = cfg.sam
    shapes = {k: v for k, v in hw.sam_shapes(s).items() if ".image_encoder." in k}
    sd = hw.make_state_dict(cfg, seed, shapes)
    from functools import partial
    enc = ref.ImageEncoderViT(
        depth=s.depth, embed_dim=s.embed_dim, img_size=s.img_size, mlp_ratio=s.mlp_ratio,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=s.heads, patch_size=s.patch, qkv_bias=True,
        use_rel_pos=True, global_attn_indexes=list(s.global_idx), window_size=s.window, out_chans=s.out_chans).eval()
    pre = "model.visual_model.image_encoder."
    missing, unexpected = enc.load_state_dict({k[len(pre):]: v for k, v in sd.items()}, strict=True)
    rng = np.random.default_rng(seed + 1000)
    x = torch.from_numpy(rng.standard_normal((1, 3, s.img_size, s.img_size), dtype=np.float32))
    stats = {}
    hooks = [blk.register_forward_hook(lambda m, a, o, i=i: stats.__setitem__(
        f"block{i}", np.array([o.mean().item(), o.std().item(), o.abs().max().item()], dtype=np.float64)))
        for i, blk in enumerate(enc.blocks)]
    with torch.no_grad():
        emb = enc(x)
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, depth=s.depth, global_idx=np.array(s.global_idx),
                        emb_sub=emb[:, :, ::2, ::2].numpy(), emb_channel_sums=emb.double().sum((0, 2, 3)).numpy(),
                        emb_abs_sum=emb.double().abs().sum().item(), **{"stat_" + k: v for k, v in stats.items()})
    print(name, "emb", tuple(emb.shape), "std %.4f" % emb.std().item(), {k: v.tolist() for k, v in stats.items()})


def llama_golden(name, cfg, seed):
    from transformers import LlamaConfig, LlamaForCausalLM
    l = cfg.llm
    hc = LlamaConfig(vocab_size=l.vocab, hidden_size=l.hidden, intermediate_size=l.ffn, num_hidden_layers=l.layers,
                     num_attention_heads=l.heads, num_key_value_heads=l.heads, rms_norm_eps=l.rms_eps,
                     rope_theta=l.rope_theta, max_position_embeddings=2048, attention_bias=False, mlp_bias=False,
                     tie_word_embeddings=False, attn_implementation="eager")
    model = LlamaForCausalLM(hc).eval()
    sd = hw.make_state_dict(cfg, seed, hw.llm_shapes(cfg))
    own = {k: v for k, v in sd.items() if not (k.startswith("model.mm_projector") or k.startswith("model.text_hidden"))}
    missing, unexpected = model.load_state_dict(own, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(seed + 2000)
    x = torch.from_numpy(rng.standard_normal((2, 40, l.hidden), dtype=np.float32))
    with torch.no_grad():
        out = model(inputs_embeds=x, output_hidden_states=True, use_cache=False)
        # KV-cached continuation: prefill 36, then 4 single-token steps
        o2 = model(inputs_embeds=x[:, :36], use_cache=True, output_hidden_states=True)
        hs = [o2.hidden_states[-1]]
        past = o2.past_key_values
        for t in range(36, 40):
            o3 = model(inputs_embeds=x[:, t:t + 1], past_key_values=past, use_cache=True, output_hidden_states=True)
            past = o3.past_key_values
            hs.append(o3.hidden_states[-1])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, inputs_embeds=x.numpy(),
                        hidden=out.hidden_states[-1].numpy(), layer0=out.hidden_states[1].numpy(),
                        logits=out.logits.numpy(), hidden_cached=torch.cat(hs, 1).numpy())
    print(name, "hidden", tuple(out.hidden_states[-1].shape), "cached-vs-full",
          (torch.cat(hs, 1) - out.hidden_states[-1]).abs().max().item())


def clip_golden(name, cfg, seed):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    c = cfg.clip
    hc = CLIPVisionConfig(hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers,
                          num_attention_heads=c.heads, image_size=c.image, patch_size=c.patch, hidden_act="quick_gelu",
                          layer_norm_eps=c.eps, attn_implementation="eager")
    model = CLIPVisionModel(hc).eval()
    sd = hw.make_state_dict(cfg, seed, hw.clip_shapes(c))
    pre = "model.vision_tower.vision_tower."
    own = {k[len(pre):]: v for k, v in sd.items()}
    if not any(k.startswith("vision_model.") for k in model.state_dict()):  # transformers 5.x flattened the prefix
        own = {k[len("vision_model."):]: v for k, v in own.items()}
    missing, unexpected = model.load_state_dict(own, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(seed + 3000)
    x = torch.from_numpy(rng.standard_normal((2, 3, c.image, c.image), dtype=np.float32))
    with torch.no_grad():
        out = model(x, output_hidden_states=True)
    feat = out.hidden_states[c.select_layer][:, 1:]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, images=x.numpy(), features=feat.numpy(),
                        hidden0=out.hidden_states[0].numpy())
    print(name, "features", tuple(feat.shape))


def host_goldens():
    """tokenizer_image_token (llava/mm_utils.py:19-44) with a stub tokenizer and conv_llava_v1.get_prompt()
    (llava/conversation.py) — pure-Python reference helpers, imported by path."""
    base = "/root/reference/2Haff/model/llava"
    pkg = importlib.util.module_from_spec(importlib.util.spec_from_file_location(
        "ref_llava", os.path.join(base, "__init__.py"), submodule_search_locations=[base]))
    sys.modules["ref_llava"] = pkg  # do NOT exec the package __init__ (it imports the model code)
    const = importlib.util.spec_from_file_location("ref_llava.constants", os.path.join(base, "constants.py"))
    cm = importlib.util.module_from_spec(const)
    sys.modules["ref_llava.constants"] = cm
    const.loader.exec_module(cm)
    spec = importlib.util.spec_from_file_location("ref_llava.mm_utils", os.path.join(base, "mm_utils.py"))
    mm = importlib.util.module_from_spec(spec)
    sys.modules["ref_llava.mm_utils"] = mm
    spec.loader.exec_module(mm)
    spec = importlib.util.spec_from_file_location("ref_llava.conversation", os.path.join(base, "conversation.py"))
    conv = importlib.util.module_from_spec(spec)
    sys.modules["ref_llava.conversation"] = conv
    spec.loader.exec_module(conv)

    class StubTok:
        bos_token_id = 1

        def __call__(self, text):
            class R:
                pass
            r = R()
            r.input_ids = [1] + [3 + (ord(ch) % 300) for ch in text]
            return r
    prompts = ["<im_start><image><im_end>\nWhere would you interact with the object to perform action open drawer",
               "no image here", "<image> leading", "a<image>b<image>c"]
    ids = [mm.tokenizer_image_token(p, StubTok()) for p in prompts]
    c = conv.conv_templates["llava_v1"].copy()
    c.messages = []
    c.append_message(c.roles[0], "<im_start><image><im_end>\nWhere would you hold the mug?")
    c.append_message(c.roles[1], "")
    # round 6: both templates the CLIs' --conv_type offers (chat.py:41-46,155; train_ds.py:115-120,188-190), in the three shapes
    # the path builds: an open turn (chat / evaluate), one closed round and two closed rounds (training conversations, whose
    # rounds utils/dataset.py:105 splits at sep2)
    by_type = {}
    for name in ("llava_v1", "llava_llama_2"):
        shapes = {}
        for tag, msgs in (("open_turn", [("<im_start><image><im_end>\nWhere would you hold the mug?", "")]),
                          ("one_round", [("<image>\nWhere would you interact with the object to perform action open drawer? Please output segmentation mask.", "[SEG].")]),
                          ("two_rounds", [("<image>\nfirst question", "first answer [SEG]."), ("second question", "It is [SEG].")])):
            t = conv.conv_templates[name].copy()
            t.messages = []
            for q, a in msgs:
                t.append_message(t.roles[0], q)
                t.append_message(t.roles[1], a)
            shapes[tag] = {"messages": [list(m) for m in msgs], "prompt": t.get_prompt()}
        t = conv.conv_templates[name]
        by_type[name] = {"roles": list(t.roles), "sep": t.sep, "sep2": t.sep2, "system": t.system, "shapes": shapes}
    # round 6: the two pure host helpers of row a1 — `preprocess` of inference.py:90-105 and ResizeLongestSide.get_preprocess_shape of
    # segment_anything/utils/transforms.py:102-113 — taken out of their files' syntax trees unchanged (inference.py is a script that
    # imports cv2 at the top, transforms.py imports torchvision: neither is installed) and evaluated on a grid of sizes / a small frame
    import ast
    from typing import Tuple
    ns = {"torch": torch, "F": torch.nn.functional, "Tuple": Tuple}
    p_inf = "/root/reference/2Haff/inference.py"
    fn = [n for n in ast.parse(open(p_inf).read(), filename=p_inf).body if isinstance(n, ast.FunctionDef) and n.name == "preprocess"]
    exec(compile(ast.Module(body=fn, type_ignores=[]), p_inf, "exec"), ns)
    p_tr = "/root/reference/2Haff/model/segment_anything/utils/transforms.py"
    rls = next(n for n in ast.parse(open(p_tr).read(), filename=p_tr).body if isinstance(n, ast.ClassDef) and n.name == "ResizeLongestSide")
    gps = [n for n in rls.body if isinstance(n, ast.FunctionDef) and n.name == "get_preprocess_shape"]
    gps[0].decorator_list = []          # (a @staticmethod: evaluated as a plain function)
    exec(compile(ast.Module(body=gps, type_ignores=[]), p_tr, "exec"), ns)
    sizes = [(1024, 1024), (480, 640), (640, 480), (1080, 1920), (333, 500), (1, 7), (2047, 2049), (1023, 1025), (225, 1000), (768, 1024)]
    shapes = [list(ns["get_preprocess_shape"](h, w, L)) for (h, w) in sizes for L in (1024, 224)]
    rng = np.random.default_rng(77)
    small = rng.integers(0, 256, size=(13, 20, 3), dtype=np.uint8)
    pre = ns["preprocess"](torch.from_numpy(small).permute(2, 0, 1).contiguous(), img_size=32)    # the call of inference.py:244-250
    # ... and the metric's own definition: calculate_iou / calculate_iocm of train_ds.py:761-800 (the script imports
    # deepspeed / cv2 at the top: the two definitions are taken out of its syntax tree), AverageMeter of utils/utils.py (importable)
    p_td = "/root/reference/2Haff/train_ds.py"
    fns = [n for n in ast.parse(open(p_td).read(), filename=p_td).body if isinstance(n, ast.FunctionDef) and n.name in ("calculate_iou", "calculate_iocm")]
    ns2 = {"np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), p_td, "exec"), ns2)
    mrng = np.random.default_rng(78)
    iou_cases = []
    for k, (pa, pb) in enumerate([(0.5, 0.5), (0.1, 0.9), (0.0, 0.3), (0.3, 0.0), (0.0, 0.0), (1.0, 1.0), (0.02, 0.02)]):
        a, b = mrng.random((24, 31)) < pa, mrng.random((24, 31)) < pb
        iou_cases.append({"seed_index": k, "p": [pa, pb], "iou": float(ns2["calculate_iou"](a, b)), "iocm": float(ns2["calculate_iocm"](a, b))})
    if "/root/reference/2Haff" not in sys.path:
        sys.path.insert(0, "/root/reference/2Haff")
    from utils.utils import AverageMeter as RefMeter
    m = RefMeter("MaskLoss", ":.4f")
    for v, n_ in ((0.5, 1), (0.25, 3), (1.0 / 3.0, 2)):
        m.update(v, n_)
    meter = {"str": str(m), "avg": m.avg, "sum": m.sum, "count": m.count}
    # ... and the 2HANDS question / answer templates (utils/aff_dataset.py:27-46; the module imports cv2 / h5py / pycocotools at the
    # top: its three constant assignments are evaluated out of the syntax tree)
    p_ad = "/root/reference/2Haff/utils/aff_dataset.py"
    assigns = [n for n in ast.parse(open(p_ad).read(), filename=p_ad).body if isinstance(n, ast.Assign)
               and getattr(n.targets[0], "id", "") in ("DEFAULT_IMAGE_TOKEN", "SHORT_QUESTION_LIST", "ANSWER_LIST")]
    ns3 = {}
    exec(compile(ast.Module(body=assigns, type_ignores=[]), p_ad, "exec"), ns3)
    templates = {"short_question_list": ns3["SHORT_QUESTION_LIST"], "answer_list": ns3["ANSWER_LIST"]}
    # ... and collate_fn ITSELF (utils/dataset.py:30-169: <image> -> <im_start><image><im_end>, padding, the label mask of every
    # round's instruction span under BOTH --conv_type values, truncation, the batch dict) — the module imports cv2 / pycocotools at the
    # top, its one function definition is taken out of the syntax tree and evaluated with the reference's own conversation_lib /
    # tokenizer_image_token / constants in scope, on the samples and the stand-in tokenizer of tests/golden_cases.py
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_cases as GC
    p_ds = "/root/reference/2Haff/utils/dataset.py"
    cf = [n for n in ast.parse(open(p_ds).read(), filename=p_ds).body if isinstance(n, ast.FunctionDef) and n.name == "collate_fn"]
    sys.path.insert(0, "/root/reference/2Haff")
    import utils.utils as ref_uu
    ns4 = {"torch": torch, "conversation_lib": conv, "tokenizer_image_token": mm.tokenizer_image_token,
           "DEFAULT_IMAGE_TOKEN": cm.DEFAULT_IMAGE_TOKEN, "IGNORE_INDEX": cm.IGNORE_INDEX, "IMAGE_TOKEN_INDEX": cm.IMAGE_TOKEN_INDEX,
           "DEFAULT_IM_START_TOKEN": ref_uu.DEFAULT_IM_START_TOKEN, "DEFAULT_IM_END_TOKEN": ref_uu.DEFAULT_IM_END_TOKEN}
    exec(compile(ast.Module(body=cf, type_ignores=[]), p_ds, "exec"), ns4)
    collate = {}
    for name in ("llava_v1", "llava_llama_2"):
        conv.default_conversation = conv.conv_templates[name]          # train_ds.py:188-190
        batch = GC.collate_samples(lambda: conv.conv_templates[name].copy())
        out = ns4["collate_fn"](batch, tokenizer=GC.StubSpTokenizer(), conv_type=name, use_mm_start_end=True, local_rank=-1)
        collate[name] = {"input_ids": out["input_ids"].tolist(), "labels": out["labels"].tolist(),
                         "attention_masks": out["attention_masks"].int().tolist(), "offset": out["offset"].tolist(),
                         "conversation_list": out["conversation_list"], "inference": bool(out["inference"]),
                         "taxonomies_list": out["taxonomies_list"].tolist(), "resize_list": [list(r) for r in out["resize_list"]],
                         "images_sum": float(out["images"].double().sum()), "keys": sorted(out.keys())}
    import json
    with open(os.path.join(OUT, "host_helpers.json"), "w") as f:
        json.dump({"prompts": prompts, "ids": ids, "conv_llava_v1_prompt": c.get_prompt(),
                   "roles": list(c.roles), "sep": c.sep, "sep2": c.sep2, "conv_templates": by_type,
                   "preprocess_shape_sizes": [list(t) for t in sizes], "preprocess_shapes_1024_224": shapes,
                   "preprocess_small_frame": small.tolist(), "preprocess_small_out": pre.numpy().tolist(),
                   "iou_cases": iou_cases, "average_meter": meter, "aff_templates": templates, "collate_fn": collate}, f, indent=1)
    print("host helpers ok:", c.get_prompt()[:80].replace("\n", "\\n"))


def llava_glue_golden(name, cfg, seed):
    """Rows a4-a6 through the REFERENCE'S OWN glue code (round 6): `CLIPVisionTower.forward / feature_select`
    (llava/model/multimodal_encoder/clip_encoder.py:31-60) and `LlavaMetaForCausalLM.encode_images /
    prepare_inputs_labels_for_multimodal` (llava/model/llava_arch.py:93-347), both imported by file path (the package's __init__ pulls
    llava_llama.py, whose AutoConfig.register fails under transformers 5.15 — llava_arch.py itself imports cleanly). The methods run
    untouched on duck-typed collaborators: a transformers CLIPVisionModel of the tiny geometry carrying the filler's weights placed in
    a CLIPVisionTower made without its network-fetching __init__, an nn.Embedding + nn.Linear projector with the filler's weights as
    the "model", a config object with mm_use_im_start_end = True (train_ds.py:76 / inference.py default). Inference-shaped rows
    (labels = None) and training-shaped rows (labels, a right-padded attention mask) are captured."""
    import types
    from transformers import CLIPVisionConfig, CLIPVisionModel
    base = "/root/reference/2Haff/model/llava/model"
    if "/root/reference/2Haff" not in sys.path:
        sys.path.insert(0, "/root/reference/2Haff")      # llava_arch.py: `from utils.utils import ...`
    pkg = importlib.util.module_from_spec(importlib.util.spec_from_file_location(
        "ref_llava_model", os.path.join(base, "__init__.py"), submodule_search_locations=[base]))
    sys.modules["ref_llava_model"] = pkg                 # NOT executed
    sub = types.ModuleType("ref_llava_model.multimodal_encoder")
    sub.__path__ = [os.path.join(base, "multimodal_encoder")]
    sys.modules["ref_llava_model.multimodal_encoder"] = sub
    spec = importlib.util.spec_from_file_location("ref_llava_model.llava_arch", os.path.join(base, "llava_arch.py"))
    arch = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = arch
    spec.loader.exec_module(arch)
    clip_mod = sys.modules["ref_llava_model.multimodal_encoder.clip_encoder"]

    c, l = cfg.clip, cfg.llm
    sd = hw.make_state_dict(cfg, seed, {**hw.clip_shapes(c), **{k: v for k, v in hw.llm_shapes(cfg).items()
                                                                 if k.startswith("model.mm_projector") or k == "model.embed_tokens.weight"}})
    hc = CLIPVisionConfig(hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers, num_attention_heads=c.heads,
                          image_size=c.image, patch_size=c.patch, hidden_act="quick_gelu", layer_norm_eps=c.eps,
                          attn_implementation="eager")
    vm = CLIPVisionModel(hc).eval()
    pre = "model.vision_tower.vision_tower."
    own = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    if not any(k.startswith("vision_model.") for k in vm.state_dict()):
        own = {k[len("vision_model."):]: v for k, v in own.items()}
    missing, unexpected = vm.load_state_dict(own, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    tower = clip_mod.CLIPVisionTower.__new__(clip_mod.CLIPVisionTower)      # its __init__ fetches a config from the hub
    torch.nn.Module.__init__(tower)
    tower.is_loaded, tower.vision_tower_name = True, "tiny-clip (seeded filler)"
    tower.select_layer, tower.select_feature, tower.vision_tower = c.select_layer, "patch", vm

    class Inner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embed_tokens = torch.nn.Embedding(l.vocab, l.hidden)
            self.mm_projector = torch.nn.Linear(c.hidden, l.hidden)
            self.embed_tokens.weight.data.copy_(sd["model.embed_tokens.weight"])
            self.mm_projector.weight.data.copy_(sd["model.mm_projector.weight"])
            self.mm_projector.bias.data.copy_(sd["model.mm_projector.bias"])
            self.vision_tower = tower

        def get_vision_tower(self):
            return self.vision_tower

    class Host(torch.nn.Module, arch.LlavaMetaForCausalLM):     # the reference's mixin; none of its methods is overridden
        def __init__(self):
            super().__init__()
            self.inner = Inner()
            self.config = types.SimpleNamespace(mm_use_im_start_end=True, tune_mm_mlp_adapter=False)

        def get_model(self):
            return self.inner

        @property
        def device(self):
            return torch.device("cpu")

    host = Host().eval()
    rng = np.random.default_rng(seed + 5000)
    B, Ltxt = 3, 9
    images = torch.from_numpy(rng.standard_normal((B, 3, c.image, c.image), dtype=np.float32))
    head = [cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]
    ids = torch.tensor([head + rng.integers(3, min(l.vocab, cfg.seg_token_idx) - 1, size=Ltxt).tolist() for _ in range(B)])
    ids[1, -3] = cfg.seg_token_idx
    with torch.no_grad():
        feats = host.encode_images(images)                        # clip_encoder.forward -> feature_select -> mm_projector
        # inference shape: HF generate hands an all-ones mask, no labels
        am = torch.ones_like(ids, dtype=torch.bool)
        _, am_out, _, emb, lab_none = host.prepare_inputs_labels_for_multimodal(ids, am, None, None, images)
        assert lab_none is None
        # training shape: labels with the instruction masked, right padding on two rows (collate_fn, utils/dataset.py:90-150)
        ids_t = ids.clone()
        labels = ids.clone()
        labels[:, :6] = -100
        am_t = torch.ones_like(ids, dtype=torch.bool)
        for b, n_pad in ((0, 2), (2, 4)):
            ids_t[b, -n_pad:] = cfg.pad_token_id
            labels[b, -n_pad:] = -100
            am_t[b, -n_pad:] = False
        _, am_t_out, _, emb_t, lab_t = host.prepare_inputs_labels_for_multimodal(ids_t, am_t, None, labels, images)
    # (the images are not stored: numpy's PCG64 stream of `seed + 5000` reproduces them, the first draw of that generator)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, images_checksum=float(images.double().sum()), input_ids=ids.numpy(),
                        image_features=feats.numpy(), attention_mask_out=am_out.numpy(), inputs_embeds=emb.numpy(),
                        input_ids_train=ids_t.numpy(), labels_train=labels.numpy(), attention_mask_train=am_t.numpy(),
                        attention_mask_train_out=am_t_out.numpy(), inputs_embeds_train=emb_t.numpy(), labels_train_out=lab_t.numpy())
    print(name, "features", tuple(feats.shape), "embeds", tuple(emb.shape), "labels", tuple(lab_t.shape))


def _load_ref_llava_arch():
    """llava_arch.py + multimodal_encoder/clip_encoder.py of the reference, by file path (the package __init__ is NOT executed: it pulls
    llava_llama.py, whose AutoConfig.register fails under transformers 5.15). -> (llava_arch module, clip_encoder module)."""
    import types
    base = "/root/reference/2Haff/model/llava/model"
    if "ref_llava_model.llava_arch" in sys.modules:
        return sys.modules["ref_llava_model.llava_arch"], sys.modules["ref_llava_model.multimodal_encoder.clip_encoder"]
    if "/root/reference/2Haff" not in sys.path:
        sys.path.insert(0, "/root/reference/2Haff")      # llava_arch.py: `from utils.utils import ...`
    pkg = importlib.util.module_from_spec(importlib.util.spec_from_file_location(
        "ref_llava_model", os.path.join(base, "__init__.py"), submodule_search_locations=[base]))
    sys.modules["ref_llava_model"] = pkg
    sub = types.ModuleType("ref_llava_model.multimodal_encoder")
    sub.__path__ = [os.path.join(base, "multimodal_encoder")]
    sys.modules["ref_llava_model.multimodal_encoder"] = sub
    spec = importlib.util.spec_from_file_location("ref_llava_model.llava_arch", os.path.join(base, "llava_arch.py"))
    arch = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = arch
    spec.loader.exec_module(arch)
    return arch, sys.modules["ref_llava_model.multimodal_encoder.clip_encoder"]


def _ref_clip_tower(clip_mod, cfg, sd):
    """The reference's CLIPVisionTower around a transformers CLIPVisionModel of the tiny geometry with the filler's weights (its
    __init__, which fetches a config from the hub, is bypassed)."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    c = cfg.clip
    hc = CLIPVisionConfig(hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers, num_attention_heads=c.heads,
                          image_size=c.image, patch_size=c.patch, hidden_act="quick_gelu", layer_norm_eps=c.eps,
                          attn_implementation="eager")
    vm = CLIPVisionModel(hc).eval()
    pre = "model.vision_tower.vision_tower."
    own = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    if not any(k.startswith("vision_model.") for k in vm.state_dict()):
        own = {k[len("vision_model."):]: v for k, v in own.items()}
    missing, unexpected = vm.load_state_dict(own, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    tower = clip_mod.CLIPVisionTower.__new__(clip_mod.CLIPVisionTower)
    torch.nn.Module.__init__(tower)
    tower.is_loaded, tower.vision_tower_name = True, "tiny-clip (seeded filler)"
    tower.select_layer, tower.select_feature, tower.vision_tower = c.select_layer, "patch", vm
    return tower


def llava_llama_forward_golden(name, cfg, seed):
    """`LlavaLlamaForCausalLM.forward` ITSELF (llava/model/language_model/llava_llama.py:55-135: multimodal splice -> LlamaModel ->
    lm_head -> shift-by-one CE; which hidden states it returns in training / eval mode) — its definition taken out of the file's
    syntax tree unchanged (the module cannot be imported: AutoConfig.register fails) and bound to a host that also carries the
    reference's LlavaMetaForCausalLM mixin (imported by path: the REAL encode_images / prepare_inputs_labels_for_multimodal) and whose
    `model` is transformers' LlamaModel of the tiny geometry with the filler's weights + the projector + the reference's
    CLIPVisionTower. Nothing of this repo's oracle runs inside the call: the whole language half of a training step and of an
    evaluation forward — reference glue around third-party models — is what the fixture holds."""
    import ast
    import types
    from typing import List, Optional, Tuple, Union
    from torch.nn import CrossEntropyLoss
    from transformers import LlamaConfig, LlamaModel
    from transformers.modeling_outputs import CausalLMOutputWithPast
    arch, clip_mod = _load_ref_llava_arch()
    path = "/root/reference/2Haff/model/llava/model/language_model/llava_llama.py"
    tree = ast.parse(open(path).read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "LlavaLlamaForCausalLM")
    fwd = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward"]
    ns = {"torch": torch, "Optional": Optional, "List": List, "Union": Union, "Tuple": Tuple,
          "CausalLMOutputWithPast": CausalLMOutputWithPast, "CrossEntropyLoss": CrossEntropyLoss}
    exec(compile(ast.Module(body=fwd, type_ignores=[]), path, "exec"), ns)

    c, l = cfg.clip, cfg.llm
    sd = hw.make_state_dict(cfg, seed, {**hw.clip_shapes(c), **hw.llm_shapes(cfg)})
    hc = LlamaConfig(vocab_size=l.vocab, hidden_size=l.hidden, intermediate_size=l.ffn, num_hidden_layers=l.layers,
                     num_attention_heads=l.heads, num_key_value_heads=l.heads, rms_norm_eps=l.rms_eps, rope_theta=l.rope_theta,
                     max_position_embeddings=2048, attention_bias=False, mlp_bias=False, tie_word_embeddings=False,
                     attn_implementation="eager")
    lm = LlamaModel(hc)
    own = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.layers.") or k in ("model.embed_tokens.weight", "model.norm.weight")}
    missing, unexpected = lm.load_state_dict(own, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    lm.mm_projector = torch.nn.Linear(c.hidden, l.hidden)
    lm.mm_projector.load_state_dict({"weight": sd["model.mm_projector.weight"], "bias": sd["model.mm_projector.bias"]})
    # (the tower is deliberately NOT registered as a child module of `lm`: transformers 5.x collects `output_hidden_states` through
    # forward hooks it installs on every submodule of the model being called — a CLIP model hanging under LlamaModel gets its layer
    # outputs re-recorded by the Llama call and returns a hidden_states tuple of another length the next time, so hidden_states[-2]
    # is another layer. Under the reference's transformers 4.31 the tuple is built by a plain loop and no such coupling exists.)
    tower = _ref_clip_tower(clip_mod, cfg, sd)
    lm.get_vision_tower = lambda: tower

    class Host(torch.nn.Module, arch.LlavaMetaForCausalLM):
        forward = ns["forward"]

        def __init__(self):
            super().__init__()
            self.model = lm
            self.lm_head = torch.nn.Linear(l.hidden, l.vocab, bias=False)
            self.lm_head.weight.data.copy_(sd["lm_head.weight"])
            self.config = types.SimpleNamespace(output_attentions=False, output_hidden_states=False, use_return_dict=True,
                                                vocab_size=l.vocab, mm_use_im_start_end=True, tune_mm_mlp_adapter=False)

        def get_model(self):
            return self.model

        @property
        def device(self):
            return torch.device("cpu")

    host = Host()
    rng = np.random.default_rng(seed + 8000)
    n, Ltxt = 3, 10
    images = torch.from_numpy(rng.standard_normal((n, 3, c.image, c.image), dtype=np.float32))
    head = [cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]
    ids = torch.tensor([head + rng.integers(3, 300, size=Ltxt).tolist() for _ in range(n)])
    ids[:, -3], ids[:, -1] = cfg.seg_token_idx, cfg.eos_token_id
    labels = ids.clone()
    labels[:, :9] = -100
    am = torch.ones_like(ids, dtype=torch.bool)
    with torch.no_grad():
        host.train()
        tr = host(images=images, attention_mask=am, input_ids=ids, labels=labels, output_hidden_states=True)
        host.eval()
        ev = host(images=images, attention_mask=am, input_ids=ids, output_hidden_states=True)
    assert isinstance(tr.hidden_states, tuple) and torch.is_tensor(ev.hidden_states) and ev.loss is None
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, input_ids=ids.numpy(), labels=labels.numpy(),
                        images_checksum=float(images.double().sum()), train_loss=np.float64(tr.loss.item()),
                        train_hidden_last=tr.hidden_states[-1].numpy(), train_logits_tail=tr.logits[:, -8:].numpy(),
                        train_logits_sum=np.float64(tr.logits.double().sum().item()), train_n_hidden=len(tr.hidden_states),
                        eval_hidden=ev.hidden_states.numpy(), eval_logits_tail=ev.logits[:, -8:].numpy())
    print(name, "CE", round(tr.loss.item(), 6), "hidden", tuple(ev.hidden_states.shape), "train hidden states", len(tr.hidden_states))


def greedy_generate_golden(name, cfg, seed):
    """The greedy loop the reference delegates to transformers (`self.generate(..., num_beams=1)`, LISA.py:443-450): argmax, append,
    a row that emitted EOS is padded from then on, the loop ends when every row has finished or at max_new_tokens. Third-party code,
    so the pin is transformers' own `generate` on a tiny LlamaForCausalLM carrying the filler's weights, fed the spliced embeddings
    (clip -> projector -> splice: pinned against the reference's glue by llava_glue_golden). Random weights never emit the real EOS:
    the EOS id of this case is the token row 0 produces at its third step (read from a first unconstrained run), so the rows stop
    at different steps."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from oracle import lisa_oracle as O
    l = cfg.llm
    sd = hw.make_state_dict(cfg, seed, {**hw.clip_shapes(cfg.clip), **hw.llm_shapes(cfg)})
    hc = LlamaConfig(vocab_size=l.vocab, hidden_size=l.hidden, intermediate_size=l.ffn, num_hidden_layers=l.layers,
                     num_attention_heads=l.heads, num_key_value_heads=l.heads, rms_norm_eps=l.rms_eps, rope_theta=l.rope_theta,
                     max_position_embeddings=2048, attention_bias=False, mlp_bias=False, tie_word_embeddings=False,
                     attn_implementation="eager")
    model = LlamaForCausalLM(hc).eval()
    own = {k: v for k, v in sd.items() if k.startswith("model.layers.") or k in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight")}
    missing, unexpected = model.load_state_dict(own, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(seed + 9000)
    B = 3
    images = torch.from_numpy(rng.standard_normal((B, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    head = [cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]
    ids = torch.tensor([head + rng.integers(3, 300, size=7).tolist() for _ in range(B)])
    with torch.no_grad():
        x = O.splice_embeddings(sd, ids, O.encode_images(sd, cfg, images))
        kw = dict(inputs_embeds=x, attention_mask=torch.ones(x.shape[:2], dtype=torch.long), do_sample=False, num_beams=1, use_cache=True)
        free = model.generate(max_new_tokens=8, eos_token_id=None, pad_token_id=cfg.pad_token_id, **kw)
        eos = int(free[0, 2])
        seq = model.generate(max_new_tokens=8, eos_token_id=eos, pad_token_id=cfg.pad_token_id, **kw)
        # row 0 alone: every row has finished after three tokens -> the loop ends early
        kw1 = dict(kw, inputs_embeds=x[:1], attention_mask=kw["attention_mask"][:1])
        seq1 = model.generate(max_new_tokens=8, eos_token_id=eos, pad_token_id=cfg.pad_token_id, **kw1)
    assert seq1.shape[1] == 3
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, input_ids=ids.numpy(), images_checksum=float(images.double().sum()),
                        free_tokens=free.numpy(), eos_token_id=eos, tokens=seq.numpy(), tokens_row0_alone=seq1.numpy())
    print(name, "free", free.tolist(), "eos", eos, "with eos", seq.tolist(), "row 0 alone", seq1.tolist())


def lisa_evaluate_golden(ref, name, cfg, seed):
    """`LISAForCausalLM.evaluate` and `get_visual_embs` THEMSELVES (model/LISA.py:432-534, :157-168) — the [SEG] row rule with its
    255-row shift, text_hidden_fcs, the cumsum split of the prompt embeddings over the samples, the per-sample prompt encoder / two
    mask decoders / postprocess loop and the `[:, 0]` slicing. LISA.py cannot be imported here (its first import, llava_llama.py,
    fails in AutoConfig.register under transformers 5.15: an ordinary error), so the two method definitions are taken out of its
    syntax tree UNCHANGED, compiled under LISA.py's own file name and bound to a host object whose collaborators are: the
    reference's Sam built from its own classes with the filler's weights (`model.visual_model`), `model.text_hidden_fcs` constructed
    as LISA.py:93-101 does, `seg_token_idx`, and a `generate()` that returns the output ids and last-step hidden states the CPU
    oracle produced for the same inputs (HF generate is third-party; Llama / CLIP are pinned against transformers separately).
    `Tensor.cuda()` is the identity for the duration of the call: the method moves two helper tensors to a GPU this container does
    not have. What is stored: the inputs evaluate() saw (ids, hidden states, frames by seed, sizes) and everything it returned."""
    import ast
    import types
    from oracle import lisa_oracle as O
    path = "/root/reference/2Haff/model/LISA.py"
    tree = ast.parse(open(path).read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "LISAForCausalLM")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("get_visual_embs", "evaluate")]
    assert [f.name for f in fns] == ["get_visual_embs", "evaluate"]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)

    s, l = cfg.sam, cfg.llm
    sd = hw.make_state_dict(cfg, seed)
    sam = build_ref_sam(ref, s)
    missing, unexpected = sam.load_state_dict({k[len("model.visual_model."):]: v for k, v in sd.items()
                                               if k.startswith("model.visual_model.")}, strict=False)
    assert not unexpected and all(("point_embeddings" in m or "not_a_point" in m or "mask_downscaling" in m) for m in missing)
    fcs = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Linear(l.hidden, l.hidden), torch.nn.ReLU(inplace=True),
                                                   torch.nn.Linear(l.hidden, cfg.out_dim), torch.nn.Dropout(0.0))]).eval()
    fcs[0][0].load_state_dict({"weight": sd["model.text_hidden_fcs.0.0.weight"], "bias": sd["model.text_hidden_fcs.0.0.bias"]})
    fcs[0][2].load_state_dict({"weight": sd["model.text_hidden_fcs.0.2.weight"], "bias": sd["model.text_hidden_fcs.0.2.bias"]})

    rng = np.random.default_rng(seed + 6000)
    B, S = 3, s.img_size
    images = torch.from_numpy(rng.standard_normal((B, 3, S, S), dtype=np.float32))
    images_clip = torch.from_numpy(rng.standard_normal((B, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    head = [cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]
    ids = torch.tensor([head + rng.integers(3, 300, size=8).tolist() for _ in range(B)])
    forced = torch.tensor([[7, cfg.seg_token_idx, 9, cfg.seg_token_idx, cfg.eos_token_id],     # two [SEG]
                           [5, 6, 7, 8, cfg.eos_token_id],                                     # none -> empty masks
                           [cfg.seg_token_idx, 8, 9, 11, cfg.eos_token_id]])                   # the first generated token
    resize_list = [(S, S), (S, S - 32), (S - 64, S)]
    original_size_list = [(S, S), (S // 2 + 3, S // 2 - 10), (150, 200)]
    with torch.no_grad():
        out_ids, hidden = O.lisa_generate(sd, cfg, images_clip, ids, 5, forced_answer=forced, use_cache=True)

    class Host:
        pass
    host = Host()
    host.seg_token_idx = cfg.seg_token_idx
    host.model = types.SimpleNamespace(visual_model=sam, text_hidden_fcs=fcs)
    host.get_visual_embs = types.MethodType(ns["get_visual_embs"], host)
    host.generate = lambda **kw: types.SimpleNamespace(sequences=out_ids, hidden_states=[hidden])
    cuda_was = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        r_ids, r_left, r_right, r_tax = ns["evaluate"](host, images_clip, images, ids, resize_list, original_size_list,
                                                       max_new_tokens=5, tokenizer=None)
    finally:
        torch.Tensor.cuda = cuda_was
    assert torch.equal(r_ids, out_ids) and [m.shape[0] for m in r_left] == [2, 0, 1]
    arrays = {"seed": seed, "input_ids": ids.numpy(), "forced": forced.numpy(), "output_ids": out_ids.numpy(), "hidden": hidden.numpy(),
              "resize_list": np.array(resize_list), "original_size_list": np.array(original_size_list),
              "images_checksum": float(images.double().sum())}
    for i in range(B):
        arrays[f"left{i}"], arrays[f"right{i}"], arrays[f"tax{i}"] = r_left[i].numpy(), r_right[i].numpy(), r_tax[i].numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(name, "masks per sample", [tuple(m.shape) for m in r_left], "taxonomy", [tuple(t.shape) for t in r_tax])


def lisa_model_forward_golden(ref, name, cfg, seed):
    """`LISAForCausalLM.model_forward` ITSELF (model/LISA.py:175-430) with the module's own `dice_loss` / `sigmoid_ce_loss` (:16-59):
    the training-time [SEG] row rule, the `offset` regrouping of the prompt embeddings, the per-sample decoders postprocessed to the
    LABEL shape, the taxonomy-weighted logits, the six losses and their normalisations — and the `inference=True` return. Same
    technique as lisa_evaluate_golden: the definitions are taken out of LISA.py's syntax tree unchanged and compiled under its file
    name; `super().forward(...)` (LlavaLlamaForCausalLM.forward -> HF Llama: third party, pinned against transformers separately)
    resolves to a base class whose forward returns the hidden states / logits / CE loss the CPU oracle computes for the same batch
    (encode_images -> splice -> llama_forward -> lm_head -> shift-by-one CE, llava_llama.py:93-118). `Tensor.cuda()` is the identity and
    stdout is swallowed for the call (the method prints "Training")."""
    import ast
    import contextlib
    import io
    import types
    from typing import List
    from oracle import lisa_oracle as O
    path = "/root/reference/2Haff/model/LISA.py"
    tree = ast.parse(open(path).read(), filename=path)
    losses = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("dice_loss", "sigmoid_ce_loss")]
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "LISAForCausalLM")
    meths = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("get_visual_embs", "model_forward")]
    assert len(losses) == 2 and [m.name for m in meths] == ["get_visual_embs", "model_forward"]
    base_src = ast.parse("class _LlmBase:\n    def forward(self, **kw):\n        return self._llm(**kw)\n").body[0]
    host_cls = ast.ClassDef(name="LISAForCausalLM", bases=[ast.Name(id="_LlmBase", ctx=ast.Load())], keywords=[], body=meths,
                            decorator_list=[])
    mod = ast.fix_missing_locations(ast.Module(body=losses + [base_src, host_cls], type_ignores=[]))
    ns = {"torch": torch, "F": torch.nn.functional, "nn": torch.nn, "List": List}
    exec(compile(mod, path, "exec"), ns)

    s, l = cfg.sam, cfg.llm
    sd = hw.make_state_dict(cfg, seed)
    sam = build_ref_sam(ref, s)
    missing, unexpected = sam.load_state_dict({k[len("model.visual_model."):]: v for k, v in sd.items()
                                               if k.startswith("model.visual_model.")}, strict=False)
    assert not unexpected and all(("point_embeddings" in m or "not_a_point" in m or "mask_downscaling" in m) for m in missing)
    fcs = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Linear(l.hidden, l.hidden), torch.nn.ReLU(inplace=True),
                                                   torch.nn.Linear(l.hidden, cfg.out_dim), torch.nn.Dropout(0.0))]).eval()
    fcs[0][0].load_state_dict({"weight": sd["model.text_hidden_fcs.0.0.weight"], "bias": sd["model.text_hidden_fcs.0.0.bias"]})
    fcs[0][2].load_state_dict({"weight": sd["model.text_hidden_fcs.0.2.weight"], "bias": sd["model.text_hidden_fcs.0.2.bias"]})

    rng = np.random.default_rng(seed + 7000)
    b, S, H0, W0 = 3, s.img_size, 120, 90
    images = torch.from_numpy(rng.standard_normal((b, 3, S, S), dtype=np.float32))
    images_clip = torch.from_numpy(rng.standard_normal((b, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    head = [cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx]
    # image 0 carries TWO conversations (offset 0, 2, 3, 4): the regrouping by `offset` is exercised
    ids = torch.tensor([head + rng.integers(3, 300, size=10).tolist() for _ in range(4)])
    ids[:, -3], ids[:, -1] = cfg.seg_token_idx, cfg.eos_token_id
    labels = ids.clone()
    labels[:, :9] = -100
    offset = torch.tensor([0, 2, 3, 4])
    n_masks = [2, 1, 1]
    masks_l = [(torch.from_numpy(rng.random((n, H0, W0))) > 0.5).float() for n in n_masks]
    masks_r = [(torch.from_numpy(rng.random((n, H0, W0))) > 0.6).float() for n in n_masks]
    tax = torch.tensor([[0.0, 0.0, 1.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]])
    batch = dict(images=images, images_clip=images_clip, input_ids=ids, labels=labels, attention_masks=torch.ones_like(ids, dtype=torch.bool),
                 offset=offset, masks_list_left=masks_l, masks_list_right=masks_r, taxonomies_list=tax,
                 label_list=[{"left": torch.zeros(H0, W0), "right": torch.zeros(H0, W0)} for _ in range(b)],
                 resize_list=[(S, S - 32), (S, S), (S - 64, S)], inference=False)

    def llm(images=None, attention_mask=None, input_ids=None, labels=None, output_hidden_states=True):
        img = O.encode_images(sd, cfg, images)
        hidden = O.llama_forward(sd, O.splice_embeddings(sd, input_ids, img), cfg.llm)
        logits = torch.nn.functional.linear(hidden, sd["lm_head.weight"])
        loss = None
        if labels is not None:
            lab = O.splice_labels(input_ids, labels)
            loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(), lab[:, 1:].reshape(-1), ignore_index=-100)
        # llava_llama.py:124-135: every layer's states (a tuple, the last one post-norm) in training mode, the post-norm tensor in eval
        return types.SimpleNamespace(hidden_states=(hidden,) if host.training else hidden, logits=logits, loss=loss)

    host = ns["LISAForCausalLM"]()
    host._llm = llm
    host.seg_token_idx = cfg.seg_token_idx
    host.model = types.SimpleNamespace(visual_model=sam, text_hidden_fcs=fcs)
    host.ce_loss = torch.nn.CrossEntropyLoss(reduction="mean")                # LISA.py:151
    host.ce_loss_weight, host.bce_loss_weight, host.dice_loss_weight = 1.0, 2.0, 0.5   # train_ds.py:92-94
    # every sample of this batch has ONE mask per decoder call except sample 0 (two conversations -> two [SEG] -> two masks): the
    # reference stacks the per-sample predictions, so the mask counts must agree; samples 1, 2 get their single mask repeated
    cuda_was = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    try:
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            # (a) equal mask counts: the three samples with one conversation each
            one = dict(batch, input_ids=ids[1:], labels=labels[1:], attention_masks=batch["attention_masks"][1:], offset=torch.tensor([0, 1, 2, 3]),
                       masks_list_left=[m[:1] for m in masks_l], masks_list_right=[m[:1] for m in masks_r])
            host.training = True
            out["train"] = host.model_forward(**one)
            # (b) inference=True on ONE image with two conversations (LISA.py:210-232 asserts a single CLIP image)
            inf = dict(batch, images=images[:1], images_clip=images_clip[:1], input_ids=ids[:2], labels=labels[:2],
                       attention_masks=batch["attention_masks"][:2], offset=torch.tensor([0, 2]), masks_list_left=masks_l[:1],
                       masks_list_right=masks_r[:1], taxonomies_list=tax[:1], label_list=batch["label_list"][:1],
                       resize_list=batch["resize_list"][:1], inference=True)
            host.training = False
            out["inference"] = host.model_forward(**inf)
    finally:
        torch.Tensor.cuda = cuda_was
    arrays = {"seed": seed, "input_ids": ids.numpy(), "labels": labels.numpy(), "taxonomies": tax.numpy(),
              "resize_list": np.array(batch["resize_list"]), "label_hw": np.array([H0, W0]),
              "images_checksum": float(images.double().sum())}
    for k, v in out["train"].items():
        arrays["train_" + k] = np.float64(v.item())
    for k in ("pred_masks_left", "pred_masks_right", "pred_taxonomies"):
        arrays["inference_" + k] = out["inference"][k].numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(name, {k: round(float(v), 6) for k, v in out["train"].items()}, "inference masks", tuple(out["inference"]["pred_masks_left"].shape))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ref = load_ref_modeling()
    sam_goldens(ref, "sam_tiny", hcfg.tiny(), seed=11, n_prompts=2, input_size=(224, 168), original_size=(120, 90))
    sam_goldens(ref, "sam_mid", hcfg.mid(), seed=12, n_prompts=1, input_size=(320, 320), original_size=(320, 320))
    sam_vith_golden(ref, "sam_vith_depth2", seed=15)
    llama_golden("llama_tiny", hcfg.tiny(), seed=13)
    clip_golden("clip_tiny", hcfg.tiny(), seed=14)
    host_goldens()
    llava_glue_golden("llava_glue_tiny", hcfg.tiny(), seed=16)
    lisa_evaluate_golden(ref, "lisa_evaluate_tiny", hcfg.tiny(), seed=17)
    lisa_model_forward_golden(ref, "lisa_model_forward_tiny", hcfg.tiny(), seed=18)
    llava_llama_forward_golden("llava_llama_forward_tiny", hcfg.tiny(), seed=19)
    greedy_generate_golden("greedy_generate_tiny", hcfg.tiny(), seed=20)


if __name__ == "__main__":
    main()
