#!/usr/bin/env python3
"""Batch inference CLI — same flags, directory walk, prompt and output files as the reference's 2Haff/inference.py
(:20-49 flags, :199-334 loop), running the model on MI355X through LisaMI355.evaluate().

  python -m 2handedafforder_amd.inference ...   (or: python 2handedafforder_amd/inference.py ...)

Offline extras: --synthetic (seeded random weights + byte tokenizer, for plumbing runs without checkpoints).
Images are read/written with PIL (cv2 is not a dependency here); thresholds and file names follow :197,299,331.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import haff  # noqa: F401
    from haff import checkpoint, config as hcfg, postprocess, preprocess, prompt as hprompt
    from haff.lisa import LisaMI355
else:
    from . import checkpoint, config as hcfg, postprocess, preprocess, prompt as hprompt
    from .lisa import LisaMI355


def parse_args(args):
    parser = argparse.ArgumentParser(description="LISA chat")
    parser.add_argument("--version", default="sjauhri/2HAff")
    parser.add_argument("--vis_save_path", default="./vis_output", type=str)
    parser.add_argument("--precision", default="bf16", type=str, choices=["fp32", "bf16", "fp16"], help="precision for inference")
    parser.add_argument("--image_size", default=1024, type=int, help="image size")
    parser.add_argument("--model_max_length", default=512, type=int)
    parser.add_argument("--lora_r", default=8, type=int)
    parser.add_argument("--vision-tower", default="openai/clip-vit-large-patch14", type=str)
    parser.add_argument("--local-rank", default=0, type=int, help="node rank")
    parser.add_argument("--load_in_8bit", action="store_true", default=False)
    parser.add_argument("--load_in_4bit", action="store_true", default=False)
    parser.add_argument("--use_mm_start_end", action="store_true", default=True)
    parser.add_argument("--conv_type", default="llava_v1", type=str, choices=["llava_v1", "llava_llama_2"])
    parser.add_argument("--benchmark-dir", default=None, type=str, help="directory containing subfolders of benchmark examples")
    # MI355X / offline extras
    parser.add_argument("--synthetic", default=None, choices=["tiny", "mid", "7b", "13b"], help="random-init model of this geometry")
    parser.add_argument("--sam-checkpoint", default=None, type=str)
    parser.add_argument("--max-new-tokens", default=512, type=int)
    parser.add_argument("--batch-size", default=1, type=int,
                        help="frames per evaluate() call (the reference runs 1; prompts of different lengths are right-padded)")
    return parser.parse_args(args)


def build_model_and_tokenizer(args):
    if args.load_in_8bit or args.load_in_4bit or args.precision == "fp16":
        raise SystemExit("bitsandbytes 4/8-bit and the DeepSpeed fp16 kernel-inject mode are outside this build's scope "
                         "(SURVEY §2.2); use --precision bf16 or fp32")
    dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32
    device = f"cuda:{args.local_rank}"
    if args.synthetic:
        cfg = {"tiny": hcfg.tiny, "mid": hcfg.mid, "7b": hcfg.haff_7b, "13b": hcfg.haff_13b}[args.synthetic]()
        sd = checkpoint.synthetic_state_dict(cfg, 1234, device, dtype)
        tokenizer = checkpoint.ByteTokenizer(cfg)
    else:
        # inference.py:115-127,158-168: slow (sentencepiece) Llama tokenizer of the checkpoint + [SEG]; the model from
        # LISAForCausalLM.from_pretrained(version, vision_tower=, seg_token_idx=) — local directories (no hub here)
        sp_file = os.path.join(args.version, "tokenizer.model")
        if not os.path.isfile(sp_file):
            raise SystemExit(f"{sp_file} not found: --version must be a local merged-checkpoint directory")
        tokenizer = checkpoint.SentencePieceTokenizer(sp_file)
        seg = tokenizer("[SEG]", add_special_tokens=False).input_ids[0]
        clip_dir = args.vision_tower if os.path.isdir(args.vision_tower) else None
        model = LisaMI355.from_pretrained(args.version, vision_tower=clip_dir, seg_token_idx=seg, torch_dtype=dtype,
                                          sam_checkpoint=args.sam_checkpoint, device=device).eval()
        cfg = model.cfg
        cfg.bos_token_id, cfg.eos_token_id, cfg.pad_token_id = tokenizer.bos_token_id, tokenizer.eos_token_id, tokenizer.pad_token_id
        return model, tokenizer, cfg, dtype
    model = LisaMI355(cfg, sd, dtype=dtype, device=device).eval()
    return model, tokenizer, cfg, dtype


def load_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def prepare_frame(image_np, cfg, dtype, device):
    """inference.py:229-256 with the host work moved to the device: the uint8 RGB frame is uploaded once; evaluate()
    derives the CLIP tensor (CLIPImageProcessor.preprocess) and the resized SAM frame (ResizeLongestSide.apply_image,
    then normalise + pad fused into the patch-embedding ingest) from it with the Pillow-exact HIP kernels.
    Returns (frames_u8 [1,H0,W0,3] on the device, resize_list, original_size_list)."""
    original_size = tuple(image_np.shape[:2])
    frames = torch.from_numpy(np.array(image_np, copy=True)).unsqueeze(0).to(device)
    resize = preprocess.get_preprocess_shape(original_size[0], original_size[1], cfg.sam.img_size)
    return frames, [resize], [original_size]


def save_mask(path, plane_u8):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(plane_u8).save(path)
    print(f"{path} has been saved.")


def output_planes(masks_left, masks_right, taxonomies, thresholds=postprocess.THRESHOLDS):
    """inference.py:276-334 on the device: {(side, th): uint8 [H0,W0] 0/255} — gating from taxonomies[0] (one host read of
    its argmax decides which files exist), all five thresholds of a hand from ONE pass over its fp32 mask."""
    out = {}
    taxonomy = taxonomies[0]
    if taxonomy.numel() == 0:
        return out
    tax = taxonomy.reshape(-1).float().contiguous()
    t = int(torch.argmax(tax))
    for side, masks, blank in (("left", masks_left, 1), ("right", masks_right, 0)):
        if t == blank:
            continue
        for pred_mask in masks:
            if pred_mask.shape[0] == 0:
                continue
            planes = postprocess.inference_planes(pred_mask[0], None, side, thresholds).cpu().numpy()
            for k, th in enumerate(thresholds):
                out[(side, th)] = planes[k]
    return out


def pad_prompts(id_rows, pad_token_id):
    """collate_fn's rule (utils/dataset.py:90-93,144-150): right-pad with pad_token_id, mask = real positions."""
    L = max(r.numel() for r in id_rows)
    ids = torch.full((len(id_rows), L), pad_token_id, dtype=torch.long)
    mask = torch.zeros((len(id_rows), L), dtype=torch.bool)
    for b, r in enumerate(id_rows):
        ids[b, :r.numel()] = r
        mask[b, :r.numel()] = True
    return ids, mask


def iter_examples(benchmark_dir):
    """The directory walk of inference.py:199-219: <dir>/<folder>/{inpainting.png, annotation.json}, sorted."""
    for dir_name in sorted(os.listdir(benchmark_dir)):
        dir_path = os.path.join(benchmark_dir, dir_name)
        if not os.path.isdir(dir_path):
            continue
        for folder_name in sorted(os.listdir(dir_path)):
            folder_path = os.path.join(dir_path, folder_name)
            if not os.path.isdir(folder_path):
                continue
            image_path = os.path.join(folder_path, "inpainting.png")
            annotation_path = os.path.join(folder_path, "annotation.json")
            if not os.path.exists(image_path) or not os.path.exists(annotation_path):
                print(f"Required files not found in {folder_path}, skipping...")
                continue
            with open(annotation_path) as f:
                narration = json.load(f).get("narration", "")
            yield dir_name, folder_name, image_path, narration


def main(argv):
    args = parse_args(argv)
    model, tokenizer, cfg, dtype = build_model_and_tokenizer(args)
    device = model.device
    examples = list(iter_examples(args.benchmark_dir))
    for i in range(0, len(examples), max(args.batch_size, 1)):
        chunk = examples[i:i + max(args.batch_size, 1)]
        frames, resize_list, original_size_list, id_rows = [], [], [], []
        for _, _, image_path, narration in chunk:
            fr, rs, osz = prepare_frame(load_rgb(image_path), cfg, dtype, device)
            frames.append(fr[0])
            resize_list += rs
            original_size_list += osz
            prompt = hprompt.build_inference_prompt(narration, args.use_mm_start_end)
            id_rows.append(hprompt.tokenizer_image_token(prompt, tokenizer, return_tensors="pt"))
        input_ids, mask = pad_prompts(id_rows, cfg.pad_token_id)
        output_ids, masks_left, masks_right, taxonomies = model.evaluate(
            None, None, input_ids.to(device), resize_list, original_size_list, max_new_tokens=args.max_new_tokens,
            tokenizer=tokenizer, frames_u8=frames, attention_mask=mask)
        for b, (dir_name, folder_name, _, _) in enumerate(chunk):   # per frame, exactly the reference's B = 1 rule
            for (side, th), plane in output_planes(masks_left[b:b + 1], masks_right[b:b + 1], taxonomies[b:b + 1]).items():
                save_mask(os.path.join(args.vis_save_path + str(th), dir_name, folder_name, f"aff_{side}.png"), plane)


if __name__ == "__main__":
    main(sys.argv[1:])
