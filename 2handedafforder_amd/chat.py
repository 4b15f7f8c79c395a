#!/usr/bin/env python3
"""Interactive CLI — same flags, prompts and output files as the reference's 2Haff/chat.py (:66-269): reads a text
prompt and an image path from stdin, wraps the prompt in the --conv_type template (llava_v1 / llava_llama_2, :155), runs LisaMI355.evaluate() on the
MI355X, prints the decoded text and writes <name>_mask_left<i>.jpg, <name>_mask_right<i>.jpg (mask*100) and the
red/blue overlay <name>_masked_img_<i>.jpg. A taxonomy argmax of 1 blanks the left mask, 0 the right one (:233-247).
"""
import os
import sys

import numpy as np
import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import haff  # noqa: F401
    from haff import postprocess, prompt as hprompt
    from haff.inference import build_model_and_tokenizer, load_rgb, parse_args, prepare_frame
else:
    from . import postprocess, prompt as hprompt
    from .inference import build_model_and_tokenizer, load_rgb, parse_args, prepare_frame

IMAGE_TOKEN_INDEX = -200


def render_outputs(image_np, mask_left, mask_right, taxonomy):
    """chat.py:226-269: gated mask planes (uint8, mask * 100 — what the reference hands to cv2.imwrite) from the device
    (haff_gate_threshold_masks: `mask > 0`, taxonomy argmax 1 blanks the left hand, 0 the right) and the overlay image."""
    left100 = postprocess.chat_plane(mask_left, taxonomy, "left").cpu().numpy()
    right100 = postprocess.chat_plane(mask_right, taxonomy, "right").cpu().numpy()
    left, right = left100 > 0, right100 > 0
    overlay = image_np.copy()
    overlay[left] = (image_np * 0.5 + left[:, :, None].astype(np.uint8) * np.array([255, 0, 0]) * 0.5)[left]
    overlay[right] = (image_np * 0.5 + right[:, :, None].astype(np.uint8) * np.array([0, 0, 255]) * 0.5)[right]
    return left100, right100, overlay


def main(argv, input_fn=input, max_turns=None):
    from PIL import Image
    args = parse_args(argv)
    os.makedirs(args.vis_save_path, exist_ok=True)
    model, tokenizer, cfg, dtype = build_model_and_tokenizer(args)
    turns = 0
    while max_turns is None or turns < max_turns:
        turns += 1
        prompt = hprompt.build_chat_prompt(input_fn("Please input your prompt: "), args.use_mm_start_end, args.conv_type)
        image_path = input_fn("Please input the image path: ")
        if not os.path.exists(image_path):
            print("File not found in {}".format(image_path))
            continue
        image_np = load_rgb(image_path)
        frames, resize_list, original_size_list = prepare_frame(image_np, cfg, dtype, model.device)
        input_ids = hprompt.tokenizer_image_token(prompt, tokenizer, return_tensors="pt").unsqueeze(0).to(model.device)
        output_ids, masks_left, masks_right, taxonomies = model.evaluate(
            None, None, input_ids, resize_list, original_size_list, max_new_tokens=args.max_new_tokens,
            tokenizer=tokenizer, frames_u8=frames)
        ids = output_ids[0][output_ids[0] != IMAGE_TOKEN_INDEX]
        text_output = tokenizer.decode(ids, skip_special_tokens=False).replace("\n", "").replace("  ", " ")
        print("text_output: ", text_output)
        stem = image_path.split("/")[-1].split(".")[0]
        for i, (ml, mr, tax) in enumerate(zip(masks_left, masks_right, taxonomies)):
            if ml.shape[0] == 0:
                continue
            left100, right100, overlay = render_outputs(image_np, ml[0], mr[0], tax)
            for name, arr in (("mask_left", left100), ("mask_right", right100)):
                save_path = "{}/{}_{}{}.jpg".format(args.vis_save_path, stem, name, i)
                Image.fromarray(arr).save(save_path)
                print("{} has been saved.".format(save_path))
            save_path = "{}/{}_masked_img_{}.jpg".format(args.vis_save_path, stem, i)
            Image.fromarray(overlay.astype(np.uint8)).save(save_path)
            print("{} has been saved.".format(save_path))


if __name__ == "__main__":
    main(sys.argv[1:])
