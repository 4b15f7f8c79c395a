"""Host pre/post-processing contracts of the 2Haff CLIs (off the GPU hot path unless noted).

  SAM frame path  : ResizeLongestSide(1024).apply_image -> (x-mean)/std -> zero pad  (inference.py:91-105,244-256;
                    segment_anything/utils/transforms.py:27-34,102-113). For frames whose long side already equals
                    img_size the resize is the identity and the HIP ingest kernel (haff_patchify_u8) fuses the rest.
  CLIP frame path : CLIPImageProcessor defaults of openai/clip-vit-large-patch14 (shortest edge 224 bicubic,
                    centre crop, 1/255, mean/std) — SURVEY §9; values are that model card's public defaults.
  output gating   : inference.py:276-334 / chat.py:226-253.
"""
import torch
import torch.nn.functional as F

SAM_MEAN = (123.675, 116.28, 103.53)
SAM_STD = (58.395, 57.12, 57.375)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
THRESHOLDS = (0.1, 0.2, 0.3, 0.5, 0.7)  # inference.py:197


def get_preprocess_shape(oldh, oldw, long_side):
    """transforms.py:102-113."""
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


def sam_preprocess(frame_u8_hwc, img_size=1024):
    """uint8 HWC RGB (long side already == img_size) -> float [3,img_size,img_size] (inference.preprocess)."""
    x = torch.as_tensor(frame_u8_hwc).permute(2, 0, 1).contiguous().float()
    x = (x - torch.tensor(SAM_MEAN).view(-1, 1, 1)) / torch.tensor(SAM_STD).view(-1, 1, 1)
    h, w = x.shape[-2:]
    return F.pad(x, (0, img_size - w, 0, img_size - h))


def resize_longest_side(frame_u8_hwc, img_size=1024):
    """ResizeLongestSide.apply_image. Identity when the long side already matches; otherwise bilinear
    (torch antialias) — the reference uses PIL's resize there; exact PIL parity is a §8f follow-up."""
    x = torch.as_tensor(frame_u8_hwc)
    h, w = x.shape[:2]
    nh, nw = get_preprocess_shape(h, w, img_size)
    if (nh, nw) == (h, w):
        return x
    y = F.interpolate(x.permute(2, 0, 1)[None].float(), (nh, nw), mode="bilinear", align_corners=False, antialias=True)
    return y[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8)


def clip_preprocess(frame_u8_hwc, size=224):
    """CLIPImageProcessor.preprocess equivalent: resize shortest edge (bicubic) -> centre crop -> normalise."""
    x = torch.as_tensor(frame_u8_hwc).permute(2, 0, 1)[None].float()
    h, w = x.shape[-2:]
    short = min(h, w)
    nh, nw = int(round(h * size / short)), int(round(w * size / short))
    if (nh, nw) != (h, w):
        x = F.interpolate(x, (nh, nw), mode="bicubic", align_corners=False, antialias=True).round().clamp(0, 255)
    top, left = (nh - size) // 2, (nw - size) // 2
    x = x[:, :, top:top + size, left:left + size] / 255.0
    return ((x - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD).view(1, 3, 1, 1))[0]


def gate_and_threshold(mask_left, mask_right, taxonomy, mode="chat", threshold=None):
    """taxonomy argmax==1 blanks the left hand, ==0 blanks the right (chat.py:233-247, inference.py:280-318).
    mode 'chat': logits > 0; mode 'inference': sigmoid(logits) > threshold."""
    t = int(torch.as_tensor(taxonomy).reshape(-1, 4)[0].argmax())

    def binarise(m):
        if mode == "chat":
            return m > 0
        return torch.sigmoid(m) > threshold
    left = binarise(mask_left) if t != 1 else torch.zeros_like(mask_left, dtype=torch.bool)
    right = binarise(mask_right) if t != 0 else torch.zeros_like(mask_right, dtype=torch.bool)
    return left, right, t
