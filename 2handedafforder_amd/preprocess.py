"""Host pre/post-processing contracts of the 2Haff CLIs (off the GPU hot path unless noted).

  SAM frame path  : ResizeLongestSide(1024).apply_image -> (x-mean)/std -> zero pad  (inference.py:91-105,244-256;
                    segment_anything/utils/transforms.py:27-34,102-113). For frames whose long side already equals
                    img_size the resize is the identity and the HIP ingest kernel (haff_patchify_u8) fuses the rest.
  CLIP frame path : CLIPImageProcessor defaults of openai/clip-vit-large-patch14 (shortest edge 224 bicubic,
                    centre crop, 1/255, mean/std) — SURVEY §9; values are that model card's public defaults.
  output gating   : on the device — postprocess.py (haff_gate_threshold_masks).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SAM_MEAN = (123.675, 116.28, 103.53)
SAM_STD = (58.395, 57.12, 57.375)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


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
    """ResizeLongestSide.apply_image on the HOST (data-loader workers, utils/aff_dataset.py:223-230): Pillow bilinear, as
    the reference (torchvision resize of a PIL image). The inference path does this on the device: FrameIngest.sam_frames."""
    from PIL import Image
    x = np.asarray(frame_u8_hwc)
    h, w = x.shape[:2]
    nh, nw = get_preprocess_shape(h, w, img_size)
    if (nh, nw) == (h, w):
        return torch.as_tensor(x)
    return torch.from_numpy(np.array(Image.fromarray(x).resize((nw, nh), Image.BILINEAR)))


def clip_preprocess(frame_u8_hwc, size=224):
    """CLIPImageProcessor.preprocess on the HOST (data-loader workers, utils/aff_dataset.py:76,228): Pillow bicubic to the
    shortest edge, centre crop, 1/255, mean/std. The inference path does this on the device: FrameIngest.clip_pixels."""
    from PIL import Image
    x = np.asarray(frame_u8_hwc)
    h, w = x.shape[:2]
    nh, nw = clip_resize_shape(h, w, size)
    if (nh, nw) != (h, w):
        x = np.array(Image.fromarray(x).resize((nw, nh), Image.BICUBIC))
    top, left = (nh - size) // 2, (nw - size) // 2
    crop = x[top:top + size, left:left + size]
    lut = clip_normalize_lut()
    return torch.from_numpy(np.stack([lut[c][crop[..., c]] for c in range(3)]))


# ----------------------------------------------------------------------------------------------------------------------
# PIL's antialiased resampling restated as integer tables (host side of the device frame ingest, rows a1/a2/f2).
# Both reference resizes run through Pillow on uint8 RGB: ResizeLongestSide.apply_image -> torchvision resize ->
# Image.resize(BILINEAR) (segment_anything/utils/transforms.py:27-34) and CLIPImageProcessor -> Image.resize(BICUBIC)
# (third-party transformers, call site inference.py:233-236). Pillow (src/libImaging/Resample.c: precompute_coeffs,
# normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc) convolves each axis with the filter stretched by the
# down-scale factor, coefficients normalised in double, rounded to 22-bit fixed point; each pass accumulates in int32 from
# 1 << 21, shifts by 22 and clamps to 0..255; horizontal pass first, uint8 intermediate. Integer work: bit-exact bar,
# pinned against Pillow itself in tests/test_preprocess_cpu.py (tables) and tests/test_preprocess_gpu.py (HIP kernels).
# ----------------------------------------------------------------------------------------------------------------------
PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_bilinear(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def _pil_bicubic(x):
    a = -0.5
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


PIL_FILTERS = {"bilinear": (_pil_bilinear, 1.0), "bicubic": (_pil_bicubic, 2.0)}


def pil_resample_tables(in_size, out_size, filt):
    """(bounds int32 [out,2] = (first input index, tap count), coeffs int32 [out,ksize]) of one axis."""
    fn, fsupport = PIL_FILTERS[filt]
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PIL_PRECISION_BITS)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = [fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            coeffs[xx, x] = int(-0.5 + k * one) if k < 0 else int(0.5 + k * one)
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs


def clip_resize_shape(h, w, size=224):
    """transformers get_resize_output_image_size(shortest_edge=size, default_to_square=False): the long side is floored."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def clip_normalize_lut():
    """float32 [3][256]: CLIPImageProcessor's rescale(1/255) then (x - mean) / std on one uint8 value, in numpy's own
    arithmetic (uint8 * python float -> float64 -> float32; float32 subtract and divide)."""
    v = (np.arange(256, dtype=np.uint8) * (1 / 255)).astype(np.float32)
    mean, std = np.array(CLIP_MEAN, dtype=np.float32), np.array(CLIP_STD, dtype=np.float32)
    return ((v[None, :] - mean[:, None]) / std[:, None]).astype(np.float32)


class FrameIngest:
    """uint8 NHWC frames in HBM -> what evaluate() consumes, all on the device (HIP kernels of csrc/frame_ingest.hip):
      sam_frames(frames)  = ResizeLongestSide(img_size).apply_image per frame (identity when the long side matches);
                            normalise + pad + patchify are fused downstream in haff_patchify_u8
      clip_pixels(frames) = CLIPImageProcessor.preprocess: shortest edge -> size (bicubic), centre crop, 1/255, mean/std
    Frames of one call share one size (a video batch); tables are built once per geometry and cached on the device."""

    def __init__(self, device):
        from . import ops
        self.ops, self.device = ops, torch.device(device)
        self._tables, self._lut = {}, None

    def _table(self, n_in, n_out, filt):
        key = (n_in, n_out, filt)
        if key not in self._tables:
            bounds, coeffs = pil_resample_tables(n_in, n_out, filt)
            assert int((bounds[:, 0] + bounds[:, 1]).max()) <= n_in
            self._tables[key] = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(coeffs).to(self.device))
        return self._tables[key]

    def resize(self, frames, out_hw, filt):
        """Image.resize((out_w, out_h), filt) of every frame: horizontal pass, uint8 intermediate, vertical pass."""
        B, H, W, _ = frames.shape
        oh, ow = out_hw
        x = frames.contiguous()
        if ow != W:
            x = self.ops.resample_u8(x, (H, ow), 0, *self._table(W, ow, filt))
        if oh != H:
            x = self.ops.resample_u8(x, (oh, ow), 1, *self._table(H, oh, filt))
        return x

    def sam_frames(self, frames, img_size):
        H, W = frames.shape[1:3]
        nh, nw = get_preprocess_shape(H, W, img_size)
        return self.resize(frames, (nh, nw), "bilinear"), (nh, nw)

    def clip_pixels(self, frames, size=224, dtype=torch.bfloat16):
        H, W = frames.shape[1:3]
        nh, nw = clip_resize_shape(H, W, size)
        x = self.resize(frames, (nh, nw), "bicubic")
        if self._lut is None:
            self._lut = torch.from_numpy(clip_normalize_lut()).to(self.device).contiguous()
        return self.ops.clip_normalize_u8(x, (nh - size) // 2, (nw - size) // 2, size, self._lut, dtype)
