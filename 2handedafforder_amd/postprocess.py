"""Output gating + thresholds of the reference's CLIs (row a15) on the device.

  inference.py:276-334 : argmax(taxonomy) != 1 -> emit left, != 0 -> emit right; per hand
                         `sigmoid(mask) > th` for th in (.1, .2, .3, .5, .7) -> 0/255 planes (cv2.imwrite PNGs)
  chat.py:226-253      : `mask > 0`; a blanked hand is still written (all zeros); planes are written as mask*100

Byte work, so the bar is bit-exact. `sigmoid(m) > th` is evaluated by the reference in fp32 (torch.sigmoid on the
fp32 mask, numpy compare against float32(th)); fp32 sigmoid is monotone, so the rule equals `m > x*(th)` for ONE fp32
constant x*(th) = the largest float with sigmoid_f32(x) <= float32(th). The constants below were found by bisection
over the fp32 ordering against torch.sigmoid (tests/test_abi_cpu.py re-derives them and checks the +-64-ulp
neighbourhood); the HIP kernel (haff_gate_threshold_masks) then only compares, reading each mask once for all
thresholds, and applies the taxonomy gate from the device-resident class probabilities.
"""
import struct

import torch

from . import ops

THRESHOLDS = (0.1, 0.2, 0.3, 0.5, 0.7)  # inference.py:197


def _f32_from_bits(b):
    return struct.unpack("<f", struct.pack("<I", b & 0xFFFFFFFF))[0]


def _ordered_to_f32(u):
    """inverse of the order-preserving map float32 -> uint32 (negative floats bit-flipped, positives offset)"""
    return _f32_from_bits(u ^ 0x80000000 if u & 0x80000000 else ~u)


def derive_sigmoid_logit_threshold(th):
    """Largest fp32 x with torch.sigmoid(float32(x)) <= float32(th), by bisection over the fp32 total order."""
    th32 = torch.tensor(th, dtype=torch.float32)

    def le(u):
        x = torch.full((16,), _ordered_to_f32(u), dtype=torch.float32)  # a full SIMD vector: the vectorised code path
        return bool((torch.sigmoid(x) <= th32).all())
    lo, hi = (~0xC2C80000) & 0xFFFFFFFF, 0x42C80000 | 0x80000000   # -100.0 .. +100.0 in ordered space
    assert le(lo) and not le(hi)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if le(mid):
            lo = mid
        else:
            hi = mid
    return _ordered_to_f32(lo)


# x*(th) as fp32 bit patterns (derive_sigmoid_logit_threshold on torch 2.10 CPU; checked by tests/test_abi_cpu.py)
_LOGIT_TH_BITS = {0.1: 0xC00C9F54, 0.2: 0xBFB17218, 0.3: 0xBF58E882, 0.5: 0x33C00000, 0.7: 0x3F58E884}
_DERIVED = {}


def sigmoid_logit_threshold(th):
    """x*(th). Note x*(0.5) = 2^-23.4 > 0: sigmoid_f32 rounds to exactly 0.5 for 0 < m <= x*, so `sigmoid(m) > 0.5` is
    not `m > 0` — the chat rule (mask > 0) and the inference rule at th = 0.5 differ on those logits."""
    key = float(th)
    if key in _LOGIT_TH_BITS:
        return _f32_from_bits(_LOGIT_TH_BITS[key])
    if key not in _DERIVED:
        _DERIVED[key] = derive_sigmoid_logit_threshold(key)
    return _DERIVED[key]


def inference_planes(mask_logits, taxonomy, side, thresholds=THRESHOLDS):
    """One hand of one prompt, inference.py rule. mask_logits fp32 [H,W] on the device, taxonomy fp32 [4] on the device.
    Returns uint8 [len(thresholds), H, W] of 0/255 (all zero when the taxonomy gate closes this hand)."""
    ths = [sigmoid_logit_threshold(t) for t in thresholds]
    tax = None if taxonomy is None else taxonomy.reshape(-1)[:4].float().contiguous()   # None: the caller gated already
    return ops.gate_threshold_masks(mask_logits.contiguous(), ths, 255, tax, 1 if side == "left" else 0)


def chat_plane(mask_logits, taxonomy, side, on_value=100):
    """One hand of one prompt, chat.py rule: (mask > 0) * 100, zeros when the gate closes. Returns uint8 [H,W].
    The gate is the reference's: torch.argmax over the WHOLE per-frame taxonomy tensor [n_prompts, 4] flattened
    (chat.py:231,243) — index 1 blanks the left hand, index 0 the right one, any other index (>= 4 can only happen with more
    than one [SEG] in the answer) blanks neither. One host read of an index per frame; chat is interactive, not the hot path."""
    k = int(torch.argmax(taxonomy.reshape(-1)))
    if k == (1 if side == "left" else 0):
        return torch.zeros(mask_logits.shape[-2:], dtype=torch.uint8, device=mask_logits.device)
    return ops.gate_threshold_masks(mask_logits.contiguous(), [0.0], on_value, None, 1 if side == "left" else 0)[0]
