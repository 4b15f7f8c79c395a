"""How the two HIP streams of one `LisaMI355.evaluate()` share the chip.

The SAM encoder (MFMA-bound, persistent GEMM tiles that fill every CU) runs on a side stream beside CLIP -> prefill -> greedy
decode on the caller's stream (LISA.py:432-534: the two halves only meet at the mask decoders). The decode steps are weight
streams: launched beside a GEMM that owns all 256 CUs they wait for it, launched after it they leave the matrix cores idle.
So for the part of the encoder that runs WHILE the decode steps run, its GEMM launches take `cap` < 256 workgroups
(`haff_gemm_stream_cap`) and the decode kernels run on the CUs left over; the cap balances the two so that they finish
together. Results never depend on it (a launch computes the same tiles with fewer workgroups).

The plan is static (the host enqueues the encoder before the decode steps exist) and comes from a small work model whose three
rates were measured on MI355X (`tools/stream_phases.py`, `profiles/r5_overlap_*`):
  * ENC_RATE / LLM_RATE: flop/s the encoder's and the CLIP + prefill launches reach on a full chip;
  * STREAM_BW_PER_CU: bytes/s one CU streams in the decode step's kernels (they scale with the CUs they get: 3.6 ms per step on
    256 CUs, 8.9 ms on 96), for the <= 32-row weight-streaming kernel and for the 33..64-row split-K path;
  * enc_share(): while both streams are MFMA-bound they serialise launch by launch, and the encoder advances by this much of
    what CLIP + prefill take — more the longer its launches are next to the prefill's (0.47 / 0.8 / 1.37 at chunks of 1/8, 1/4,
    1/2 of the step's frames).
Balanced when  W_enc / cap == W_dec / (256 - cap)  (CU-seconds of the encoder inside the window / of the decode steps).
"""
from . import flops as hflops

CAPS = (128, 160, 192, 224)     # multiples of 32: 16 / 20 / 24 / 28 workgroups per XCD keep the tile raster's M-groups whole
ENC_RATE = 1.13e15
LLM_RATE = 1.26e15
STREAM_BW_PER_CU = (16.4e9, 27.0e9)
MIN_FRAMES = 4                  # below: the encoder's launches are shorter than 256 tiles anyway (measured: no gain at 1)


def enc_share(chunk, frames):
    return 2.33 * (min(chunk, frames) / float(frames)) ** 0.77


def decode_step_bytes(cfg, frames, positions):
    """HBM bytes of one KV-cached decode step: every Llama weight once + lm_head + the K / V rows read."""
    l = cfg.llm
    w = l.layers * (4 * l.hidden * l.hidden + 3 * l.hidden * l.ffn) + l.hidden * l.vocab
    kv = l.layers * frames * positions * 2 * l.hidden
    return 2.0 * (w + kv)


def plan(cfg, frames, chunk, prompt_tokens, new_tokens, late):
    """-> (caps, wait): caps[i] = workgroups per persistent GEMM launch of encoder chunk i (None: no cap anywhere); wait = the
    encoder's stream waits for the prefill on the GPU (late mode: the whole encoder runs beside the decode steps).
    frames per step, encoder chunk size, prompt ids per row (the <image> sentinel included), tokens to generate, late = the
    encoder is enqueued behind the prefill (few frames)."""
    steps = new_tokens - 1
    if frames < MIN_FRAMES or steps <= 0 or chunk <= 0:
        return None, False
    n_chunks = (frames + chunk - 1) // chunk
    T = prompt_tokens + cfg.clip.n_patches - 1
    enc_s = hflops.sam_encoder_flops(cfg.sam) / ENC_RATE                     # per frame, full chip
    parts = hflops.frame_flops(cfg, max(prompt_tokens - 4, 0), 1)
    llm_s = frames * (parts["clip"] + parts["projector_fcs"] + parts["llm"]) / LLM_RATE
    starts = [min(i * chunk, frames) * enc_s for i in range(n_chunks)]
    total = frames * enc_s
    if late:
        first = 0
    else:
        done = enc_share(chunk, frames) * llm_s     # encoder seconds behind it when the first decode step is enqueued
        first = next((i for i, s in enumerate(starts) if s >= done - 0.5 * chunk * enc_s), None)
        if first is None:
            return None, False
    w_enc = (total - starts[first]) * 256.0
    w_dec = steps * decode_step_bytes(cfg, frames, T + steps) / STREAM_BW_PER_CU[0 if frames <= 32 else 1]
    c = 256.0 * w_enc / (w_enc + w_dec)
    if c > 240.0:
        return None, False
    cap = min(CAPS, key=lambda v: abs(v - c))
    return [256] * first + [cap] * (n_chunks - first), bool(late)


def auto_chunk(frames, late):
    """Encoder chunk size for `sam_chunk="auto"`: one chunk of up to 16 frames in late mode; otherwise a quarter of the step's
    frames (8..16), so that the second half of the encoder lines up with the decode steps (64 frames: 16, 32: 8)."""
    if late:
        return max(1, min(frames, 16))
    return max(8, min(16, (frames // 4 + 7) // 8 * 8))
