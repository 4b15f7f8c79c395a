"""How the two HIP streams of one `LisaMI355.evaluate()` share the chip.

The SAM encoder (MFMA-bound, persistent GEMM tiles that fill every CU) runs on a side stream beside CLIP -> prefill -> greedy
decode on the caller's stream (LISA.py:432-534: the two halves only meet at the mask decoders). The decode steps are weight
streams: launched beside a GEMM that owns all 256 CUs they wait for it, launched after it they leave the matrix cores idle.
So for the part of the encoder that runs WHILE the decode steps run, its GEMM launches take `cap` < 256 workgroups
(`haff_gemm_stream_cap`) and the decode kernels run on the CUs left over; the cap balances the two so that they finish
together. Results never depend on it (a launch computes the same tiles with fewer workgroups).

The plan is static (the host enqueues the encoder before the decode steps exist) and comes from a small work model whose three
rates were measured on MI355X (`tools/stream_phases.py`, `profiles/r5_overlap_*`):
  * Rates.enc / Rates.llm: flop/s the encoder's and the CLIP + prefill launches reach on a full chip;
  * Rates.stream_bw_per_cu: bytes/s one CU streams in the decode step's kernels (they scale with the CUs they get: 3.6 ms per step on
    256 CUs, 8.9 ms on 96), for the <= 32-row weight-streaming kernel and for the 33..64-row split-K path;
  * enc_share(): while both streams are MFMA-bound they serialise launch by launch, and the encoder advances by this much of
    what CLIP + prefill take — more the longer its launches are next to the prefill's (0.47 / 0.8 / 1.37 at chunks of 1/8, 1/4,
    1/2 of the step's frames).
Balanced when  W_enc / cap == W_dec / (256 - cap)  (CU-seconds of the encoder inside the window / of the decode steps).

Round 6: the rates are a `Rates` value. `NOMINAL` holds the numbers fitted in round 5 (one box of a pool whose boxes differ by
~5 % on every MFMA-bound shape); `calibrate(device)` scales them by two timed probes on THIS device at model construction (one
MFMA-bound product of the encoder's lin1 shape, one weight-streaming product cycling through more weights than the 256 MB
Infinity Cache holds): the plan follows the box it runs on. What the plan assumes about the call is `expected_new_tokens`
(LisaMI355's default 8: a `[SEG]` answer): more decode steps than planned run alone after the encoder (nothing lost against
no plan), fewer leave the last passes capped for nothing (bounded by 1 - cap/256 <= 12.5 % of those passes at 64 frames).
`LisaMI355.sam_chunk_caps = None` is the documented safe default for callers that are not throughput batches of short answers.
"""
from dataclasses import dataclass, replace

from . import flops as hflops

CAPS = (128, 160, 192, 224)     # multiples of 32: 16 / 20 / 24 / 28 workgroups per XCD keep the tile raster's M-groups whole
MIN_FRAMES = 4                  # below: the encoder's launches are shorter than 256 tiles anyway (measured: no gain at 1)


@dataclass(frozen=True)
class Rates:
    enc: float = 1.13e15                          # flop/s of the encoder's launches on a full chip
    llm: float = 1.26e15                          # ... of CLIP + prefill
    stream_bw_per_cu: tuple = (16.4e9, 27.0e9)    # bytes/s one CU streams: <= 32-row kernel, 33..64-row split-K path
    source: str = "nominal (round 5 fit, profiles/r5_overlap_*)"


NOMINAL = Rates()
# what the two probes of calibrate() read on the kind of box NOMINAL was fitted on (profiles/r6_overlap_probe.txt)
PROBE_NOMINAL = {"mfma_flops": 1.10e15, "stream_bytes": 4.40e12}
_calibrated = {}


def _time_ms(fn, warm, reps):
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def probe(device, repeats=4):
    """-> {"mfma_flops", "stream_bytes"}: the rate of one encoder-shaped product (16384 x 5120 x 1280, bias + GELU: lin1 of a
    ViT-H block, image_encoder.py:191 / common.py:13-26) and of an 8-row product against 4 x 180 MB of weights read in turn (one
    Llama gate|up of a decode step, llava_llama.py:93-102; four copies so that no read is served by the Infinity Cache)."""
    import torch
    from . import ops
    g = torch.Generator(device=device).manual_seed(0)
    x = torch.randn((16384, 1280), device=device, generator=g).to(torch.bfloat16)
    w = (torch.randn((5120, 1280), device=device, generator=g) * 0.03).to(torch.bfloat16)
    b = torch.zeros((5120,), device=device, dtype=torch.float32)
    xs = torch.randn((8, 4096), device=device, generator=g).to(torch.bfloat16)
    ws = [(torch.randn((22016, 4096), device=device, generator=g) * 0.02).to(torch.bfloat16) for _ in range(4)]
    mfma = stream = 0.0
    for _ in range(repeats):     # the best of a few readings: the first ones of a cold device read 7 % / 16 % low (clock ramp)
        ms = _time_ms(lambda: ops.linear(x, w, bias=b, act=ops.ACT_GELU), 3, 8)
        mfma = max(mfma, 2.0 * 16384 * 5120 * 1280 / (ms * 1e-3))
        ms = _time_ms(lambda: [ops.linear(xs, wi) for wi in ws], 1, 4)
        stream = max(stream, 4 * 2.0 * 22016 * 4096 / (ms * 1e-3))
    return {"mfma_flops": mfma, "stream_bytes": stream}


def calibrate(device):
    """NOMINAL scaled by this device's two probe readings (clamped to +-25 %: a probe disturbed by another process must not
    produce a wild plan); cached per device for the life of the process. ~60 ms once."""
    import torch
    key = str(torch.device(device))
    if key not in _calibrated:
        p = probe(device)
        clamp = lambda v: min(1.25, max(0.75, v))   # noqa: E731
        fm, fs = clamp(p["mfma_flops"] / PROBE_NOMINAL["mfma_flops"]), clamp(p["stream_bytes"] / PROBE_NOMINAL["stream_bytes"])
        _calibrated[key] = replace(NOMINAL, enc=NOMINAL.enc * fm, llm=NOMINAL.llm * fm,
                                   stream_bw_per_cu=tuple(v * fs for v in NOMINAL.stream_bw_per_cu),
                                   source="calibrated: mfma probe %.0f TFLOP/s (x%.3f), stream probe %.2f TB/s (x%.3f)" %
                                          (p["mfma_flops"] / 1e12, fm, p["stream_bytes"] / 1e12, fs))
    return _calibrated[key]


def enc_share(chunk, frames):
    return 2.33 * (min(chunk, frames) / float(frames)) ** 0.77


def decode_step_bytes(cfg, frames, positions):
    """HBM bytes of one KV-cached decode step: every Llama weight once + lm_head + the K / V rows read."""
    l = cfg.llm
    w = l.layers * (4 * l.hidden * l.hidden + 3 * l.hidden * l.ffn) + l.hidden * l.vocab
    kv = l.layers * frames * positions * 2 * l.hidden
    return 2.0 * (w + kv)


def plan(cfg, frames, chunk, prompt_tokens, new_tokens, late, rates=NOMINAL):
    """-> (caps, wait): caps[i] = workgroups per persistent GEMM launch of encoder chunk i (None: no cap anywhere); wait = the
    encoder's stream waits for the prefill on the GPU (late mode: the whole encoder runs beside the decode steps).
    frames per step, encoder chunk size, prompt ids per row (the <image> sentinel included), tokens to generate, late = the
    encoder is enqueued behind the prefill (few frames); rates: NOMINAL or calibrate(device)."""
    steps = new_tokens - 1
    if frames < MIN_FRAMES or steps <= 0 or chunk <= 0:
        return None, False
    n_chunks = (frames + chunk - 1) // chunk
    T = prompt_tokens + cfg.clip.n_patches - 1
    enc_s = hflops.sam_encoder_flops(cfg.sam) / rates.enc                     # per frame, full chip
    parts = hflops.frame_flops(cfg, max(prompt_tokens - 4, 0), 1)
    llm_s = frames * (parts["clip"] + parts["projector_fcs"] + parts["llm"]) / rates.llm
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
    w_dec = steps * decode_step_bytes(cfg, frames, T + steps) / rates.stream_bw_per_cu[0 if frames <= 32 else 1]
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
