"""Algorithmic work per frame of the 2Haff hot path (2 x MAC; KV-cached minimum) — SURVEY.md §8(d).

Used by bench.py for `roofline.achieved`; checked in tests against SURVEY's 10.01 TFLOP (7B) / 13.73 (13B)."""


def sam_encoder_flops(s):
    C, g, ws = s.embed_dim, s.grid, s.window
    N = g * g
    gp = (g + ws - 1) // ws * ws
    t_win = gp * gp
    f = 2.0 * N * (3 * s.patch * s.patch) * C
    for i in range(s.depth):
        glob = i in s.global_idx
        T = N if glob else t_win
        S = g if glob else ws
        f += 2.0 * T * 4 * C * C                      # qkv + proj
        f += 2.0 * N * 2 * s.mlp_ratio * C * C        # MLP
        ntok = N if glob else ws * ws
        f += 2.0 * 2 * T * ntok * C                   # QK^T + PV over all heads
        f += 2.0 * T * 2 * S * C                      # decomposed rel-pos terms
    f += 2.0 * N * C * s.out_chans + 2.0 * N * 9 * s.out_chans * s.out_chans
    return f


def clip_flops(c):
    n = c.n_patches + 1
    layers = c.layers + 1 + c.select_layer if c.select_layer < 0 else c.select_layer
    f = 2.0 * c.n_patches * 3 * c.patch * c.patch * c.hidden
    f += layers * (2.0 * n * (4 * c.hidden * c.hidden + 2 * c.hidden * c.mlp) + 2.0 * 2 * n * n * c.hidden)
    return f


def llm_flops(l, T, n_gen):
    per_tok = l.layers * (4 * l.hidden * l.hidden + 3 * l.hidden * l.ffn)
    f = 2.0 * per_tok * (T + n_gen - 1)
    f += 2.0 * l.layers * l.hidden * sum(t + 1 for t in range(T + n_gen - 1))  # causal QK^T+PV = 2*H*(t+1) MAC
    f += 2.0 * l.hidden * l.vocab * n_gen
    return f


def decoder_flops(s, n_prompts=1):
    C, N = s.out_chans, s.grid * s.grid
    per = 0.0
    per += 2.0 * N * C * (C // 2) * 2 * 3            # k,v projections of image tokens, 2 layers + final
    per += 2.0 * N * C * (C // 2) * 2 + 2.0 * N * (C // 2) * C * 2   # i2t q-proj + out-proj, 2 layers
    per += 2.0 * 2 * 6 * N * (C // 2) * 5            # attention scores+values (3 t2i + 2 i2t)
    per += 2.0 * N * C * C                            # first transposed conv
    per += 2.0 * N * 4 * 64 * 128                     # second transposed conv
    per += 2.0 * N * 16 * 32                          # hypernetwork dot
    return 2 * n_prompts * per                        # left + right


def frame_flops(cfg, text_tokens=32, n_gen=8):
    L = 4 + text_tokens
    T = L + cfg.clip.n_patches - 1
    parts = {
        "sam_encoder": sam_encoder_flops(cfg.sam),
        "clip": clip_flops(cfg.clip),
        "projector_fcs": 2.0 * cfg.clip.n_patches * cfg.clip.hidden * cfg.llm.hidden
                         + 2.0 * (cfg.llm.hidden * cfg.llm.hidden + cfg.llm.hidden * cfg.out_dim),
        "llm": llm_flops(cfg.llm, T, n_gen),
        "decoders": decoder_flops(cfg.sam),
    }
    parts["total"] = sum(parts.values())
    return parts
