"""LLaVA half of the 2Haff hot path on MI355X: CLIP ViT-L/14 tower -> linear projector -> embedding splice ->
Llama decoder with a KV cache -> greedy token. Host-side orchestration of the HIP kernels (no torch math).

Mirrors the reference's
  2Haff/model/llava/model/multimodal_encoder/clip_encoder.py:31-60 (hidden_states[select_layer][:,1:])
  2Haff/model/llava/model/llava_arch.py:35,93-96 (projector), :185-208,252-256 (splice)
  2Haff/model/llava/model/language_model/llava_llama.py:55-135 (forward; eval returns post-norm hidden)
whose arithmetic lives in transformers' CLIPVisionModel / LlamaModel. Unlike the reference (config.use_cache =
False, LISA.py:115 — every generated token re-runs CLIP and the whole prefix) the decoder here keeps K/V
resident in HBM; a causal model makes the two schedules numerically equivalent.
"""
import torch

from . import ops

CLIP = "model.vision_tower.vision_tower.vision_model"


def _f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def _pad_cols(w, mult):
    k = w.shape[1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w
    return torch.nn.functional.pad(w, (0, kp - k))


class ClipTowerHip:
    def __init__(self, sd, cfg, dtype, device):
        c = self.cfg = cfg
        self.dtype, self.device = dtype, device
        dev = device
        self.hd = c.hidden // c.heads
        wp = sd[CLIP + ".embeddings.patch_embedding.weight"].reshape(c.hidden, -1)
        self.w_patch = _pad_cols(wp, 8).to(dev, dtype).contiguous()  # K = 3*14*14 = 588 -> 592
        self.cls = sd[CLIP + ".embeddings.class_embedding"].to(dev, dtype)
        self.pos = sd[CLIP + ".embeddings.position_embedding.weight"].to(dev, dtype).contiguous()
        self.pre = (_f32(sd[CLIP + ".pre_layrnorm.weight"], dev), _f32(sd[CLIP + ".pre_layrnorm.bias"], dev))
        self.n_run = c.layers + 1 + c.select_layer if c.select_layer < 0 else c.select_layer
        self.layers = []
        for i in range(self.n_run):
            L = f"{CLIP}.encoder.layers.{i}"
            wq = torch.cat([sd[f"{L}.self_attn.{n}_proj.weight"] for n in ("q", "k", "v")], 0)
            bq = torch.cat([sd[f"{L}.self_attn.{n}_proj.bias"] for n in ("q", "k", "v")], 0)
            self.layers.append({
                "n1": (_f32(sd[L + ".layer_norm1.weight"], dev), _f32(sd[L + ".layer_norm1.bias"], dev)),
                "wqkv": wq.to(dev, dtype).contiguous(), "bqkv": _f32(bq, dev),
                "wo": sd[L + ".self_attn.out_proj.weight"].to(dev, dtype).contiguous(),
                "bo": _f32(sd[L + ".self_attn.out_proj.bias"], dev),
                "n2": (_f32(sd[L + ".layer_norm2.weight"], dev), _f32(sd[L + ".layer_norm2.bias"], dev)),
                "w1": sd[L + ".mlp.fc1.weight"].to(dev, dtype).contiguous(), "b1": _f32(sd[L + ".mlp.fc1.bias"], dev),
                "w2": sd[L + ".mlp.fc2.weight"].to(dev, dtype).contiguous(), "b2": _f32(sd[L + ".mlp.fc2.bias"], dev)})
        self._maps = {}

    def _maps_for(self, B):
        if B not in self._maps:
            n = self.cfg.n_patches
            into = (torch.arange(B)[:, None] * (n + 1) + 1 + torch.arange(n)[None, :]).reshape(-1)
            outof = torch.full((B, n + 1), -1, dtype=torch.int64)
            outof[:, 1:] = torch.arange(B)[:, None] * n + torch.arange(n)[None, :]
            self._maps[B] = (into.to(torch.int32).to(self.device), outof.reshape(-1).to(torch.int32).to(self.device))
        return self._maps[B]

    def hidden(self, images):
        """[B,3,224,224] -> residual stream [B*(n+1), C] after n_run layers (cls row included)."""
        c = self.cfg
        B = images.shape[0]
        g = c.image // c.patch
        n, C, H, hd = c.n_patches, c.hidden, c.heads, self.hd
        rows = ops.patchify_nchw(images.to(self.dtype), c.patch, g, g, self.w_patch.shape[1], self.dtype)
        into, _ = self._maps_for(B)
        h = torch.empty((B * (n + 1), C), dtype=self.dtype, device=self.device)
        h.view(B, n + 1, C)[:, 0] = self.cls
        ops.linear(rows, self.w_patch, row_map=into, out=h)
        ops.add_bcast(h, self.pos, mod=n + 1, out=h)
        h = ops.layernorm(h, self.pre[0], self.pre[1], c.eps)
        for L in self.layers:
            y = ops.layernorm(h, L["n1"][0], L["n1"][1], c.eps)
            qkv = ops.linear(y, L["wqkv"], bias=L["bqkv"]).view(B, n + 1, 3, H, hd)
            a = ops.attention(qkv[:, :, 0].permute(0, 2, 1, 3), qkv[:, :, 1].permute(0, 2, 1, 3),
                              qkv[:, :, 2].permute(0, 2, 1, 3), hd ** -0.5)
            ops.linear(a.view(B * (n + 1), C), L["wo"], bias=L["bo"], resid=h, out=h)
            y = ops.layernorm(h, L["n2"][0], L["n2"][1], c.eps)
            y = ops.linear(y, L["w1"], bias=L["b1"], act=ops.ACT_QUICK_GELU)
            ops.linear(y, L["w2"], bias=L["b2"], resid=h, out=h)
        return h

    def project(self, h, B, w_proj, b_proj):
        """mm_projector on the patch rows only (drop cls): -> [B, n, H_llm] contiguous."""
        _, outof = self._maps_for(B)
        n = self.cfg.n_patches
        out = ops.linear(h, w_proj, bias=b_proj, row_map=outof, out_rows=B * n)
        return out.view(B, n, -1)


class LlamaHip:
    def __init__(self, sd, cfg, dtype, device):
        l = self.cfg = cfg
        self.dtype, self.device = dtype, device
        dev = device
        self.hd = l.hidden // l.heads
        self.embed = sd["model.embed_tokens.weight"].to(dev, dtype).contiguous()
        self.layers = []
        F = l.ffn
        assert F % 16 == 0, "SwiGLU interleave needs ffn % 16 == 0"
        for i in range(l.layers):
            L = f"model.layers.{i}"
            wqkv = torch.cat([sd[f"{L}.self_attn.{n}_proj.weight"] for n in ("q", "k", "v")], 0)
            wg, wu = sd[L + ".mlp.gate_proj.weight"], sd[L + ".mlp.up_proj.weight"]
            # rows interleaved in 16-row groups [gate x16 | up x16] so the GEMM epilogue sees (gate, up) pairs
            wgu = torch.stack([wg.view(F // 16, 16, -1), wu.view(F // 16, 16, -1)], dim=1).reshape(2 * F, -1)
            self.layers.append({
                "n1": _f32(sd[L + ".input_layernorm.weight"], dev),
                "wqkv": wqkv.to(dev, dtype).contiguous(),
                "wo": sd[L + ".self_attn.o_proj.weight"].to(dev, dtype).contiguous(),
                "n2": _f32(sd[L + ".post_attention_layernorm.weight"], dev),
                "wgu": wgu.to(dev, dtype).contiguous(),
                "wd": sd[L + ".mlp.down_proj.weight"].to(dev, dtype).contiguous()})
            del wqkv, wgu
        self.norm = _f32(sd["model.norm.weight"], dev)
        self.lm_head = sd["lm_head.weight"].to(dev, dtype).contiguous()
        # Decode steps of <= 8 rows (round 5; <= 4 before) carry RMSNorm between the products (ops.linear_rms): no norm kernels, the q/k/v and gate/up
        # weights get a second copy with the norm weight folded in (built on first use; +9 GB at 7B, +18 GB at 13B of 288)
        self.carry_rms = dtype == torch.bfloat16 and self.hd == 128 and l.hidden % 128 == 0 and l.ffn % 128 == 0
        self.carry_rms_max_rows = 8   # the consumer side of haff_gemm_bf16_rms gathers the partials of <= 8 rows
        self._folded = None
        # Round 6: the whole <= 8-row decode step as ONE launch (ops.decode_chain, csrc/decode_chain.hip): the five stages of every
        # layer are workgroup ranges chained by arrival counters, weights are requested before a workgroup waits for its inputs.
        # Same arithmetic as the five-launch layer below (bit-identical at 5..8 rows). False: the five launches per layer.
        self.decode_chain = True      # "stages": the same kernel as one launch per (layer, stage) (tests, A/B)
        self._cs = None
        # Prefill-sized batches (>= 1024 rows: where the 8-wave tile runs anyway): RoPE and the KV-cache append ride in the q|k|v projection's epilogue
        # (ops.qkv_rope): the weights get a second, row-permuted copy on first use (+3.2 GB at 7B, +6.3 GB at 13B of 288)
        self.fused_qkv_rope = True    # ("force": any prefill; False: haff_gemm_bf16 + haff_rope_cache)
        self._wqkv_rope = None
        # fp32 RESIDUAL STREAM (bf16 mode; round 5): the hidden-state stream lives in HBM as fp32 — o_proj /
        # down_proj add their fp32 accumulators to it and write fp32, RMSNorm reads it and rounds the NORMALISED row to bf16 once
        # for the bf16 MFMA products. Off by default (it gives up the norm-carrying 5-launch decode layer at <= 8 rows).
        self.fp32_stream = False

    def _cos_sin(self, tmax):
        if self._cs is None or self._cs.shape[0] < tmax:
            d = self.hd
            inv = 1.0 / (self.cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
            ang = torch.arange(tmax, dtype=torch.float32)[:, None] * inv[None, :]
            self._cs = torch.cat([ang.cos(), ang.sin()], dim=1).contiguous().to(self.device)
        return self._cs

    def new_cache(self, B, tmax):
        H = self.cfg.hidden
        return {"k": [torch.empty((B, tmax, H), dtype=self.dtype, device=self.device) for _ in self.layers],
                "v": [torch.empty((B, tmax, H), dtype=self.dtype, device=self.device) for _ in self.layers],
                "len": 0, "tmax": tmax,
                # per-row decode state (ragged batches): next position and visible key count of every row
                "pos": torch.zeros((B,), dtype=torch.int32, device=self.device),
                "nk": torch.ones((B,), dtype=torch.int32, device=self.device)}

    def forward(self, x, cache, keep_rows=None):
        """x [B,T,H] embeddings of the next T positions; appends to the cache; returns post-norm hidden [B,T,H].
        keep_rows (round 6; int64 [n] flat row indices b * T + t, or None = all): the rows whose FINAL hidden state the caller will
        read. Every position still runs through every layer's attention inputs (its K / V are what later tokens attend to), but in
        the LAST layer only these rows take o_proj, the MLP and the final norm — nothing downstream of the last layer's K / V
        depends on the other rows (llava_llama.py:93-105 returns them, LISA.py:443-485 never looks at them); their rows of the
        result are zero. 75 % of the last layer's linear FLOPs: 0.9 % of the step at 64 frames."""
        l = self.cfg
        B, T, H = x.shape
        nh, hd = l.heads, self.hd
        pos0 = cache["len"]
        assert pos0 + T <= cache["tmax"]
        cs = self._cos_sin(cache["tmax"])
        x = x.reshape(B * T, H).clone() if not x.is_contiguous() else x.reshape(B * T, H)
        s32 = bool(self.fp32_stream) and self.dtype == torch.bfloat16
        nd = self.dtype if s32 else None     # norm output dtype: bf16 rows from the fp32 stream
        if s32:
            x = x.float()
        fused = self.fused_qkv_rope and T > 1 and \
            ops.qkv_rope_supported(B * T, nh, hd, H, self.dtype, 1 if self.fused_qkv_rope == "force" else 1024)
        if fused and self._wqkv_rope is None:
            self._wqkv_rope = [ops.rope_permute_rows(L["wqkv"]) for L in self.layers]
        for li, L in enumerate(self.layers):
            h = ops.rmsnorm(x, L["n1"], l.rms_eps, out_dtype=nd)
            kc, vc = cache["k"][li], cache["v"][li]
            tk = pos0 + T
            if fused:
                q = ops.qkv_rope(h, self._wqkv_rope[li], kc, vc, cs, B, T, nh, hd, pos0).view(B, T, nh, hd).permute(0, 2, 1, 3)
            else:
                qkv = ops.linear(h, L["wqkv"])
                ops.rope_cache(qkv, kc, vc, cs, B, T, nh, nh, hd, pos0)
                q = qkv.view(B, T, 3, nh, hd)[:, :, 0].permute(0, 2, 1, 3)
            k = kc.view(B, cache["tmax"], nh, hd).permute(0, 2, 1, 3)[:, :, :tk]
            v = vc.view(B, cache["tmax"], nh, hd).permute(0, 2, 1, 3)[:, :, :tk]
            a = ops.attention(q, k, v, hd ** -0.5, causal=T > 1, q_pos0=tk - T)
            a2 = a.view(B * T, H)
            if keep_rows is not None and li == len(self.layers) - 1:
                a2, x = a2.index_select(0, keep_rows), x.index_select(0, keep_rows)   # (row gathers: the products below see n rows)
            x = ops.linear(a2, L["wo"], resid=x, out=x)
            h = ops.rmsnorm(x, L["n2"], l.rms_eps, out_dtype=nd)
            g = ops.linear(h, L["wgu"], swiglu=True)
            x = ops.linear(g, L["wd"], resid=x, out=x)
        cache["len"] = pos0 + T
        y = ops.rmsnorm(x, self.norm, l.rms_eps, out_dtype=nd)
        if keep_rows is not None:
            full = torch.zeros((B * T, H), dtype=y.dtype, device=y.device)
            full.index_copy_(0, keep_rows, y)
            y = full
        return y.view(B, T, H)

    def decode_rows(self, x1, cache):
        """One KV-cached position per row at PER-ROW positions: x1 [B,1,H] is the embedding of row b's next token, which
        sits at position cache["pos"][b] (device int32 [B]) and attends that row's first pos+1 cached keys — greedy decode
        of right-padded prompts of different lengths (padding rule of utils/dataset.py:90-93). The caller advances
        cache["pos"] / cache["nk"] (nk = pos + 1). Returns post-norm hidden [B,1,H]."""
        l = self.cfg
        B, T, H = x1.shape
        assert T == 1
        nh, hd = l.heads, self.hd
        cs = self._cos_sin(cache["tmax"])
        pos, nk = cache["pos"], cache["nk"]
        x = x1.reshape(B, H).clone() if not x1.is_contiguous() else x1.reshape(B, H)
        s32 = bool(self.fp32_stream) and self.dtype == torch.bfloat16
        nd = self.dtype if s32 else None
        if s32:
            x = x.float()
        elif self.carry_rms and B <= self.carry_rms_max_rows:
            return self._decode_rows_carry(x, cache, cs, nk)
        for li, L in enumerate(self.layers):
            h = ops.rmsnorm(x, L["n1"], l.rms_eps, out_dtype=nd)
            qkv = ops.linear(h, L["wqkv"])
            kc, vc = cache["k"][li], cache["v"][li]
            if self.dtype == torch.bfloat16 and hd == 128:
                # RoPE of q and the new k, the cache append and the attention over the row's pos+1 keys in ONE launch
                a = ops.decode_attention_rope(qkv, kc, vc, cs, nh, hd, hd ** -0.5, nk)
            else:
                ops.rope_cache_rows(qkv, kc, vc, cs, B, 1, nh, nh, hd, pos)
                q = qkv.view(B, 1, 3, nh, hd)[:, :, 0].permute(0, 2, 1, 3)
                k = kc.view(B, cache["tmax"], nh, hd).permute(0, 2, 1, 3)
                v = vc.view(B, cache["tmax"], nh, hd).permute(0, 2, 1, 3)
                a = ops.attention_decode_rows(q, k, v, hd ** -0.5, nk)
            x = ops.linear(a.view(B, H), L["wo"], resid=x, out=x)
            h = ops.rmsnorm(x, L["n2"], l.rms_eps, out_dtype=nd)
            g = ops.linear(h, L["wgu"], swiglu=True)
            x = ops.linear(g, L["wd"], resid=x, out=x)
        return ops.rmsnorm(x, self.norm, l.rms_eps, out_dtype=nd).view(B, 1, H)

    def fold_norm_weights(self):
        """The q|k|v and gate|up weights with their RMSNorm's gamma folded into the columns (what the norm-carrying decode products
        multiply the RAW residual stream with); built once, on first use."""
        if self._folded is None:
            self._folded = [((L["wqkv"].float() * L["n1"][None, :]).to(torch.bfloat16).contiguous(),
                             (L["wgu"].float() * L["n2"][None, :]).to(torch.bfloat16).contiguous()) for L in self.layers]
        return self._folded

    def _decode_rows_carry(self, x, cache, cs, nk):
        """decode_rows for <= 8 rows without norm kernels: o_proj / down_proj write, beside the residual stream, each
        workgroup's sum of squares of its slice of it; the next q/k/v or gate/up product (on norm-weight-folded weights and the
        RAW stream) turns the partials into 1/rms in its epilogue. 5 launches per layer instead of 7 (layer 0 takes its
        statistic from one pass over the embedding rows)."""
        l = self.cfg
        B, H = x.shape
        nh, hd = l.heads, self.hd
        self.fold_norm_weights()
        stats = ops.row_stats(x, l.rms_eps, rms=True)
        if self.decode_chain and x.is_contiguous() and ops.decode_chain_supported(B, H, l.ffn, nh, len(self.layers)):
            ch = cache.get("chain")
            if ch is None:
                dev, bf = self.device, torch.bfloat16
                z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)   # noqa: E731
                ch = cache["chain"] = {
                    "table": ops.decode_chain_table([(self._folded[i][0], L["wo"], self._folded[i][1], L["wd"], cache["k"][i], cache["v"][i])
                                                     for i, L in enumerate(self.layers)]),
                    "qkv": z((B, 3 * H), bf), "att": z((B, H), bf), "g": z((B, l.ffn), bf), "ws": z((H // 16, 2, 16, 16), torch.float32),
                    "ssq_a": z((H // 16, 16), torch.float32), "ssq_b": z((H // 16, 16), torch.float32),
                    # arrival counters + tickets + the sticky error word: zeroed here once, the launch re-zeroes what it counts with
                    "sync": z((ops.decode_chain_sync_words(len(self.layers), H),), torch.int32)}
            pa, pb = ch["ssq_a"], ch["ssq_b"]
            ops.decode_chain(ch["table"], len(self.layers), x, ch["qkv"], ch["att"], ch["g"], pa, pb, ch["ws"], stats, l.rms_eps, cs, nk, nh,
                             cache["tmax"], hd ** -0.5, ch["sync"], per_stage_launches=self.decode_chain == "stages")
            return ops.rmsnorm(x, self.norm, l.rms_eps).view(B, 1, H)
        if "ssq" not in cache:
            cache["ssq"] = [torch.zeros((H // 16, 16), dtype=torch.float32, device=self.device) for _ in range(2)]
        pa, pb = cache["ssq"]
        for li, L in enumerate(self.layers):
            wq, wgu = self._folded[li]
            if li == 0:
                qkv = ops.linear(x, wq, ln_stats=stats)
            else:
                qkv = ops.linear_rms(x, wq, ssq_in=pb, eps=l.rms_eps)
            a = ops.decode_attention_rope(qkv, cache["k"][li], cache["v"][li], cs, nh, hd, hd ** -0.5, nk)
            ops.linear_rms(a.view(B, H), L["wo"], resid=x, out=x, ssq_out=pa)
            g = ops.linear_rms(x, wgu, swiglu=True, ssq_in=pa, eps=l.rms_eps)
            ops.linear_rms(g, L["wd"], resid=x, out=x, ssq_out=pb)
        return ops.rmsnorm(x, self.norm, l.rms_eps).view(B, 1, H)

    def next_token_logits(self, hidden_last):
        """hidden_last [B,H] -> fp32 logits [B,V] (lm_head, no bias; llava_llama.py:105)."""
        return ops.linear(hidden_last, self.lm_head, out_dtype=torch.float32)
