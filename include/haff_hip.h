/*
 * haff_hip.h — C-ABI of libhaff_hip.so, the MI355X (gfx950 / CDNA4) hot path of pearl-robot-lab/2HandedAfforder.
 *
 * The reference has no FFI / plugin boundary of its own: its per-frame path is Python
 * (`LISAForCausalLM.evaluate`, 2Haff/model/LISA.py:432-534) over torch + transformers. This library sits UNDER
 * that Python boundary: each entry point replaces the torch / transformers op(s) cited next to it, and is what a
 * maintainer binds with ctypes from the reference's own modules.
 *
 * Conventions
 *   - all pointers are caller-owned DEVICE pointers (HBM) unless marked "host"; no internal allocation; re-entrant;
 *     `stream` is a hipStream_t passed as void* (0 = null stream); kernels are only enqueued. No process-wide setting:
 *     the one piece of state the library keeps is haff_gemm_stream_cap's table, keyed by stream and guarded by a mutex
 *     (scheduling only, results never depend on it) — host threads that enqueue on different streams do not interact.
 *   - return value: 0 ok, -1 bad argument, -2 unsupported shape, -3 launch error. No exceptions cross the ABI.
 *   - dtype codes: 0 = bf16 (raw 16-bit payload), 1 = f32. bf16 = throughput mode (bf16 MFMA, fp32 accumulate,
 *     fp32 softmax/norm statistics); f32 = parity mode (every op in fp32, for the 1e-3 mask-logit criterion).
 *   - strides/leading dimensions are in ELEMENTS.
 *   - activation codes: 0 none, 1 GELU(erf), 2 quick-GELU (x*sigmoid(1.702x)), 3 ReLU, 4 SiLU.
 */
#ifndef HAFF_HIP_H
#define HAFF_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- GEMM with fused epilogue: C[M,N] = act(A[M,K] . W[N,K]^T + bias) + resid -------------------------------
 * Replaces every nn.Linear / 1x1 conv / patchify conv on the path: SAM qkv/proj/MLP (image_encoder.py:223-224,258;
 * common.py:13-26), CLIP & Llama linears (transformers, reached from clip_encoder.py:53-56 / llava_llama.py:93-105),
 * mm_projector (llava_arch.py:35), text_hidden_fcs (LISA.py:95-101), SAM decoder linears (transformer.py:206-209),
 * first ConvTranspose2d of the upscaler (mask_decoder.py:55-57) as a per-pixel GEMM.
 * row_map (int32[M], may be null): output AND residual row = row_map[m]; negative = row dropped (fuses
 * window_unpartition, image_encoder.py:291-318). swiglu != 0: W rows interleaved in 16-row groups
 * [gate x16 | up x16], output width N/2 = silu(gate)*up (LlamaMLP). out_f32: C (and resid) are f32.
 * Requirements: K % 8 == 0, lda % 8 == 0, ldw % 8 == 0, A and W 16-byte aligned. */
int haff_gemm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias,
                   const void* resid, long ldr, const int* row_map, int M, int N, int K, int act, int out_f32,
                   int swiglu, void* stream);
/* same, with an explicit tile choice for measurements: 0 auto, 1 = 128x128, 2 = 256x256, 3 = 192x256 workgroup tile */
int haff_gemm_bf16_cfg(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias,
                       const void* resid, long ldr, const int* row_map, int M, int N, int K, int act, int out_f32,
                       int swiglu, int tile_cfg, void* stream);
/* haff_gemm_bf16 with a caller-provided DEVICE workspace (16-B aligned; 16 * M * N bytes are used when it applies): for
 * 33..64-row products on weights of <= 8192 rows (the decode-step o_proj / down_proj of llava_llama.py:93-102 at batch 64)
 * the weight-streaming kernel splits K over workgroups, fp32 partial tiles go through the workspace and are summed in a
 * fixed order (deterministic). NULL workspace, or a shape with no useful split: identical to haff_gemm_bf16. */
int haff_gemm_bf16_ws(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias,
                      const void* resid, long ldr, const int* row_map, int M, int N, int K, int act, int out_f32,
                      int swiglu, void* workspace, long workspace_bytes, void* stream);
/* Decode-sized product (M <= 16; with ssq_in: M <= 8, ssq_n <= 512, ssq_in 16-B aligned; K % 128 == 0) that carries Llama's RMSNorm between products without a norm kernel
 * (transformers LlamaDecoderLayer as reached from 2Haff/model/llava/model/language_model/llava_llama.py:93-102:
 * input_layernorm -> q/k/v_proj, post_attention_layernorm -> gate/up_proj). ssq_in != NULL: row m of A . W^T is scaled by
 * rsqrt(sum_{b < ssq_n} ssq_in[b][m] / K + eps) before the epilogue — W must have the norm weight folded into its columns
 * and A is the un-normalised residual stream. ssq_out != NULL (bf16 output, no SwiGLU): workgroup b writes
 * ssq_out[b][m] = sum over its output columns of bf16(C[m][n])^2 (m < 16), *n_parts_out (host int, may be NULL) = number of
 * workgroups — exactly what the next product's ssq_in / ssq_n take. Both arrays: DEVICE fp32 [parts][16]. */
int haff_gemm_bf16_rms(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias,
                       const void* resid, long ldr, int M, int N, int K, int act, int out_f32, int swiglu,
                       const float* ssq_in, int ssq_n, float eps, float* ssq_out, int* n_parts_out, void* stream);
/* One KV-cached decode step of n_layers Llama layers at M <= 8 rows as ONE launch (csrc/decode_chain.hip): the five stages of
 * every transformers LlamaDecoderLayer (q|k|v product with input_layernorm carried in -> RoPE + cache append + attention ->
 * o_proj + residual -> gate|up SwiGLU with post_attention_layernorm carried in -> down_proj + residual; reached from
 * llava_llama.py:93-102 during LISA.py:443-450's greedy generate) are workgroup ranges of one grid chained by arrival counters:
 * a workgroup requests its weight slab before it waits for its inputs, so the weight stream does not drain at stage boundaries.
 * Replaces, per layer, haff_gemm_bf16_rms x4 + haff_decode_attention_rope_rows_bf16; same arithmetic statement for statement
 * (bit-identical where that path splits a head over 4 waves: 5..8 rows).
 *   layers: HOST array of n_layers haff_chain_layer {wqkv [3H][H] (input_layernorm gamma folded into the columns), wo [H][H],
 *           wgu [2F][H] (16-row [gate | up] groups, post_attention_layernorm gamma folded in), wd [H][F], kcache, vcache
 *           [M][tmax][H]} of DEVICE pointers, copied into the kernel arguments;
 *   x [M][H] bf16 residual stream, in place; stats0 f32 [M][2] {mean, rstd} of its rows on entry (haff_row_stats, rms);
 *   qkv [M][3H], att [M][H], g [M][F] bf16, ssq_a / ssq_b f32 [H/16][16] and ws f32 [H/16][2][16][16] (the partial tiles of
 *   o_proj / down_proj, whose K extent is split over two workgroups and added in a fixed order): scratch; cos_sin f32 [tmax][128];
 *   nk_rows device int32 [M] = position of the new token + 1;
 *   sync: DEVICE uint32 [haff_decode_chain_sync_words(n_layers, hidden)], zeroed ONCE by the caller — the launch re-zeroes its arrival
 *   counters (8 shards + 8 replicas per (layer, stage), a 128-byte line each), the last line holds a sticky flag set when a
 *   bounded wait ran out (the grid then drains with a garbage result): haff_decode_chain_status.
 * hidden % 256 == 0, ffn % 256 == 0, hidden == heads * 128, hidden <= 8192, M <= 8, n_layers <= 48, pointers 16-B aligned;
 * otherwise -2 / -1. per_stage_launches != 0: the same kernel as one launch per (layer, stage) — identical arithmetic without the
 * chaining (tests, A/B). Handed-off bytes travel as agent-scope (sc1) stores and loads, no fences. haff_decode_chain_supported: workgroups per layer of the launch, 0 = unsupported geometry.
 * haff_decode_chain_status (synchronises the stream): 0 = every wait so far was satisfied, 1 = one ran out. */
typedef struct haff_chain_layer { const void *wqkv, *wo, *wgu, *wd; void *kcache, *vcache; } haff_chain_layer;
int haff_decode_chain_bf16(const haff_chain_layer* layers, int n_layers, int M, int hidden, int ffn, int heads, void* x,
                           void* qkv, void* att, void* g, float* ssq_a, float* ssq_b, float* ws, const float* stats0, float eps,
                           const float* cos_sin, const int* nk_rows, int tmax, float scale, unsigned* sync,
                           int per_stage_launches, void* stream);
int haff_decode_chain_supported(int M, int hidden, int ffn, int heads, int n_layers);
int haff_decode_chain_sync_words(int n_layers, int hidden);
int haff_decode_chain_status(const unsigned* sync, int n_layers, void* stream);
/* haff_gemm_bf16 with a gather on the A side: logical row m reads A row a_map[m] (0 <= a_map[m] < a_rows). Runs the
 * window-unpartition projection over real tokens only (image_encoder.py:186-188,291-318 drop the padded rows right
 * after proj). */
int haff_gemm_bf16_gather(const void* A, long lda, const int* a_map, long a_rows, const void* W, long ldw, void* C,
                          long ldc, const float* bias, const void* resid, long ldr, const int* row_map, int M, int N,
                          int K, int act, int out_f32, int swiglu, void* stream);
/* haff_gemm_bf16 with a LayerNorm / RMSNorm folded into the product: the normalised activations never exist in HBM.
 *   C[m][n] = act( rstd_m * (sum_k A[m][k] W'[n][k] - mean_m * colsum[n]) + bias'[n] )
 * with W' = W * gamma (column scaling, done once by the caller), bias' = bias + W.beta, colsum[n] = sum_k W'[n][k]
 * (null for RMSNorm), ln_stats[m] = {mean_m, rstd_m} from haff_row_stats. Replaces norm1 -> qkv and norm2 -> lin1 of the
 * SAM blocks (image_encoder.py:179,191), the RMSNorms in front of Llama's qkv / gate-up, CLIP's layer_norm1/2. */
int haff_gemm_bf16_ln(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias,
                      const void* resid, long ldr, const int* row_map, const float* ln_stats, const float* ln_colsum,
                      int M, int N, int K, int act, int out_f32, int swiglu, void* stream);
/* Product (optionally with a folded norm: ln_stats / ln_colsum as above, or both null) whose bf16 output is scattered HEAD-MAJOR:
 * the windowed q|k|v projection of a ViT-H block (image_encoder.py:223-224 the qkv Linear, :263-288 window_partition, :238-239 the
 * reshape into heads) writes q, k, v of every (window, head) as ONE contiguous [tokens][d] block, so the window attention's K / V
 * staging reads whole 128-B lines. Product column n = part * (heads * d) + h * d + c of product row m is stored at
 *   C[part * part_stride + h * head_stride + row_map[m] * d + c]      (bf16 elements; row_map[m] < 0 drops the row)
 * i.e. with row_map[m] = window * heads * n_tok + token, head_stride = n_tok * d, part_stride = (windows + 1) * heads * n_tok * d,
 * C is [part][window][head][token][d] with one spare window whose token 0 holds the pad token (the caller writes it).
 * Whole 256 x 256 tiles only (M % 256 == 0, N % 256 == 0, K % 64 == 0), N == parts * heads * d with parts <= 3, d % 8 == 0;
 * otherwise HAFF_ERR_UNSUPPORTED (-2): the caller keeps the token-major layout (haff_gemm_bf16_ln / haff_gemm_bf16 with a row map). */
int haff_gemm_bf16_heads(const void* A, long lda, const void* W, long ldw, void* C, const float* bias, const int* row_map,
                         const float* ln_stats, const float* ln_colsum, int M, int N, int K, int d, int heads, long part_stride,
                         long head_stride, void* stream);
/* How many workgroups the persistent 8-wave tile launches enqueued ON `stream` take from now on (256 = one per CU, the default;
 * read when a launch is enqueued or captured). Scheduling, not arithmetic: results are bit-identical for every value. The caller
 * lowers it for the launches of ONE stream so that kernels of another stream find free CUs while they run — LisaMI355.evaluate does
 * for the later passes of the SAM encoder (image_encoder.py:107-121) on its encoder stream, which run beside the HBM-bound decode
 * steps (LISA.py:443-450) of the caller's stream. The setting is per stream (round 5's was process-wide: two models or two host
 * threads raced on it): launches on other streams are unaffected. cap: a multiple of 8 in 8..256 (a workgroup's tiles stay on one
 * XCD); any other value changes nothing (a query). Returns the stream's previous cap (256 when it had none), or
 * HAFF_ERR_UNSUPPORTED (-2) when 32 streams are capped at once. Setting 256 releases the stream's entry. */
int haff_gemm_stream_cap(void* stream, int cap);
/* Llama prefill q|k|v projection with rotate-half RoPE and the KV-cache append in the epilogue (transformers
 * LlamaAttention.forward via llava_llama.py:93-102) — replaces haff_gemm_bf16 + haff_rope_cache on prefill-sized batches.
 * A bf16 [B*T][K]; Wp bf16 [3*H*d][K]: the fused q|k|v weights with the rows of every 256-row tile permuted — natural tile row
 * wn*64 + t*16 + i holds logical row (wn>>1)*128 + (t>>1)*64 + (wn&1)*32 + (t&1)*16 + i (wn, t in 0..3, i in 0..15), so that a
 * lane owns a column and its rotate-half partner; q_out bf16 [B*T][ldq] receives the rotated q (H*d columns); kcache / vcache
 * bf16 [B][Tmax][H*d] receive the rotated k and v at rows pos0 .. pos0+T-1; cos_sin f32 [Tmax][d] = cos | sin.
 * d == 128, (H*d) % 256 == 0, K % 64 == 0; otherwise HAFF_ERR_UNSUPPORTED (-2). */
int haff_gemm_bf16_qkv_rope(const void* A, long lda, const void* Wp, long ldw, void* q_out, long ldq, void* kcache,
                            void* vcache, const float* cos_sin, int B, int T, int Tmax, int pos0, int H, int d, int K,
                            void* stream);
/* residual product that also emits the LayerNorm statistics of its output rows (proj / lin2 of a SAM block,
 * image_encoder.py:186-193): C = A.W^T + bias + resid (bf16; C may alias resid; a_map: optional A-side gather as in
 * haff_gemm_bf16_gather); stat_out f32 [M][N/64][2] = {sum, sum of squares} of each 64-column slice of the fp32 results.
 * haff_row_stats_finalize turns them into the {mean, rstd} rows haff_gemm_bf16_ln takes. Whole 256 x 256 tiles only
 * (M % 256 == 0, N % 256 == 0, K % 64 == 0), else HAFF_ERR_UNSUPPORTED (-2): the caller keeps haff_row_stats. */
int haff_gemm_bf16_rowstats(const void* A, long lda, const int* a_map, long a_rows, const void* W, long ldw, void* C,
                            long ldc, const float* bias, const void* resid, long ldr, int M, int N, int K,
                            float* stat_out, void* stream);
/* haff_gemm_bf16_rowstats on an FP32 residual stream (the timed mode's option `fp32_stream`): X32 f32
 * [M][ldx] is read and X32 + A.W^T + bias written back IN PLACE in fp32 (the residual adds of image_encoder.py:186-193 no longer
 * round the stream to bf16 twice per block); C16 bf16 [M][ldc] receives the same sums rounded once — the MFMA operand of the next
 * haff_gemm_bf16_ln, whose folded LayerNorm takes stat_out (f32 [M][N/64][2], from the fp32 sums) through
 * haff_row_stats_finalize. a_map: optional A-side gather. Whole 256 x 256 tiles only (M % 256 == 0, N % 256 == 0, K % 64 == 0),
 * ldx % 4 == 0, bias required; otherwise HAFF_ERR_UNSUPPORTED (-2) / HAFF_ERR_BAD_ARG (-1). */
int haff_gemm_bf16_rowstats32(const void* A, long lda, const int* a_map, long a_rows, const void* W, long ldw, float* X32,
                              long ldx, void* C16, long ldc, const float* bias, int M, int N, int K, float* stat_out,
                              void* stream);
/* parity-mode twin: everything f32. K % 4 == 0, lda/ldw % 4 == 0. */
int haff_gemm_f32(const float* A, long lda, const float* W, long ldw, float* C, long ldc, const float* bias,
                  const float* resid, long ldr, const int* row_map, int M, int N, int K, int act, int swiglu,
                  void* stream);

/* ---- fused attention: out = softmax(scale * q.k^T + bias [+ causal mask]) . v ---------------------------------
 * Replaces SAM Attention.forward + add_decomposed_rel_pos (image_encoder.py:235-260,354-392), CLIPAttention,
 * LlamaAttention (prefill and KV-cached decode) and the SAM decoder Attention (transformer.py:220-242).
 * q/k/v/o: [B][H][N][d] views given by (batch, head, token) strides, unit stride on d; d % 8 == 0 (bf16: d <= 128).
 * causal != 0: key j visible to query i iff j <= i + q_pos0.
 * relh/relw (null = no bias): f32 [B*H][Nq][S] from haff_relpos_tables; key j -> (kh, kw) = (j / S, j % S).
 * bf16 kernel supports S == 64 (one KV tile per key-grid row) or S <= 32. */
int haff_attention_bf16(const void* q, long q_sb, long q_sh, long q_st, const void* k, long k_sb, long k_sh, long k_st,
                        const void* v, long v_sb, long v_sh, long v_st, void* o, long o_sb, long o_sh, long o_st,
                        int B, int H, int Nq, int Nk, int d, float scale, int causal, int q_pos0,
                        const float* relh, const float* relw, int S, void* stream);
int haff_attention_f32(const float* q, long q_sb, long q_sh, long q_st, const float* k, long k_sb, long k_sh, long k_st,
                       const float* v, long v_sb, long v_sh, long v_st, float* o, long o_sb, long o_sh, long o_st,
                       int B, int H, int Nq, int Nk, int d, float scale, int causal, int q_pos0,
                       const float* relh, const float* relw, int S, void* stream);

/* KV-cached decode over RAGGED caches (batched prompts of different lengths; reference padding rules:
 * 2Haff/utils/dataset.py:90-93,144-150): one query per (batch, head); batch b attends its first nk_rows[b] cached keys
 * (DEVICE int32 [B], 1 <= nk_rows[b] <= Nk). q/o: [B][H][d] with (batch, head) strides; k/v as haff_attention_bf16. */
int haff_attention_decode_rows_bf16(const void* q, long q_sb, long q_sh, const void* k, long k_sb, long k_sh, long k_st,
                                    const void* v, long v_sb, long v_sh, long v_st, void* o, long o_sb, long o_sh, int B,
                                    int H, int Nk, int d, float scale, const int* nk_rows, void* stream);
int haff_attention_decode_rows_f32(const float* q, long q_sb, long q_sh, const float* k, long k_sb, long k_sh, long k_st,
                                   const float* v, long v_sb, long v_sh, long v_st, float* o, long o_sb, long o_sh, int B,
                                   int H, int Nk, int d, float scale, const int* nk_rows, void* stream);
/* the same decode position with RoPE and the KV-cache append fused in: replaces haff_rope_cache_rows +
 * haff_attention_decode_rows_bf16 (transformers LlamaAttention with a KV cache, llava_llama.py:93-102). qkv [B][ld]: raw
 * q | k | v of the new position (H x d each); caches [B][Tmax][H*d]; cos_sin f32 [Tmax][d]; nk_rows[b] = new position + 1
 * (DEVICE int32 [B]); out [B][H*d]; d == 128. Results are bit-identical to the two-kernel path. */
int haff_decode_attention_rope_rows_bf16(const void* qkv, long ld, void* kcache, void* vcache, const float* cos_sin, void* out,
                                         int B, int H, int d, int Tmax, float scale, const int* nk_rows, void* stream);
/* decomposed rel-pos terms (image_encoder.py:376-384, from the UNSCALED q, :244-248):
 * relh[bh][q][kh] = q . Rh[qh - kh + S - 1], relw[bh][q][kw] = q . Rw[qw - kw + S - 1]; N = S*S queries.
 * tab_*: [2S-1][d] (f32 for the generic entry, bf16 for the MFMA entry). */
int haff_relpos_tables(const void* q, long q_sb, long q_sh, long q_st, const float* tab_h, const float* tab_w,
                       float* relh, float* relw, int B, int H, int S, int d, int dtype, void* stream);
int haff_relpos_tables_bf16(const void* q, long q_sb, long q_sh, long q_st, const void* tab_h, const void* tab_w,
                            float* relh, float* relw, int B, int H, int S, int d, void* stream);

/* flash attention pair of the fine-tune path (transformers LlamaAttention under autograd; row a16, LISA.py:175-430):
 * haff_attention_lse_bf16 = haff_attention_bf16 (no bias) that also returns lse f32 [B][H][Nq] = log2 sum_k 2^(scale*log2(e)*q.k)
 * over the visible keys; haff_attention_bwd_bf16 = dq, dk, dv from q, k, v, o, dout and lse without the probabilities ever
 * existing in HBM (d == 128; bitwise repeatable: no atomics). q / o / dout / dq: [B][Nq][ld], k / v / dk / dv: [B][Nk][ld],
 * head h at columns h*128; workspace f32 with >= B*H*(Nq + 3 + 128*roundup(Nq, 64)) values; causal needs q_pos0 >= 0. */
int haff_attention_lse_bf16(const void* q, long q_sb, long q_sh, long q_st, const void* k, long k_sb, long k_sh, long k_st,
                            const void* v, long v_sb, long v_sh, long v_st, void* o, long o_sb, long o_sh, long o_st, int B, int H,
                            int Nq, int Nk, int d, float scale, int causal, int q_pos0, float* lse, void* stream);
int haff_attention_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                            void* dq, void* dk, void* dv, float* workspace, long workspace_elems, long ld, int B, int H, int Nq,
                            int Nk, int d, float scale, int causal, int q_pos0, void* stream);

/* TN product of the fine-tune step: out [N1][N2] = A^T . B, A [M][lda] (N1 columns), B [M][ldb] (N2 columns), bf16, contraction
 * over the M rows — the weight gradient dW = dY^T . X of a trainable Linear (torch.nn.Linear under autograd: text_hidden_fcs and
 * the mask decoders, train_ds.py:232-244) without transposed copies of dY and X (csrc/gemm_tn.hip: fragments through the
 * transposing LDS read). N1, N2, lda, ldb % 8 == 0, 16-byte aligned bases, else HAFF_ERR_UNSUPPORTED (the caller then takes
 * haff_transpose + haff_gemm_bf16). out contiguous, bf16 or f32; workspace f32 with haff_gemm_tn_workspace_elems values
 * (split partials, added in index order: repeatable to the bit). */
int haff_gemm_tn_workspace_elems(long M, int N1, int N2);
int haff_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, long M, int N1, int N2, float* workspace,
                      long workspace_elems, void* out, int out_f32, void* stream);

/* rank-r adapter path of the fine-tune step (peft LoRA on q_proj / v_proj: 2Haff/train_ds.py:192-230; RoPE of the adapted
 * projections: llava_llama.py -> transformers LlamaAttention). bf16, head dim d == 128, rank <= 8 per adapter; see csrc/lora.hip.
 * tT / dtT: [16][ld] = the TRANSPOSED rank activations, rows 0..7 the q adapter, 8..15 the v adapter (unused ranks zero);
 * Bq / Bv: [H][8] (ldb == 8, unused rank columns zero); A2: [16][K] = [Aq; 0; Av; 0]; cos_sin f32 [T][128], position = row % T.
 * haff_lora_qkv_rope_fwd: q_out = rope(qkv[:, :H] + scale * t_q.Bq^T), k_out = rope(qkv[:, H:2H]), v_out = qkv[:, 2H:] + scale * t_v.Bv^T
 * haff_lora_qkv_rope_bwd: dqkv [M][3H] = [rope^T dq | rope^T dk | dv] (the adjoint of the q|k|v product's output)
 * haff_lora_dx:           dx (+)= scale * keep .* (dt . A2)   (keep: optional dropout mask values [M][K])
 * haff_lora_tn:           out = scale * sT . big  (sT [R][lds] with R = 8 or 16, lds % 8 == 0, lds >= roundup(M, 16), padding finite; big [M][N]; contraction over the M
 *                         rows; out [j_valid][N] or, transposed, [N][j_valid], bf16 or f32) — the dA / dB contraction without a
 *                         transposed copy of either operand; workspace f32 with haff_lora_tn_workspace_elems values;
 *                         row-block partials are added in index order (deterministic). */
int haff_lora_qkv_rope_fwd(const void* qkv, long ld_qkv, const void* tT, long ldt, const void* Bq, const void* Bv, int ldb,
                           const float* cos_sin, void* q_out, void* k_out, void* v_out, long ldo, long M, int H, int d, int T,
                           float scale, void* stream);
int haff_lora_qkv_rope_bwd(const void* dq, const void* dk, const void* dv, long ld_in, const float* cos_sin, void* dqkv,
                           long ld_out, long M, int H, int d, int T, void* stream);
int haff_lora_dx(const void* dtT, long ldt, const void* A2, long lda, const void* keep, long ldk, void* dx, long ldx,
                 int accumulate, long M, int K, float scale, void* stream);
/* the same with TWO dropout masks — peft gives each adapted Linear its own lora_dropout (train_ds.py:218-230):
 * dx (+)= scale * (keep_q o (dt[0:8]^T . A2[0:8]) + keep_v o (dt[8:16]^T . A2[8:16])). */
int haff_lora_dx2(const void* dtT, long ldt, const void* A2, long lda, const void* keep_q, const void* keep_v, long ldk, void* dx,
                  long ldx, int accumulate, long M, int K, float scale, void* stream);
int haff_lora_tn_workspace_elems(long M, int R, int N);
int haff_lora_tn(const void* sT, long lds, int R, const void* big, long ldb, long M, int N, float* workspace,
                 long workspace_elems, void* out, long ldo, int out_f32, int transposed, int j_valid, float scale, void* stream);

/* fused SAM WINDOW attention with the decomposed rel-pos bias computed in the kernel (one pass over HBM; replaces
 * haff_relpos_tables_bf16 + haff_attention_bf16 for the 28 windowed ViT-H blocks): Attention.forward
 * (image_encoder.py:235-260) + add_decomposed_rel_pos (:354-392) + get_rel_pos with q_size == k_size (:322-351).
 * q/k/v/o: bf16 [n_windows][H][S*S][d] views given by (window, head, token) strides; tab_*: bf16 [2S-1][d].
 * grid_h/grid_w > 0: windows tile images of grid_h x grid_w tokens; window tokens beyond the grid are pads that the
 * caller never wrote — their q/k/v are read from token row pad_token (= projection of a zero token = the qkv bias;
 * window_partition pads AFTER norm1, image_encoder.py:179-183,263-288). grid_h == 0: every token is real.
 * Supported geometry: S == 14, d == 80; otherwise HAFF_ERR_UNSUPPORTED (-2) and the caller takes the generic pair. */
int haff_window_attention_bf16(const void* q, long q_sb, long q_sh, long q_st, const void* k, long k_sb, long k_sh,
                               long k_st, const void* v, long v_sb, long v_sh, long v_st, void* o, long o_sb,
                               long o_sh, long o_st, int n_windows, int H, int S, int d, float scale,
                               const void* tab_h, const void* tab_w, int grid_h, int grid_w, long pad_token,
                               void* stream);

/* fused SAM GLOBAL attention (the 4 global ViT-H blocks, 64 x 64 tokens): Attention.forward (image_encoder.py:235-260) +
 * add_decomposed_rel_pos (:354-392), the rel-pos terms computed in the kernel's prologue from the parameter tables (replaces
 * haff_relpos_tables_bf16 + haff_attention_bf16: no fp32 [B*H][N][S] tables cross HBM). q/k/v/o: bf16 [B][H][S*S][d] views by
 * (batch, head, token) strides; k and v must share one row layout (same strides, v at a non-negative offset behind k — the
 * fused qkv projection output); tab_*: bf16 [2S-1][d]. Supported geometry: S == 64, d == 80; otherwise
 * HAFF_ERR_UNSUPPORTED (-2) and the caller takes the two-kernel path. */
int haff_global_attention_bf16(const void* q, long q_sb, long q_sh, long q_st, const void* k, long k_sb, long k_sh,
                               long k_st, const void* v, long v_sb, long v_sh, long v_st, void* o, long o_sb,
                               long o_sh, long o_st, int B, int H, int S, int d, float scale,
                               const void* tab_h, const void* tab_w, void* stream);

/* ---- row norms ----------------------------------------------------------------------------------------------
 * haff_layernorm: nn.LayerNorm / LayerNorm2d on channels-last rows (common.py:31-43; image_encoder.py:179,191;
 * transformer.py:134-144; CLIP layer norms). in_map (int32[rows], may be null): out row i normalises in row
 * in_map[i]; negative = zero row (window_partition's zero pad AFTER norm1, image_encoder.py:179-183,276-288).
 * haff_rmsnorm: LlamaRMSNorm (fp32 variance). w, b: f32[C]. C % 8 == 0, C <= 8192.
 * dtype: 0 = bf16 rows, 1 = f32 rows, 2 = f32 x -> bf16 y (an fp32 residual stream feeding a bf16 product). */
int haff_layernorm(const void* x, long ldx, void* y, long ldy, const float* w, const float* b, const int* in_map,
                   int rows, int C, float eps, int dtype, void* stream);
int haff_rmsnorm(const void* x, long ldx, void* y, long ldy, const float* w, int rows, int C, float eps, int dtype,
                 void* stream);
/* per-row {mean, rstd} only (rms != 0: {0, rsqrt(mean(x^2)+eps)}): stats f32 [rows][2]; dtype 0 = bf16, 1 = f32. */
int haff_row_stats(const void* x, long ldx, float* stats, int rows, int C, float eps, int rms, int dtype, void* stream);

/* stats[rows][2] = {mean, rstd} from haff_gemm_bf16_rowstats' partials f32 [rows][slots][2] (slots added in order); C = row length. */
int haff_row_stats_finalize(const float* partials, float* stats, int rows, int slots, int C, float eps, void* stream);

/* ---- data movement ------------------------------------------------------------------------------------------ */
/* conv(k=s=P) rows: x [B][Cin][Hin][Win] -> out [B*gh*gw][Kp], column (c*P+ky)*P+kx, zero beyond Cin*P*P.
 * SAM PatchEmbed.proj (image_encoder.py:418-426), CLIP patch_embedding. */
int haff_patchify_nchw(const void* x, void* out, int B, int Cin, int Hin, int Win, int P, int gh, int gw, int Kp,
                       int in_dtype, int out_dtype, void* stream);
/* same rows straight from uint8 NHWC frames [B][Hf][Wf][3]: (x-mean)/std, zero pad right/bottom
 * (inference.preprocess, inference.py:91-105). mean3/std3: HOST pointers to 3 floats (0..255 scale). */
int haff_patchify_u8(const void* frames, void* out, int B, int Hf, int Wf, int P, int gh, int gw, int Kp,
                     const float* mean3, const float* std3, int out_dtype, void* stream);
/* neck 3x3 conv pad 1 (image_encoder.py:100-106): x [B][H][W][C] -> [B*H*W][9*C], column (ky*3+kx)*C+c */
int haff_im2col3x3(const void* x, void* out, int B, int H, int W, int C, int dtype, void* stream);
/* embed_tokens gather + image-feature splice (llava_arch.py:185-208,252-256): ids int64 [B][L] with the sentinel
 * at img_pos[b]; out [B][L+n_img-1][Hd] */
int haff_embed_splice(const long* ids, const int* img_pos, const void* embed, const void* img, void* out, int B, int L,
                      int n_img, int Hd, int dtype, void* stream);
/* rotate-half RoPE on q,k in place (qkv [B*Tq][ld]: q | k | v) + KV-cache append at positions pos0..pos0+Tq-1
 * (caches [B][Tmax][Hkv*d]); cos_sin f32 [Tmax][d] = cos(d/2) | sin(d/2). d % 16 == 0. */
int haff_rope_cache(void* qkv, long ld, void* kcache, void* vcache, const float* cos_sin, int B, int Tq, int Hq, int Hkv,
                    int d, int pos0, int Tmax, int dtype, void* stream);
/* same with a per-row start position: row b's Tq new positions begin at pos0_rows[b] (DEVICE int32 [B];
 * pos0_rows[b] + Tq <= Tmax is the caller's contract) — greedy decode of right-padded prompts of different lengths */
int haff_rope_cache_rows(void* qkv, long ld, void* kcache, void* vcache, const float* cos_sin, int B, int Tq, int Hq,
                         int Hkv, int d, const int* pos0_rows, int Tmax, int dtype, void* stream);
/* greedy token: first index of the row maximum (generate(num_beams=1), LISA.py:443-450) */
int haff_argmax_rows(const float* x, long ld, long* out, int rows, int V, void* stream);
/* One generated token per row of the greedy decode loop, on the device (replaces the per-step host bookkeeping of
 * transformers' generate as used by LISA.py:443-450; sits inside the decode hipGraph). Row b, step s = steps[b]:
 * token = *use_forced ? forced[b][s] : nxt_raw[b]; pad if finished[b]; out_ids[b][lens[b]+s] = token; finished[b] |= token == eos;
 * tok[b] = token; pos[b] = t_rows[b]+s; nk[b] = pos[b]+1; for s >= 1 and h1 != NULL: hidden[b][t_rows[b]+s-1] = h1[b]
 * (row_bytes bytes, multiple of 16); steps[b] = s+1. All pointers device memory; hid_sb = bytes between rows b of hidden. */
int haff_decode_book(const long* nxt_raw, const long* forced, long forced_ld, const int* use_forced, int* steps,
                     unsigned char* finished, long* out_ids, long out_ld, const long* lens, const int* t_rows, long* tok,
                     int* pos, int* nk, const void* h1, void* hidden, long hid_sb, long row_bytes, long pad, long eos, int B,
                     void* stream);
/* out[r] = a[r] + b[r % mod]  (PE adds, transformer.py:166-178; src + dense prompt, mask_decoder.py:141) */
int haff_add_bcast(const void* a, const void* b, void* out, long rows, int C, int mod, int dtype, void* stream);
/* row softmax to f32 (taxonomy head, mask_decoder.py:177) */
int haff_softmax_rows(const void* x, float* out, int rows, int C, int dtype, void* stream);

/* ---- SAM decoder tail + post-processing ------------------------------------------------------------------------
 * haff_upscale_mask: LayerNorm2d(64) -> GELU -> ConvTranspose2d(64->32,k2,s2) -> GELU -> dot with the hypernetwork
 * vector of mask token 0 (mask_decoder.py:58-64,153-165,110-114). up1 [n*h*w][4*64] = output of the first
 * transposed conv as GEMM, columns (dy*2+dx)*64+co; w2 f32 [64][4*32] column (dy2*2+dx2)*32+c2; hyper f32 [n][32];
 * out f32 [n][4h][4w]. */
int haff_upscale_mask(const void* up1, const float* ln_w, const float* ln_b, const float* w2, const float* b2,
                      const float* hyper, float* out, int n_prompts, int h, int w, float eps, int dtype, void* stream);
/* F.interpolate(bilinear, align_corners=False) of the top-left [Hc][Wc] crop of in [N][Hs][Ws] -> out [N][Ho][Wo]
 * (both stages of Sam.postprocess_masks, sam.py:177-188) */
int haff_resize_bilinear(const float* in, float* out, int N, int Hs, int Ws, int Hc, int Wc, int Ho, int Wo,
                         void* stream);
/* ---- device-side frame ingest: the host resizes of the reference (Pillow, on uint8 RGB) and CLIP's normalisation ----
 * One axis of Pillow's antialiased resampling, bit-exact with Image.resize(BILINEAR|BICUBIC): replaces
 * ResizeLongestSide.apply_image (segment_anything/utils/transforms.py:27-34) and the resize inside
 * CLIPImageProcessor.preprocess (third-party transformers; call sites inference.py:233-236, utils/aff_dataset.py:76,228).
 * in u8 [B][Hin][Win][3] -> out u8 [B][Hout][Wout][3]; axis 0 resamples W (Hout == Hin), axis 1 resamples H (Wout == Win);
 * bounds i32 [n_out][2] (first input index, tap count) and coeffs i32 [n_out][ksize] (22-bit fixed point) are DEVICE tables
 * built by preprocess.pil_resample_tables. Horizontal pass first, then vertical. */
int haff_resample_u8(const void* in, void* out, int B, int Hin, int Win, int Hout, int Wout, int axis, const int* bounds,
                     const int* coeffs, int ksize, void* stream);
/* centre crop + 1/255 + (x-mean)/std of CLIPImageProcessor as a f32 [3][256] DEVICE lut: u8 NHWC window (top,left,S,S)
 * -> out [B][3][S][S], out_dtype 0 bf16 / 1 f32 (the images_clip argument of LISAForCausalLM.evaluate, LISA.py:432) */
int haff_clip_normalize_u8(const void* in, void* out, int B, int Hin, int Win, int top, int left, int S, const float* lut,
                           int out_dtype, void* stream);
/* mask > logit_th -> 0/255 bytes (inference.py:294-301 with logit_th = logit(th); chat.py:226 with 0) */
int haff_threshold_masks(const float* in, void* out, long total, float logit_th, void* stream);
/* a15 in one pass — the output gating + thresholds of 2Haff/inference.py:276-334 and chat.py:226-253:
 * planes[t][i] = (argmax(taxonomy) != blank_class && logits[i] > thresholds_host[t]) ? on_value : 0.
 * logits f32 [total] (16-B aligned), planes u8 [n_th][plane_stride] (plane_stride >= total, % 4 == 0), thresholds_host = HOST array of n_th <= 8 LOGIT thresholds
 * (sigmoid(m) > th restated as m > x*(th), exact in f32: 2handedafforder_amd/postprocess.py), taxonomy = DEVICE pointer
 * to the prompt's 4 class probabilities (argmax ties -> first, as torch.argmax) or NULL = gate open; blank_class 1 for the
 * left hand, 0 for the right (inference.py:278,305; chat.py:233,243). on_value 255 (inference PNGs) / 100 (chat JPGs). */
int haff_gate_threshold_masks(const float* logits, void* planes, long total, long plane_stride,
                              const float* thresholds_host, int n_th, int on_value, const float* taxonomy, int blank_class,
                              void* stream);


/* ==== training path (LoRA fine-tune: train_ds.py:489-622 driving model_forward, LISA.py:175-430) ==================
 * Contractions of the backward pass reuse the GEMM kernels: dX = dY.W (NT GEMM on a transposed weight copy),
 * dW = dY^T.X (NT GEMM on transposed operands), attention-shaped products over (batch, head) through the batched
 * entry points. z = zo*nb_inner + zi selects operand base + zo*s?o + zi*s?i (elements). */
int haff_gemm_bf16_batched(const void* A, long lda, long sAo, long sAi, const void* W, long ldw, long sWo, long sWi,
                           void* C, long ldc, long sCo, long sCi, int nb_outer, int nb_inner, int M, int N, int K,
                           int out_f32, void* stream);
int haff_gemm_f32_batched(const float* A, long lda, long sAo, long sAi, const float* W, long ldw, long sWo, long sWi,
                          float* C, long ldc, long sCo, long sCi, int nb_outer, int nb_inner, int M, int N, int K,
                          void* stream);
/* batched transpose: in [z][R][ld_in] (C valid columns) -> out [z][Cp][Rp] contiguous, zero padded */
int haff_transpose(const void* in, long ld_in, long s_in_o, long s_in_i, void* out, int R, int C, int Rp, int Cp,
                   int nb_outer, int nb_inner, int dtype, void* stream);
/* activation forward / adjoint (dx = dy * act'(x)); SwiGLU on the interleaved [gate x16 | up x16] layout */
int haff_act_fwd(const void* x, void* y, long n, int act, int dtype, void* stream);
int haff_act_bwd(const void* x, const void* dy, void* dx, long n, int act, int dtype, void* stream);
int haff_swiglu_fwd(const void* gu, void* y, long M, int F, int dtype, void* stream);
int haff_swiglu_bwd(const void* gu, const void* dy, void* dgu, long M, int F, int dtype, void* stream);
/* out = alpha*a + beta*b (b may be null) */
int haff_axpby(const void* a, const void* b, void* out, long n, float alpha, float beta, int dtype, void* stream);
/* ---- ordered reductions of the fine-tune step: no atomics, bitwise repeatable for a given launch geometry (round 4). Each
 * producer writes per-block partial sums into a caller-provided DEVICE fp32 buffer; haff_reduce_partials adds them in index
 * order: out[j] (+)= sum_{p < n_parts} partials[(j / out_inner) * group_stride + p * part_stride + j % out_inner].
 * Replaces the atomic forms haff_sumsq / haff_mask_loss_stats / haff_colsum / haff_scatter_add_rows inside train_ops / autograd
 * (global gradient norm of train_ds.py:381, the loss sums of LISA.py:16-59, bias gradients, the embedding gradient). ---- */
int haff_reduce_partials(const float* partials, float* out, int n_out, int n_parts, int out_inner, long part_stride,
                         long group_stride, int accumulate, void* stream);
int haff_sumsq_partials(const void* g, float* partials /* >= 1024 */, long n, int dtype, int* n_parts, void* stream);
int haff_mask_loss_stats_partials(const float* x, const float* t, float* partials /* [n_samples][*n_parts][4], n_parts <= 256 */,
                                  int n_samples, long n, float wgt, int* n_parts, void* stream);
int haff_colsum_parts(long R);   /* row blocks haff_colsum_partials writes: partials is [parts][C] */
int haff_colsum_partials(const void* x, float* partials, long R, int C, int dtype, void* stream);
/* dE[id] += sum of the rows with that id, in the order of a STABLE sort (sorted_ids ascending, order = the permutation) */
int haff_scatter_add_rows_sorted(const long* sorted_ids, const long* order, const void* dx, float* dE, long rows, int C, int dtype,
                                 void* stream);
/* out[r][c] = a[r][c] * alpha[r * alpha_stride]: alpha fp32 in DEVICE memory, one scalar (alpha_stride 0) or one per row (1).
 * The upstream gradient of the loss nodes of LISA.py:414-422 (ce / taxonomy CE) applied in fp32 without a host read. */
int haff_scale_dev(const void* a, void* out, long rows, long cols, const float* alpha, long alpha_stride, int dtype, void* stream);
/* out = a * b elementwise (LoRA dropout mask) */
int haff_mul(const void* a, const void* b, void* out, long n, int dtype, void* stream);
/* LayerNorm (rms=0) / RMSNorm (rms=1) adjoint: dx; dyx (f32 [rows][C], may be null) = dy*xhat for the weight grad */
int haff_norm_bwd(const void* x, const void* dy, const float* w, void* dx, float* dyx, int rows, int C, float eps, int rms,
                  int dtype, void* stream);
/* the same with the residual branch's gradient folded in: x of a pre-norm block feeds the norm AND the residual add (transformers'
 * LlamaDecoderLayer; image_encoder.py:186-193), so dx = norm adjoint(dy) + add (add: same shape / dtype as x; dx may alias it). */
int haff_norm_bwd_add(const void* x, const void* dy, const float* w, const void* add, void* dx, float* dyx, int rows, int C,
                      float eps, int rms, int dtype, void* stream);
/* out[c] += sum_r x[r][c] (bias / norm-weight gradients); out f32, zeroed by the caller */
int haff_colsum(const void* x, float* out, long R, int C, int dtype, void* stream);
/* row softmax of scale*s (+ causal mask, row r = query r % Nq, key j visible iff j <= q + q_pos0) and its adjoint
 * dS = scale * P o (dP - rowsum(dP o P)); columns >= Nk are written as zeros (K-padding of the following GEMMs) */
int haff_softmax_fwd(const float* s, long ld, void* p, long ldp, long rows, int Nq, int Nk, float scale, int causal,
                     int q_pos0, int dtype, void* stream);
int haff_softmax_bwd(const void* p, long ldp, const float* dp, long ld, void* ds, long rows, int Nk, float scale, int dtype,
                     void* stream);
/* rotate-half RoPE on x [rows][H][d], position pos0 + row % Tlen; adjoint != 0 applies the transpose rotation */
int haff_rope(const void* x, long ldx, void* y, long ldy, const float* cos_sin, long rows, int Tlen, int H, int d, int pos0,
              int adjoint, int dtype, void* stream);
/* fused shift-free cross entropy: row_loss[r] = lse(logits[r]) - logits[r][labels[r]] (0 when labels[r] < 0);
 * dlogits (may be null) = (softmax - onehot) * gscale on valid rows (llava_llama.py:108-118 after the caller's shift) */
int haff_cross_entropy(const void* logits, long ld, const long* labels, float* row_loss, void* dlogits, long rows, int V,
                       float gscale, int dtype, void* stream);
/* per-sample mask losses (LISA.py:16-59) on f32 logits scaled by wgt: stats[s] = {bce_sum, sum(p*t), sum(p), sum(t)}
 * (zeroed by caller); grad: dx = wgt*(c_bce*(p-t)/n + c_dice*d dice/dz) */
int haff_mask_loss_stats(const float* x, const float* t, float* stats, int n_samples, long n, float wgt, void* stream);
int haff_mask_loss_grad(const float* x, const float* t, const float* stats, float* dx, int n_samples, long n, float wgt,
                        float c_bce, float c_dice, void* stream);
/* the same with the upstream gradients of {bce, dice} read from device memory (coef f32 [n_samples][2]): no host read-back */
int haff_mask_loss_grad_dev(const float* x, const float* t, const float* stats, float* dx, int n_samples, long n, float wgt,
                            const float* coef, void* stream);
/* taxonomy loss (LISA.py:414-417): CrossEntropyLoss on the already soft-maxed probabilities p = softmax(z) with a soft
 * target t: loss[r] = -sum_c t_c log_softmax(p)_c; probs / dz may be null; C <= 8 */
int haff_taxonomy_ce(const float* z, const float* t, float* probs, float* loss, float* dz, int rows, int C, void* stream);
/* adjoint of haff_resize_bilinear (din zeroed by caller) */
int haff_resize_bilinear_bwd(const float* dout, float* din, int N, int Hs, int Ws, int Hc, int Wc, int Ho, int Wo,
                             void* stream);
/* embedding gradient dE[ids[r]] += dx[r] (f32 accumulator; ids < 0 skipped) */
int haff_scatter_add_rows(const long* ids, const void* dx, float* dE, long rows, int C, int dtype, void* stream);
/* out += sum(g^2) (gradient-norm clipping, train_ds.py:370) */
int haff_sumsq(const void* g, float* out, long n, int dtype, void* stream);
/* fused AdamW on fp32 master weights (+ optional bf16 copy), torch semantics, gradient pre-scaled by gscale */
int haff_adamw_step(float* master, float* m, float* v, const void* g, void* param_lp, long n, float lr, float beta1,
                    float beta2, float eps, float wd, int step, float gscale, int g_dtype, int lp_dtype, void* stream);
/* the same, the gradient scale multiplied by *gscale_dev (device f32: the clip coefficient computed on the device) */
int haff_adamw_step_dev(float* master, float* m, float* v, const void* g, void* param_lp, long n, float lr, float beta1,
                        float beta2, float eps, float wd, int step, float gscale, const float* gscale_dev, int g_dtype,
                        int lp_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HAFF_HIP_H */
