// bf16 MFMA GEMM with fused epilogues — the workhorse of the 2Haff hot path on MI355X.
//
//   C[M,N] = epi( A[M,K] · W[N,K]^T )      A, W bf16 row-major (torch nn.Linear layout), fp32 accumulate
//
// Replaces every nn.Linear / 1x1-conv / patchify-conv on the reference path:
//   SAM ViT-H qkv/proj/MLP   (2Haff/model/segment_anything/modeling/image_encoder.py:223-224,258; common.py:13-26)
//   CLIP q/k/v/out/fc1/fc2   (transformers CLIPEncoderLayer, called at clip_encoder.py:53-56)
//   Llama q/k/v/o/gate/up/down/lm_head (transformers LlamaDecoderLayer, called at llava_llama.py:93-105)
//   mm_projector (llava_arch.py:35), text_hidden_fcs (LISA.py:95-101), SAM decoder linears (transformer.py:206-209)
//
// Design (gfx950): two tiles of one kernel template — 256x256x64 with 8 waves (2x4, 128x64 per wave; one persistent
// workgroup per CU, 130 KiB LDS) for anything that fills the chip with it, 128x128x64 with 4 waves (2x2, 64x64 per wave; two
// workgroups per CU) for small / ragged problems, K tails and the batched entry point. MFMA is v_mfma_f32_16x16x32_bf16.
// Both operands are staged HBM->LDS with global_load_lds_dwordx4 (no VGPR round trip); the LDS image is lane-linear and
// the XOR swizzle (chunk ^= row&7) is applied on the per-lane SOURCE address and again on the ds_read_b128 address, which
// makes the fragment reads bank-conflict free (SQ_LDS_BANK_CONFLICT = 0).
// The 8-wave tile runs a PING-PONG RING LOOP (round 3): its two wave groups sit one workgroup barrier apart, so on every
// SIMD one wave multiplies (32 MFMAs on fragments it holds in registers) while its partner reads its next fragments from
// LDS and issues its share of the operand requests; a K-tile is requested as four 16 KiB quarters spread over the loop, up
// to two K-tiles ahead, and the one wait per K-tile is a COUNTED s_waitcnt vmcnt(4) — the request stream never drains and
// runs on across the tile boundary into the workgroup's next tile. (Rounds 1-2 drained it with vmcnt(0) once per K-tile
// because a counted wait had once been blamed for stale data; tools/probes/vmcnt_order_probe.hip and
// tools/vmcnt_forensics.py show LDS-DMA leaves vmcnt in issue order and that failure was a write-after-read race.)
// The MFMA is issued "swapped" (W rows as the A operand, activation rows as the B operand): each lane then holds 4
// CONSECUTIVE output columns of one output row, so SwiGLU pairs (gate, up) land in the same lane.
// Epilogue: interior tiles leave from registers — one v_permlane16_swap per register pairs two neighbouring 4-column
// chunks into 16-B stores, four lanes cover 64 contiguous bytes of a row, the residual is prefetched one pass ahead behind
// counted waits and the bias rides in as one more LDS-DMA a K loop earlier. Ragged / unaligned tiles go through a
// wave-private fp32 LDS image and leave as whole 128-B row segments.
// Workgroup ids are remapped XCD-aware (ids that share an XCD get neighbouring tiles) and grouped up to 8 M-tiles deep so
// the 4 MiB per-XCD L2 holds the A and W panels the concurrently running tiles share (read hit rate 87 % measured).
#include <type_traits>

#include <cstdlib>
#include "haff_common.h"
#include <mutex>

// Tuning hooks (ablation switches HAFF_EXP_*, phase traces HAFF_GEMM_TRACE / _TRACE2, A/B switches HAFF_EPI_LDS /
// HAFF_GEMM_NO_NT / HAFF_GEMM_GELU_SCALAR, environment overrides of the raster) exist only in builds made with
// -DHAFF_TUNING (tools/build_gemm_variant.sh); the product library carries none of them and reads no environment.
#ifndef HAFF_TUNING
#undef HAFF_EXP_NODMA
#undef HAFF_EXP_NOREAD
#undef HAFF_EXP_NOSTORE
#undef HAFF_EXP_NOEPI
#undef HAFF_EXP_SAMETILE
#undef HAFF_GEMM_TRACE
#undef HAFF_GEMM_TRACE2
#undef HAFF_GEMM_TRACE3
#undef HAFF_EPI_LDS
#undef HAFF_GEMM_NO_NT
#undef HAFF_GEMM_GELU_SCALAR
#endif

namespace {

constexpr int BK = 64;


__device__ __attribute__((aligned(16))) unsigned int haff_zero_page[8];  // 32 B of zeros for K-tail chunks

struct GemmArgs {
  const bf16_t* A; long lda;
  const bf16_t* W; long ldw;
  void* C; long ldc;
  const float* bias;
  const void* resid; long ldr;
  const int* row_map;
  const int* a_map;   // optional gather: logical A row m is stored at A row a_map[m]
  int group_m;        // M-tiles per raster group (L2 reuse window), chosen by the launcher
  // LayerNorm / RMSNorm folded into the product (gamma pre-multiplied into W, beta into bias by the caller):
  //   C[m][n] = act( rstd_m * (acc[m][n] - mean_m * colsum[n]) + bias[n] ),  ln_stats[m] = {mean_m, rstd_m}
  // colsum[n] = sum_k W[n][k] (of the gamma-scaled, bf16-rounded weights); null colsum (RMSNorm): mean term dropped.
  const float* ln_stats;
  const float* ln_colsum;
  int M, N, K;
  int act, out_f32, swiglu;
  // batched launches (blockIdx.y = z): operand z lives at base + (z / nb_inner) * s?o + (z % nb_inner) * s?i elements
  int nb_inner;
  long sAo, sAi, sWo, sWi, sCo, sCi;
  int deep_k;         // 128x128 tile: K loop with TWO K-tiles of DMA in flight (two barriers per K-tile); set by the launcher
  int k_total;        // split-K with a short last slice: inner batch zi covers K columns [zi*K, min((zi+1)*K, k_total)); 0 = every slice has K
  int nt_out;         // non-temporal output stores (large outputs; chosen by the launcher)
  // weight-streaming kernel with K split over blockIdx.y: each slice writes its fp32 partial tile to ws[slice][M][N]
  // (caller-provided workspace) and skinny_reduce_kernel applies the epilogue to the slice sum, in slice order
  float* ws;
  int ksplit;
  // RMSNorm carried between decode-sized products without a norm kernel (weight-streaming kernel, M <= 16):
  //   producer (ssq_out): workgroup b writes ssq_out[b][m] = sum over ITS output columns of bf16(C[m][n])^2, m < 16;
  //   consumer (ssq_in):  rstd_m = rsqrt(sum_b ssq_in[b][m] / K + ssq_eps) (b < ssq_n, fixed order) scales row m of the
  //   product of the RAW activations with gamma-folded weights — the same fold as ln_stats, with the statistic assembled
  //   from the producer's partials instead of a pass over the row.
  float* ssq_out;
  const float* ssq_in;
  int ssq_n;
  float ssq_eps;
  // LayerNorm statistics of the OUTPUT rows, emitted by the producer (8-wave tile, register epilogue with a bf16 residual,
  // whole interior tiles only — haff_gemm_bf16_rowstats): wave (n-tile, wn) writes {sum, sum of squares} of its 64 output
  // columns of row m to stat_out[m][stat_slots][2], slot = first column / 64; haff_row_stats_finalize adds the slots in
  // order. The consumer product then needs no pass over the rows it normalises.
  float* stat_out;
  int stat_slots;
  // fp32 RESIDUAL STREAM beside the bf16 one (round 6, haff_gemm_bf16_rowstats32; specialised 8-wave instance only): the residual
  // is read from and the sum written back to res32[m][ld32] in fp32 (in place), and C receives the bf16 copy of the same values
  // (the next product's MFMA operand); stat_out as above, from the fp32 values. The stream is rounded to bf16 ONCE per consumer
  // instead of once per residual add (64 times through a ViT-H).
  float* res32;
  long ld32;
  // Llama prefill q|k|v projection with RoPE and the KV-cache append in the epilogue (haff_gemm_bf16_qkv_rope; 8-wave tile,
  // register epilogue, bf16, no residual). W's rows arrive PERMUTED inside every 256-row tile so that a lane holds column c
  // of a head in col-block t and its rotate-half partner c + 64 in col-block t + 2 (natural tile column wn*64 + t*16 + i is
  // logical column (wn>>1)*128 + (t>>1)*64 + (wn&1)*32 + (t&1)*16 + i). Tiles of the q block: rotated, stored to C (the q
  // buffer, ldc); k block: rotated, stored to rope_k; v block: stored to rope_v; caches [B][rope_tmax][rope_hd], row (b, pos0 + t)
  // for product row m = b * rope_t + t. rope_cs: f32 [rope_tmax][128] = cos(0..63) | sin(0..63).
  const float* rope_cs;
  void *rope_k, *rope_v;
  int rope_t, rope_tmax, rope_pos0, rope_hd;
  // HEAD-MAJOR scatter of the output (haff_gemm_bf16_heads; 8-wave tile, register epilogue, bf16, row map, no residual): product
  // column n = part * hm_hd + h * hm_d + c of product row m is stored at C[part * hm_part + h * hm_head + row_map[m] * hm_d + c]
  // (elements). hm_d == 0: off.
  int hm_d, hm_hd;
  long hm_part, hm_head;
};

// Epilogue activations of the throughput (bf16) path. GELU matters for the K=1280 SAM MLP GEMM, whose epilogue touches
// 5120 columns per row: a polynomial erf (below) instead of ocml erff. The fp32 parity kernel keeps erff.
template <int ACT>
__device__ __forceinline__ float gemm_act(float x) {
  if constexpr (ACT == HAFF_ACT_GELU) {
    // erf(z) = z * P(z^2) on |z| <= 3 (clamped: 1 - erf(3) = 2.2e-5), P = degree-8 Chebyshev fit: no transcendental
    // (v_exp / v_rcp issue at quarter rate: the Abramowitz-Stegun 7.1.26 form used before cost 14 % of the SAM lin1
    // GEMM, this one 8 %; writing it on float2 did not make hipcc emit v_pk_fma_f32 and measured the same).
    // |gelu error| < 9e-5, two orders below the bf16 rounding of the output.
    const float z = __builtin_amdgcn_fmed3f(x * 0.70710678118654752440f, -3.0f, 3.0f);
    const float u = z * z;
    float pz = 4.9182759198629356e-08f;
    pz = pz * u - 2.2677306787954876e-06f;
    pz = pz * u + 4.6147291868692264e-05f;
    pz = pz * u - 0.0005535572418011725f;
    pz = pz * u + 0.004437862429767847f;
    pz = pz * u - 0.02564961276948452f;
    pz = pz * u + 0.11186250299215317f;
    pz = pz * u - 0.3758186101913452f;
    pz = pz * u + 1.1283628940582275f;
    const float hx = 0.5f * x;
    return hx + hx * (z * pz);
  } else if constexpr (ACT == HAFF_ACT_QUICK_GELU) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
  } else if constexpr (ACT == HAFF_ACT_RELU) {
    return fmaxf(x, 0.0f);
  } else if constexpr (ACT == HAFF_ACT_SILU) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
  } else {
    return x;
  }
}

// The same polynomial on two values at once: v_pk_fma_f32 / v_pk_mul_f32 (the epilogue has no MFMA beside it, where the
// packed forms are an anti-lever; here they halve the instruction count of the 13-op chain). -DHAFF_GEMM_GELU_SCALAR: the
// scalar form, for A/B.
typedef float haff_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ haff_f2 gelu_pair(haff_f2 x) {
  const haff_f2 lo = {-3.0f, -3.0f}, hi = {3.0f, 3.0f};
  haff_f2 z = x * 0.70710678118654752440f;
  z = __builtin_elementwise_min(__builtin_elementwise_max(z, lo), hi);
  const haff_f2 u = z * z;
  haff_f2 pz = {4.9182759198629356e-08f, 4.9182759198629356e-08f};
  auto step = [&](float c) { const haff_f2 cc = {c, c}; pz = __builtin_elementwise_fma(pz, u, cc); };
  step(-2.2677306787954876e-06f);
  step(4.6147291868692264e-05f);
  step(-0.0005535572418011725f);
  step(0.004437862429767847f);
  step(-0.02564961276948452f);
  step(0.11186250299215317f);
  step(-0.3758186101913452f);
  step(1.1283628940582275f);
  const haff_f2 hx = x * 0.5f;
  return __builtin_elementwise_fma(hx, z * pz, hx);
}

#ifdef HAFF_GEMM_TRACE2  // fine timestamps (100 MHz wall clock) of waves 0 and 4 around the epilogue of a workgroup's LAST tile
__device__ unsigned long long haff_gemm_trace2_buf[256 * 2 * 16];
#define HAFF_TRACE2(i) do { if ((tid & 255) == 0 && blockIdx.x < 256 && blockIdx.y == 0) haff_gemm_trace2_buf[(blockIdx.x * 2 + (tid >> 8)) * 16 + (i)] = wall_clock64(); } while (0)
extern "C" int haff_gemm_trace2_read(unsigned long long* host, int n_words) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(haff_gemm_trace2_buf), sizeof(unsigned long long) * n_words) == hipSuccess ? 0 : 1;
}
#else
#define HAFF_TRACE2(i) do {} while (0)
#endif
#ifdef HAFF_GEMM_TRACE3  // round 5: 32 stamps per (workgroup, wave 0 / wave 4) of a MIDDLE tile (one that has a successor): tile start, the
// last K-tile's slots and its wait for the next tile's first K-tile, the epilogue's passes (pass 0 and 4 in four sub-steps), the
// barrier behind it. tools/gemm_trace3.py
__device__ unsigned long long haff_gemm_trace3_buf[256 * 2 * 32];
// stamps go to LDS (a global store per stamp sat in vmcnt and the next vmcnt(0) of the traced wave waited for it: "pre" read 1 us)
// and are flushed once per tile, behind the barrier that follows the epilogue
#define HAFF_TRACE3(i) do { __builtin_amdgcn_sched_barrier(0); if (has_next && (tid & 255) == 0) haff_t3s[tid >> 8][(i)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define HAFF_TRACE3_FLUSH() do { if ((tid & 255) < 32 && blockIdx.x < 256 && blockIdx.y == 0) haff_gemm_trace3_buf[(blockIdx.x * 2 + (tid >> 8)) * 32 + (tid & 255)] = haff_t3s[tid >> 8][tid & 255]; } while (0)
extern "C" int haff_gemm_trace3_read(unsigned long long* host, int n_words) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(haff_gemm_trace3_buf), sizeof(unsigned long long) * n_words) == hipSuccess ? 0 : 1;
}
#else
#define HAFF_TRACE3(i) do {} while (0)
#define HAFF_TRACE3_FLUSH() do {} while (0)
#endif
#ifdef HAFF_GEMM_TRACE  // phase timestamps (100 MHz wall clock) of each workgroup's first tile, for tools/gemm_trace.py
__device__ unsigned long long haff_gemm_trace_buf[8192 * 8];
#define HAFF_TRACE(i) do { if (tid == 0 && blockIdx.x < 8192 && blockIdx.y == 0) haff_gemm_trace_buf[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define HAFF_TRACE(i) do {} while (0)
#endif

// Output stores of the tile kernel. nt: non-temporal — a C tile of a LARGE output is written once and next read by
// another kernel after hundreds of MB of other traffic; keeping it out of the XCD's L2 leaves the A/W panels resident
// (+3 % on 131072x5120x1280, +6 % with a residual epilogue; neutral on the K >= 4096 shapes; tools/gemm_variant.py).
// The launcher sets it for outputs of 64 MB and more (decode-sized outputs are re-read from L2 by the next kernel).
typedef unsigned int haff_u32x4 __attribute__((ext_vector_type(4)));
typedef float haff_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store8_c(bf16_t* p, const float (&v)[8], bool nt) {
  if (nt) {
    haff_u32x4 r = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    asm volatile("");   // keeps hipcc from merging this store with the plain one of the other branch (the hint would be dropped)
    __builtin_nontemporal_store(r, reinterpret_cast<haff_u32x4*>(p));
    asm volatile("");
  } else {
    store8(p, v);
  }
}
__device__ __forceinline__ void store8_c(float* p, const float (&v)[8], bool nt) {
  if (nt) {
    haff_f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    asm volatile("");
    __builtin_nontemporal_store(a, reinterpret_cast<haff_f32x4*>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<haff_f32x4*>(p + 4));
    asm volatile("");
  } else {
    store8(p, v);
  }
}

// v_permlane16_swap_b32: lanes 16..31 of x trade places with lanes 0..15 of y, lanes 48..63 of x with lanes 32..47 of y
// (checked on MI355X). Inline asm, not __builtin_amdgcn_permlane16_swap: with float operands hipcc (ROCm 7.2) used the
// builtin's FIRST result where the second was asked for (every second group of four output columns came out wrong).
// The s_nop pair covers the VALU-write -> permlane-read and permlane-write -> VALU-read wait states the compiler would
// otherwise insert itself (it pads nothing around inline asm).
__device__ __forceinline__ void permlane16_swap(unsigned& x, unsigned& y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}

// v_permlane32_swap_b32: lanes 32..63 of x trade places with lanes 0..31 of y.
__device__ __forceinline__ void permlane32_swap(unsigned& x, unsigned& y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
// sum over the four lanes {l, l^16, l^32, l^48}, in every one of them, on the VALU alone (no ds_bpermute, no lgkmcnt wait):
// a swap of two copies leaves (x_row0, x_row0, x_row2, x_row2) and (x_row1, x_row1, x_row3, x_row3); the same by halves
__device__ __forceinline__ float quad_row_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane16_swap(a, b);
  v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  a = __builtin_bit_cast(unsigned, v);
  b = a;
  permlane32_swap(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// BM x BN output tile per workgroup, WM x WN waves (each wave owns (BM/WM) x (BN/WN), BN/WN == 64).
//   <128,128,2,2>: 256 threads, 64 KiB LDS, 2 workgroups/CU  — small / ragged problems
//   <256,256,2,4>: 512 threads, 128 KiB LDS, 1 workgroup/CU  — half the L2->LDS bytes per MFMA (the 128^2 tile needs
//                  ~64 B/clk/CU from L2 at full MFMA rate, more than the ~56 B/clk/CU the L2 can deliver)
//   <192,256,2,4>: the same ring loop on 96 x 64 per wave (round 4) for row counts that quantise badly into 256-row tiles:
//                  the fine-tune step's 2808-row products are 11 x 16 = 176 tiles of 256^2 on 256 CUs but 15 x 16 = 240 of these
// Compile-time specialisation of the 8-wave tile's epilogue (round 5). One kernel instance served every epilogue variant at
// run time until round 4: 33 000 instructions (250 KB of code for a 64 KB instruction cache), 92 spilled SGPRs, and ~0.9 us
// between the end of a tile's K loop and the first pass of its epilogue spent finding the way through it
// (profiles/r5_gemm_trace3_tile_boundary.txt). FLAGS = GF_SPEC | features: the launcher has checked on the HOST that every tile
// is interior (M % BM == 0, N % BN == 0), every pointer 16-B aligned, ldc / ldr multiples of 8 and the launch not batched, so the
// instance carries the register epilogue of exactly its feature set and nothing else (same arithmetic, same results; the
// generic instance FLAGS = 0 keeps every path and serves the rest). Measured on plain products: +1.5 ... +3.9 % per launch.
enum : unsigned {
  GF_SPEC = 1u, GF_BIAS = 2u, GF_LN = 4u, GF_CSUM = 8u, GF_RES = 16u, GF_STAT = 32u, GF_MAP = 64u, GF_HM = 128u, GF_ROPE = 256u,
  GF_RAGM = 512u,   // the row count need not be a multiple of BM: the M-side interior tests stay at run time (Llama's 64 x 291 rows)
  GF_RES32 = 1024u, // with GF_RES: the residual stream is fp32 (GemmArgs::res32), C gets its bf16 copy
  GF_ACT_SHIFT = 16
};

template <int BM, int BN, int WM, int WN, bool OUT_F32, bool SWIGLU, unsigned FLAGS = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kernel(GemmArgs p) {
  constexpr bool PP = (WM * WN == 8);   // the 8-wave tile runs the persistent ping-pong ring loop
  constexpr bool SPEC = (FLAGS & GF_SPEC) != 0;
  constexpr bool MFULL = SPEC && !(FLAGS & GF_RAGM);   // every M-tile is whole
  static_assert(!SPEC || (PP && !OUT_F32 && (BM == 256 || !(FLAGS & (GF_LN | GF_MAP)))),
                "specialised epilogues exist for the bf16 8-wave tiles (the 192-row form without the norm fold / row map, whose staging is 256 rows wide)");
  // feature tests: compile-time constants in a specialised instance, the argument block's pointers otherwise
  const bool has_bias = SPEC ? (FLAGS & GF_BIAS) != 0 : p.bias != nullptr;
  const bool has_ln = SPEC ? (FLAGS & GF_LN) != 0 : p.ln_stats != nullptr;
  const bool has_csum = SPEC ? (FLAGS & GF_CSUM) != 0 : p.ln_colsum != nullptr;
  const bool has_res = SPEC ? (FLAGS & GF_RES) != 0 : p.resid != nullptr;
  const bool has_stat = SPEC ? (FLAGS & GF_STAT) != 0 : p.stat_out != nullptr;
  const bool has_map = SPEC ? (FLAGS & GF_MAP) != 0 : p.row_map != nullptr;
  const bool has_hm = SPEC ? (FLAGS & GF_HM) != 0 : p.hm_d != 0;
  const bool has_rope = SPEC ? (FLAGS & GF_ROPE) != 0 : p.rope_cs != nullptr;
  const int act = SPEC ? (int)((FLAGS >> GF_ACT_SHIFT) & 7u) : p.act;
  auto al16 = [](const void* q) { return SPEC || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };   // (checked on the host for SPEC)
  static_assert((BM == 256 && BN == 256 && WM == 2 && WN == 4) || (BM == 192 && BN == 256 && WM == 2 && WN == 4) ||
                (BM == 128 && BN == 128 && WM == 2 && WN == 2), "three tiles");
  constexpr int NTHREADS = 64 * WM * WN;
  constexpr int A_ELEMS = BM * BK, W_ELEMS = BN * BK;
  constexpr int STAGE_ELEMS = A_ELEMS + W_ELEMS;
  constexpr int NA = BM * 8 / NTHREADS, NW = BN * 8 / NTHREADS;  // 16-B chunks per thread per K-tile
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;             // 16x16 MFMA tiles per wave
  constexpr int WNC = BN / WN;                                    // columns per wave: 64, or 128 for the 4-wave 256^2 tile
  static_assert(WNC == 64, "wave tile width");
  // [buf][A | W] (+ PP: the tile's bias 1 KB | folded-norm row statistics {mean, rstd} x 256 rows 2 KB | column sums 1 KB |
  // output row map 1 KB)
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE_ELEMS + (PP ? 2560 : 0)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef HAFF_GEMM_TRACE3
  __shared__ unsigned long long haff_t3s[2][32];
#endif
#if defined(HAFF_TUNING) && defined(HAFF_EXP_FORCE_PLAIN)   // experiment: what a compile-time-specialised epilogue would cost (bias only, interior tiles)
  p.act = 0; p.ln_stats = nullptr; p.ln_colsum = nullptr; p.resid = nullptr; p.row_map = nullptr; p.a_map = nullptr; p.rope_cs = nullptr;
  p.stat_out = nullptr; p.hm_d = 0; p.nb_inner = 0; p.nt_out = 0;
#endif
  if (!SPEC && p.nb_inner > 0) {
    const int zo = blockIdx.y / p.nb_inner, zi = blockIdx.y - zo * p.nb_inner;
    p.A += zo * p.sAo + zi * p.sAi;
    p.W += zo * p.sWo + zi * p.sWi;
    const long coff = zo * p.sCo + zi * p.sCi;
    if (p.k_total) p.K = min(p.K, p.k_total - zi * p.K);
    p.C = OUT_F32 ? static_cast<void*>(reinterpret_cast<float*>(p.C) + coff)
                  : static_cast<void*>(reinterpret_cast<bf16_t*>(p.C) + coff);
  }

  // ---- workgroup -> output tile ----
  // Ids are remapped XCD-aware (ids equal mod 8 share an XCD/L2) and grouped GROUP_M row-tiles deep so concurrently
  // running tiles share A and W panels in the 4 MiB per-XCD L2.
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  auto tile_origin = [&](int t, int& tm0, int& tn0) {
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    const int GROUP_M = p.group_m;
    const int per_group = GROUP_M * tiles_n;
    const int g = lin / per_group;
    const int first_m = g * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_g = lin - g * per_group;
    tm0 = (first_m + in_g % gsz) * BM;
    tn0 = (in_g / gsz) * BN;
#ifdef HAFF_EXP_SAMETILE   // timing experiment: every workgroup computes tile (0, 0): all operand bytes come from L2
    tm0 = 0; tn0 = 0;
#endif
  };
  // The 8-wave tile is PERSISTENT when the launcher caps the grid (one workgroup per CU): a workgroup takes tiles
  // blockIdx.x, blockIdx.x + gridDim.x, ... (same XCD, consecutive waves of its raster), and the first K-tile of the next
  // tile is requested BEFORE the epilogue of the current one into the LDS buffer the epilogue does not use, so the
  // first-load latency and the store drain of tile i sit under each other instead of in sequence (K = 1280: the epilogue
  // and the first load were ~7 of a tile's ~41 us).
  int tile = blockIdx.x;
  int m0, n0;
  tile_origin(tile, m0, n0);
  HAFF_TRACE(0);

  // ---- per-thread staging coordinates: 16-B chunks of the K-tile ----
  // LDS position pos = i*NTHREADS + tid (lane-linear); row = pos>>3; logical chunk = (pos&7) ^ (row&7)
  // The 8-wave tile is only launched for K % 64 == 0 and operands under 4 GiB, so it keeps 32-bit offsets from a
  // uniform base (saves 16 VGPRs for the ping-pong loop); the 4-wave tile keeps pointers and the zero-page K tail.
  constexpr bool OFF32 = PP;
  const bf16_t* a_src[OFF32 ? 1 : NA];
  const bf16_t* w_src[OFF32 ? 1 : NW];
  int a_kcol[OFF32 ? 1 : NA], w_kcol[OFF32 ? 1 : NW];
  unsigned a_off[OFF32 ? NA : 1], w_off[OFF32 ? NW : 1];
  auto stage_coords = [&](int tm0, int tn0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int pos = i * NTHREADS + tid, row = pos >> 3;
      const int kcol = ((pos & 7) ^ (row & 7)) * 8;
      int arow = min(tm0 + row, p.M - 1);
      if (p.a_map) arow = p.a_map[arow];
      if constexpr (OFF32) {
        a_off[i] = ((unsigned)arow * (unsigned)p.lda + kcol) * 2u;  // bytes
      } else {
        a_kcol[i] = kcol;
        a_src[i] = p.A + (long)arow * p.lda + kcol;
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int pos = i * NTHREADS + tid, row = pos >> 3;
      const int kcol = ((pos & 7) ^ (row & 7)) * 8;
      if constexpr (OFF32) {
        w_off[i] = ((unsigned)(min(tn0 + row, p.N - 1)) * (unsigned)p.ldw + kcol) * 2u;
      } else {
        w_kcol[i] = kcol;
        w_src[i] = p.W + (long)min(tn0 + row, p.N - 1) * p.ldw + kcol;
      }
    }
  };
  stage_coords(m0, n0);
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(haff_zero_page);

  auto stage = [&](int buf, int k0) {
    bf16_t* sA = smem + buf * STAGE_ELEMS;
    bf16_t* sW = sA + A_ELEMS;
    const bf16_t* a_base = p.A + k0;
    const bf16_t* w_base = p.W + k0;
    if constexpr (OFF32) {  // keep SGPR-base + 32-bit-VGPR-offset addressing (no hoisted 64-bit pointer per chunk)
      asm volatile("" : "+s"(a_base));
      asm volatile("" : "+s"(w_base));
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bf16_t* ga;
      if constexpr (OFF32) {
        unsigned o = a_off[i];
        asm volatile("" : "+v"(o));  // keeps the zero-extension next to the load so it selects the saddr form
        ga = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(a_base) + o);
      } else {
        ga = (k0 + a_kcol[i]) < p.K ? a_src[i] + k0 : zero_src;
      }
      bf16_t* la = sA + (i * NTHREADS + wave * 64) * 8;  // wave-uniform LDS base; hardware adds lane*16
      __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)la, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const bf16_t* gw;
      if constexpr (OFF32) {
        unsigned o = w_off[i];
        asm volatile("" : "+v"(o));
        gw = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(w_base) + o);
      } else {
        gw = (k0 + w_kcol[i]) < p.K ? w_src[i] + k0 : zero_src;
      }
      bf16_t* lw = sW + (i * NTHREADS + wave * 64) * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)gw, (lptr_t)lw, 16, 0, 0);
    }
  };

  // PP loop: a K-tile is staged as four 16 KiB QUARTERS (A rows 0-127 / 128-255, W rows 0-127 / 128-255 of the stage), two
  // DMA instructions per wave each, so the request stream can be spread over the phases of the K loop.
  auto stage_a_q = [&](int buf, int k0, auto half) {
    if constexpr (OFF32) {
      constexpr int H = decltype(half)::value;
      bf16_t* sA = smem + buf * STAGE_ELEMS;
      const bf16_t* a_base = p.A + k0;
      asm volatile("" : "+s"(a_base));
#pragma unroll
      for (int i = 2 * H; i < (2 * H + 2 < NA ? 2 * H + 2 : NA); ++i) {   // NA = 4 requests per K-tile (3 for the 192-row tile: 2 + 1)
        unsigned o = a_off[i];
        asm volatile("" : "+v"(o));
        const bf16_t* ga = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(a_base) + o);
        bf16_t* la = sA + (i * NTHREADS + wave * 64) * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)la, 16, 0, 0);
      }
    }
  };
  auto stage_w_q = [&](int buf, int k0, auto half) {
    if constexpr (OFF32) {
      constexpr int H = decltype(half)::value;
      bf16_t* sW = smem + buf * STAGE_ELEMS + A_ELEMS;
      const bf16_t* w_base = p.W + k0;
      asm volatile("" : "+s"(w_base));
#pragma unroll
      for (int i = 2 * H; i < 2 * H + 2; ++i) {
        unsigned o = w_off[i];
        asm volatile("" : "+v"(o));
        const bf16_t* gw = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(w_base) + o);
        bf16_t* lw = sW + (i * NTHREADS + wave * 64) * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)gw, (lptr_t)lw, 16, 0, 0);
      }
    }
  };
  using Q0 = std::integral_constant<int, 0>;
  using Q1 = std::integral_constant<int, 1>;

  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fh = lane >> 4;
  const int nk = (p.K + BK - 1) / BK;

  f32x4 acc[TN][TM];  // [ni][mi]
#if defined(HAFF_TUNING) && defined(HAFF_EXP_MFMA32)
  f32x16 accw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) accw[i][j] = 0.f;
#endif
  bf16x8 wf[2][TN], af[2][TM];  // fragments of one K-tile: [32-deep k-step][16-row tile] (unused by the PP loop)
  constexpr int TMH = TM / 2;               // 16-row A tiles per half of the wave tile (4; 3 for the 192-row tile)
  bf16x8 pa[2][TMH > 0 ? TMH : 1], pwl[2][2], pwh[2][2];   // PP loop: one half of the A fragments, both halves of the W fragments
  auto read_frags = [&](int buf, int ks) {
    const bf16_t* sA = smem + buf * STAGE_ELEMS;
    const bf16_t* sW = sA + A_ELEMS;
    const int c = ks * 4 + fh;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int rw = wn * (BN / WN) + t * 16 + fr;
      wf[ks][t] = *reinterpret_cast<const bf16x8*>(sW + rw * BK + ((c ^ (rw & 7)) << 3));
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int ra = wm * (BM / WM) + t * 16 + fr;
      af[ks][t] = *reinterpret_cast<const bf16x8*>(sA + ra * BK + ((c ^ (ra & 7)) << 3));
    }
  };
  auto mfma_rows = [&](int ks, int mi_lo, int mi_hi) {
#pragma unroll
    for (int mi = mi_lo; mi < mi_hi; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][ni], af[ks][mi], acc[ni][mi], 0, 0, 0);
  };

  int buf0 = 0;           // LDS buffer that holds K-tile 0 of the current tile
  if constexpr (PP) {     // K-tile 0 of the first tile; later tiles get theirs from the K loop of the tile before
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (;;) {              // tiles of this workgroup (one pass unless the grid was capped: persistent 8-wave tile)
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  int m0e = m0, n0e = n0;                       // this tile's origin (the PP loop moves m0 / n0 on to the next tile)
  const int tile_next = tile + (int)gridDim.x;
  const bool has_next = PP && tile_next < nwg;
  HAFF_TRACE3(0);
  const bool pf_next = has_next && nk >= 2;   // the K loop requests the next tile's K-tile 0 (a single-K-tile product cannot: its
                                              // requests would have to go out before its own loop starts; see the loop's end)
  // K loop, software-pipelined ACROSS the workgroup barrier. One barrier per K-tile: after it every wave's share of
  // tile kt+1 has landed and every wave is done reading tile kt-1's buffer, so the DMA of tile kt+2 may overwrite it
  // and has a whole K-tile of MFMAs to land. The second half of k-step 1's MFMAs (operands already in registers) is
  // held back and issued AFTER the barrier, where it keeps the MFMA pipe busy while the first fragments of the
  // next K-tile come back from LDS. (Measured alternatives, tools/gemm_variant.py: two barriers per K-tile -7 %,
  // no hold-back -4 %, an LDS-counter split barrier and a ping-pong wave schedule no better than this.)
  if constexpr (PP) {
    // ---- ping-pong ring loop (round 3) ----
    // The two wave groups (wm = 0: waves 0-3, wm = 1: waves 4-7; waves w and w+4 share a SIMD) run the same program ONE
    // workgroup barrier apart: while one group multiplies a 64x32 quadrant of its 128x64 wave tile over the whole K-tile
    // (16 MFMAs, operands in registers), the other reads its next fragments from LDS and issues its share of ONE 16 KiB
    // quarter of a later K-tile. A K-tile is four phases:
    //   phase 0: read W lo (4), A lo (8);  DMA A rows   0-127 of K-tile kt+1;  quadrant (A lo, W lo)
    //   phase 1: read W hi (4);            DMA A rows 128-255 of K-tile kt+1;  quadrant (A lo, W hi)
    //   phase 2: read A hi (8);            DMA W rows   0-127 of K-tile kt+2;  quadrant (A hi, W hi)
    //   phase 3: (W lo is still held);     DMA W rows 128-255 of K-tile kt+2;  quadrant (A hi, W lo);  wait for K-tile kt+1
    // The request stream never drains: the one wait per K-tile is COUNTED (s_waitcnt vmcnt(4): the two W quarters of
    // K-tile kt+2 stay in flight across the barriers), every quarter has 5 to 12 barrier intervals (>= 1.3k cycles) to
    // land, and a wave issues two DMA instructions per phase instead of eight in a burst. Counted waits over LDS-DMA are
    // safe: the operations leave vmcnt in issue order (tools/probes/vmcnt_order_probe.hip); what bit round 1 was a
    // write-after-read race (tools/vmcnt_forensics.py). Hazards of this schedule, in barrier intervals
    // ("slots"; group g reads / issues in slot 8kt + 2q + g and multiplies in the next one):
    //   RAW  every wave waits for its share of K-tile kt+1 in phase 3 of K-tile kt, BEFORE the barrier that closes that
    //        slot; the first reads of K-tile kt+1 come at least one barrier later.
    //   WAR  a quarter is requested only after a barrier that follows the last read of its previous content, and every
    //        load section ends with an explicit s_waitcnt lgkmcnt(0) BEFORE its barrier (hipcc would otherwise sink the
    //        wait below the barrier): W quarters are last read in phase 1 (slots 8kt+2, +3) and requested in phases 2, 3
    //        (slots 8kt+4 ...), which is why W lo stays in registers for phase 3; A quarters of buffer cur^1 were last
    //        read in phase 2 of K-tile kt-1.
    // Tile boundary (persistent form): K-tile "nk" is K-tile 0 of the workgroup's next tile (staging coordinates move on
    // at phase 2 of K-tile nk-2), K-tile "nk+1" is not requested — its buffer holds the epilogue's staging images — and
    // its W quarters go out right after the epilogue, its A quarters in phases 0 and 1 as always.
    auto read_a = [&](int buf, auto half) {
      constexpr int H = decltype(half)::value;
      const bf16_t* sA = smem + buf * STAGE_ELEMS;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fh;
#pragma unroll
        for (int t = 0; t < TMH; ++t) {
          const int ra = wm * (BM / WM) + (H * TMH + t) * 16 + fr;
          pa[ks][t] = *reinterpret_cast<const bf16x8*>(sA + ra * BK + ((c ^ (ra & 7)) << 3));
        }
      }
    };
    auto read_w = [&](int buf, bf16x8 (&dst)[2][2], auto half) {
      constexpr int H = decltype(half)::value;
      const bf16_t* sW = smem + buf * STAGE_ELEMS + A_ELEMS;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fh;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int rw = wn * (BN / WN) + (H * 2 + t) * 16 + fr;
          dst[ks][t] = *reinterpret_cast<const bf16x8*>(sW + rw * BK + ((c ^ (rw & 7)) << 3));
        }
      }
    };
    // one 64x32 quadrant of the wave tile over the whole K-tile: 16 MFMAs, operands in registers
    auto quad = [&](const bf16x8 (&wq)[2][2], auto ni0, auto mi0) {
      constexpr int N0 = decltype(ni0)::value, M0 = decltype(mi0)::value;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < TMH; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[N0 + j][M0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ks][j], pa[ks][t], acc[N0 + j][M0 + t], 0, 0, 0);
    };
#if defined(HAFF_TUNING) && defined(HAFF_EXP_MFMA32)
    // timing experiment only (results are wrong): the same fragment reads feeding v_mfma_f32_32x32x16_bf16 — half as many
    // MFMA instructions of twice the length, i.e. the matrix pipe as busy as before while the SIMD's issue port is held
    // 8 of 32 cycles instead of 8 of 16
    auto quad32 = [&](const bf16x8 (&wq)[2][2], auto ni0, auto mi0) {
      constexpr int N0 = decltype(ni0)::value, M0 = decltype(mi0)::value;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            constexpr int dummy = 0;
            const int idx = (N0 / 2) * 4 + (M0 / 4) * 2 + j;
            accw[idx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][j], pa[ks][2 * j + tt], accw[idx], 0, 0, 0);
          }
    };
#define quad quad32
#endif
    auto slot_barrier = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto close_load = [&]() {   // my LDS reads are DONE before the barrier (WAR), then the barrier; nothing moves across
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slot_barrier();
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using IH = std::integral_constant<int, TMH>;
    // W quarters of K-tile 1 (the buffer was the previous tile's epilogue staging; the barrier behind that epilogue, or
    // the one behind the first tile's prologue, has passed)
    // the tile's 256 bias values ride along as ONE more DMA instruction (wave 0), a whole K loop ahead of the epilogue that
    // reads them from LDS: a global load issued there sat in front of pass 0 with its full latency exposed, once per tile
    // (round 4: a uniform base in SGPRs + the lane's 16-byte offset. With per-lane 64-bit addresses hipcc kept them alive across
    // the tile loop, spilled them, and the reload's s_waitcnt vmcnt(0) drained the previous tile's stores in front of these
    // requests)
    auto dma_row16 = [&](const void* base, float* lds_dst) {
      asm volatile("" : "+s"(base));
      unsigned o = (unsigned)lane * 16u;
      asm volatile("" : "+v"(o));
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(base) + o), (lptr_t)lds_dst, 16, 0, 0);
    };
    float* sXw = reinterpret_cast<float*>(smem + 2 * STAGE_ELEMS);
    if (has_bias && (SPEC || n0 + BN <= p.N) && al16(p.bias) && wave == 0) dma_row16(p.bias + n0, sXw);
    // folded norm: the tile's 256 {mean, rstd} rows and 256 column sums the same way (waves 1-3, three DMA instructions):
    // loaded at the top of the epilogue they put a memory round trip in front of pass 0 of every tile (+0.35 % of the step,
    // norm-folded products 1100 -> 1108 TFLOP/s: bench.py `by_epilogue`, profiles/r3_ln_prefetch_ab.txt)
    if (BM == 256 && has_ln && (MFULL || m0 + BM <= p.M) && (SPEC || n0 + BN <= p.N) && al16(p.ln_stats)) {   // (256 rows per request pair: the 192-row tile loads them in its epilogue)
      if (wave == 1) dma_row16(p.ln_stats + 2 * (long)m0, sXw + 256);
      if (wave == 2) dma_row16(p.ln_stats + 2 * (long)(m0 + 128), sXw + 512);
      if (wave == 3 && has_csum && al16(p.ln_colsum)) dma_row16(p.ln_colsum + n0, sXw + 768);
    }
    // round 4: the output row map of the tile's 256 rows the same way (wave 4). Read per lane from global memory at the top of the
    // epilogue (orow_l below) it put a dependent memory round trip in front of pass 0 of every windowed q|k|v tile.
    if (BM == 256 && has_map && (MFULL || m0 + BM <= p.M) && al16(p.row_map) && wave == 4) dma_row16(p.row_map + m0, sXw + 1024);
    if (nk > 1) {
      stage_w_q(buf0 ^ 1, BK, Q0{});
      stage_w_q(buf0 ^ 1, BK, Q1{});
    }
    HAFF_TRACE(1);
    HAFF_TRACE3(1);
    if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = (kt & 1) ^ buf0;
      const bool a_next = (kt + 1 < nk) || pf_next;
      const int a_k0 = (kt + 1 < nk) ? (kt + 1) * BK : 0;
      const bool w_next = (kt + 2 < nk) || (kt + 2 == nk && pf_next);
      const int w_k0 = (kt + 2 < nk) ? (kt + 2) * BK : 0;
#ifdef HAFF_EXP_NODMA     // timing experiments only (results are wrong): no operand requests / no fragment reads in the loop
#define PP_DMA(x) do {} while (0)
#else
#define PP_DMA(x) x
#endif
#ifdef HAFF_EXP_NOREAD
#define PP_READ(x) do { if (kt == 0) { x; } } while (0)
#else
#define PP_READ(x) x
#endif
      // ---- load slot A: W lo, W hi, A lo; requests for the A quarters of K-tile kt+1 ----
      if (kt == nk - 1) HAFF_TRACE3(2);
      PP_READ(read_w(cur, pwl, Q0{}));
      PP_READ(read_a(cur, Q0{}));
      PP_READ(read_w(cur, pwh, Q1{}));
      if (a_next) {
        PP_DMA(stage_a_q(cur ^ 1, a_k0, Q0{}));
        PP_DMA(stage_a_q(cur ^ 1, a_k0, Q1{}));
      }
      close_load();
      if (kt == nk - 1) HAFF_TRACE3(3);
      // ---- multiply slot A: quadrants (A lo, W lo), (A lo, W hi) ----
      __builtin_amdgcn_s_setprio(1);
      quad(pwl, I0{}, I0{});
      quad(pwh, I2{}, I0{});
      __builtin_amdgcn_s_setprio(0);
      slot_barrier();
      if (kt == nk - 1) HAFF_TRACE3(4);
      // ---- load slot B: A hi; requests for the W quarters of K-tile kt+2; the wait for K-tile kt+1 ----
      if (kt + 2 == nk && pf_next) {   // from here on the staging coordinates are the next tile's
        tile_origin(tile_next, m0, n0);
        stage_coords(m0, n0);
      }
      PP_READ(read_a(cur, Q1{}));
      if (kt == nk - 1) HAFF_TRACE3(5);
      if (w_next) {
        PP_DMA(stage_w_q(cur, w_k0, Q0{}));
        PP_DMA(stage_w_q(cur, w_k0, Q1{}));
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K-tile kt+1 landed (my share); W of K-tile kt+2 stays in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (kt == nk - 1) HAFF_TRACE3(6);
      close_load();
      if (kt == nk - 1) HAFF_TRACE3(7);
      // ---- multiply slot B: quadrants (A hi, W hi), (A hi, W lo) ----
      __builtin_amdgcn_s_setprio(1);
      quad(pwh, I2{}, IH{});
      quad(pwl, I0{}, IH{});
      __builtin_amdgcn_s_setprio(0);
      if (!(wm == 1 && kt == nk - 1)) slot_barrier();   // group 1 gives back the barrier it took at the top
    }
  } else {
    // 4-wave tile: two workgroups per CU cover for each other. One barrier per K-tile: tile kt+1's DMA is issued at the top
    // of iteration kt (its buffer was last read in iteration kt-1, which ended with the barrier), has the whole MFMA phase
    // to land, and is waited for in full before the barrier that publishes it. The explicit lgkmcnt(0) BEFORE that barrier
    // is what round 1's form lacked: hipcc had sunk the wait for the last two ds_read_b128 below its end-of-iteration
    // barrier, so a fast wave could request the next tile into a buffer a slow wave was still reading — a write-after-read
    // race that showed as rare wrong fragments beside a second stream (tools/vmcnt_forensics.py).
    if (p.deep_k) {
      // Launches whose workgroups are few and whose K loop is what they wait on (one-frame prefill / CLIP / SAM products,
      // split-K slices, decode batches of 33..64 rows): with ONE K-tile of DMA in flight an iteration lasts one memory round
      // trip (~1 us) whatever its 32 MFMAs take (0.2 us) — 288 x 12288 x 4096 streamed its weights at 1.5 TB/s. Here the
      // fragments of K-tile kt go to registers first, a barrier hands its buffer straight back to the DMA (K-tile kt + 2),
      // and the MFMAs run from registers with two K-tiles in flight behind a COUNTED wait (8 requests per wave per K-tile;
      // LDS-DMA leaves vmcnt in issue order: tools/probes/vmcnt_order_probe.hip). Two barriers per K-tile cost compute-bound
      // launches 7 % (measured in round 1), which is why the launcher sets this for the short ones only.
      constexpr int NS = NA + NW;   // DMA instructions per wave per K-tile
      static_assert(NS == 8, "counted wait below is written for 8 requests per K-tile");
      stage(0, 0);
      if (nk > 1) {
        stage(1, BK);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        read_frags(cur, 0);
        read_frags(cur, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of this buffer are done BEFORE the barrier (WAR)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) stage(cur, (kt + 2) * BK);
        mfma_rows(0, 0, TM);
        mfma_rows(1, 0, TM);
        if (kt + 1 < nk) {   // K-tile kt + 1 has landed (mine); kt + 2 may stay in flight
          if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);   // (issuing it after the fragment reads instead measured -5 %)
      if (kt == 0) HAFF_TRACE(1);
      read_frags(cur, 0);
      read_frags(cur, 1);
      mfma_rows(0, 0, TM);
      mfma_rows(1, 0, TM);
      __builtin_amdgcn_sched_group_barrier(0x100, TN + TM, 0);
#pragma unroll
      for (int g = 0; g < TM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (TN + TM + TM - 1) / TM, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, TN * TM, 0);
      if (kt + 1 < nk) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    }
  }
#if defined(HAFF_TUNING) && defined(HAFF_EXP_MFMA32)
#undef quad
  if constexpr (PP) {   // keep the experiment's accumulators alive: hand them to the epilogue (garbage in, garbage out)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[(i >> 2) * 2 + (j >> 1)][(i & 3) * 2 + (j & 1)] = f32x4{accw[i][4 * j], accw[i][4 * j + 1], accw[i][4 * j + 2], accw[i][4 * j + 3]};
  }
#endif
  // the epilogue reuses stage memory: every wave is done reading fragments (PP: the loop's last barrier says so)
  if constexpr (!PP) __builtin_amdgcn_s_barrier();
  HAFF_TRACE(2);
  HAFF_TRACE2(0);
  HAFF_TRACE3(9);
  // the buffer the last K-tile was read from takes the LDS-staged epilogue's images (ragged tiles); the other one holds the
  // next tile's first K-tile, which the ring loop has requested AND waited for already
  const int ebuf = ((nk - 1) & 1) ^ buf0;

  // ---- epilogue, staged through LDS so global traffic is whole 128-B row segments ----
  // lane holds D[n = 4*fh + reg][m = fr] of each 16x16 tile -> (+bias, act) -> fp32 LDS image [16 rows][WCOLS],
  // wave-private (LDS ops of one wave execute in order: no workgroup barrier inside the epilogue) -> read back
  // 8 consecutive columns per lane -> (+residual) -> 16-B stores.
  // Every global LOAD of the epilogue (bias, row map, residual) is issued before the stores it would otherwise
  // queue behind: vmcnt retires in order, so a load issued after a store waits for that store's round trip.
  constexpr int WCOLS = SWIGLU ? WNC / 2 : WNC;   // output columns owned by a wave
  constexpr int RS = WCOLS + 4;             // LDS row stride in floats (pad keeps b128 accesses conflict-free)
  constexpr int LPR = WCOLS / 8;            // lanes per output row on the read-back side
  constexpr int RPS = 64 / LPR;             // rows per read-back step
  constexpr int STEPS = 16 / RPS;
  constexpr int WROWS = BM / WM;            // output rows owned by a wave (64 or 128)
  static_assert(WM * WN * 16 * RS * 4 <= STAGE_ELEMS * 2, "epilogue staging must fit in one LDS stage");
  float* sEp = reinterpret_cast<float*>(smem + ebuf * STAGE_ELEMS) + wave * (16 * RS);
  const int n_total_out = SWIGLU ? (p.N >> 1) : p.N;
  const int n_wave_in = n0e + wn * WNC;                         // first (interleaved) input column of the wave
  const int n_wave_out = SWIGLU ? (n_wave_in >> 1) : n_wave_in;
  const int m_wave = m0e + wm * WROWS;
  const bool c_vec = ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) && ((p.ldc & 7) == 0);
  const bool r_vec = has_res && ((reinterpret_cast<uintptr_t>(p.resid) & 15) == 0) && ((p.ldr & 7) == 0);
  const bool fast = SPEC || (c_vec && (!has_res || r_vec) && (n_wave_out + WCOLS <= n_total_out));
  const bool nt_out = p.nt_out != 0;

  float bias_r[TN][4];
  if (!has_bias) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) bias_r[ni][r] = 0.f;
  } else if (PP && (SPEC || n0e + BN <= p.N) && al16(p.bias)) {
    const float* sBias = reinterpret_cast<const float*>(smem + 2 * STAGE_ELEMS) + wn * WNC;   // staged at the top of the tile
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) load4(sBias + ni * 16 + fh * 4, bias_r[ni]);
  } else if (n_wave_in + WNC <= p.N && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) load4(p.bias + n_wave_in + ni * 16 + fh * 4, bias_r[ni]);
  } else {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_wave_in + ni * 16 + fh * 4 + r;
        bias_r[ni][r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
      }
  }
  HAFF_TRACE3(28);
  // folded norm: this wave's {mean, rstd} rows and column sums go through LDS (free stage memory past the staging
  // images) so the per-pass code reads them back instead of holding 2*TM + 4*TN more registers
  static_assert((WM * WN * 16 * RS + WM * WN * (2 * WROWS + WNC)) * 4 <= STAGE_ELEMS * 2, "epilogue LDS fits one stage");
  float* sStat = reinterpret_cast<float*>(smem + ebuf * STAGE_ELEMS) + WM * WN * 16 * RS + wave * (2 * WROWS + WNC);   // behind the staging images
  float* sCsum = sStat + 2 * WROWS;
  // (8-wave tile, interior tile: both came in by DMA at the top of the tile's K loop — see there)
  const bool ln_pref = PP && BM == 256 && has_ln && (MFULL || m0e + BM <= p.M) && (SPEC || n0e + BN <= p.N) && al16(p.ln_stats) &&
                       (!has_csum || al16(p.ln_colsum));
  if (ln_pref) {
    float* sLn = reinterpret_cast<float*>(smem + 2 * STAGE_ELEMS) + 256;
    sStat = sLn + 2 * wm * WROWS;
    if (has_csum) sCsum = sLn + 512 + wn * WNC;
    else {   // RMSNorm: no mean term — zeros from the wave's own scratch
#pragma unroll
      for (int h = 0; h < WNC / 64; ++h) sCsum[h * 64 + lane] = 0.f;
      __builtin_amdgcn_wave_barrier();
    }
  } else if (has_ln) {
#pragma unroll
    for (int h = 0; h < (WROWS + 63) / 64; ++h) {
      const int m = min(m_wave + h * 64 + lane, p.M - 1);
      const float2 st = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (long)m);
      if (h * 64 + lane < WROWS) *reinterpret_cast<float2*>(sStat + 2 * (h * 64 + lane)) = st;
    }
#pragma unroll
    for (int h = 0; h < WNC / 64; ++h) {
      const int n = n_wave_in + h * 64 + lane;
      sCsum[h * 64 + lane] = (has_csum && n < p.N) ? p.ln_colsum[n] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
  }
  HAFF_TRACE3(29);
  // output row of wave-row (lane) and (lane + 64): -1 = dropped. Distributed to the read-back lanes by ds_bpermute.
  int orow_l[(WROWS + 63) / 64];
  const bool map_staged = PP && BM == 256 && has_map && (MFULL || m0e + BM <= p.M) && al16(p.row_map);   // (the DMA above)
#pragma unroll
  for (int h = 0; h < (WROWS + 63) / 64; ++h) {
    const int m = m_wave + h * 64 + lane;
    if (map_staged) orow_l[h] = reinterpret_cast<const int*>(smem + 2 * STAGE_ELEMS)[1024 + wm * WROWS + h * 64 + lane];
    else orow_l[h] = ((MFULL || m < p.M) && h * 64 + lane < WROWS) ? (has_map ? p.row_map[m] : m) : -1;
  }
  HAFF_TRACE3(30);
#ifndef HAFF_EPI_LDS   // -DHAFF_EPI_LDS: every tile through the LDS-staged epilogue below (A/B runs)
  // ---- register epilogue (interior tiles, 16-B aligned rows) ----
  // The swapped MFMA orientation leaves 4 consecutive output columns of ONE row in each lane (chunk c = 16-column
  // group, columns 16c + 4fh .. +3). Two neighbouring chunks are made into 8 consecutive columns per lane by ONE
  // v_permlane16_swap per register (lanes 16 apart trade halves: lane fh even keeps chunk 2j and receives the next four
  // columns of it from lane fh+1, which receives chunk 2j+1's previous four in return), so a lane stores 16 B and the
  // four lanes of a row cover 64 contiguous bytes of it: no LDS round trip, no wave barriers, no dependent
  // write -> read -> store chain per pass. (Measured on the LDS-staged form: its stores cost 8 % of a K = 1280 launch, the
  // staging around them 20 %: tools/gemm_variant.py nostore / noepi.)
#ifdef HAFF_EXP_NOEPI   // timing experiment: the accumulators are kept alive, nothing is computed or written
  if (fast) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) asm volatile("" ::"v"(acc[ni][mi]));
  } else
#endif
  if (fast) {
    // ALL = every row of the wave tile is written (no row map, not the ragged last M-tile): the loads and stores below are
    // then unconditional, which lets hipcc wait for a prefetched residual with a COUNTED vmcnt (behind an exec-masked
    // store it falls back to vmcnt(0): every pass would drain the previous pass's stores first).
    // RES = a bf16 residual is added: a compile-time property of the instance so that the plain epilogue carries no wait at
    // all (a register copy of a "maybe loaded" value made hipcc drain vmcnt — every store of the previous pass — once per
    // pass, in the LDS-staged epilogue of rounds 1-2 as well) and the residual one waits with a counted vmcnt.
    auto reg_epilogue = [&](auto all_tag, auto res_tag) {
    constexpr bool ALL = decltype(all_tag)::value;
    constexpr bool RES = decltype(res_tag)::value && !OUT_F32;
    constexpr bool R32 = RES && ALL && SPEC && (FLAGS & GF_RES32) != 0;   // fp32 residual stream in / out + bf16 copy
    constexpr int NCH = SWIGLU ? TN / 2 : TN;   // 4-column chunks per lane per pass
    const int coff = 16 * (fh & 1) + 4 * (fh & 2);   // first of this lane's 8 columns inside a chunk pair (bf16 output)
    auto out_row = [&](int mi) -> int {
      if constexpr (ALL) return m_wave + mi * 16 + fr;
      else return __shfl(orow_l[(mi * 16) >> 6], (mi * 16 + fr) & 63);
    };
    // bf16 residual of pass mi + 1, requested before the stores of pass mi go out
    uint4 rnext[NCH / 2 > 0 ? NCH / 2 : 1];
    haff_f32x4 rnext32[R32 ? NCH / 2 : 1][2];
    // (R32) the lane's 8 fp32 stream columns of pass mi sit at x_lane + mi * x_pass + 128 * j bytes
    char* x_lane = nullptr;
    long x_pass = 0;
    if constexpr (R32) {
      x_lane = reinterpret_cast<char*>(p.res32) + ((long)(m_wave + fr) * p.ld32 + n_wave_out + coff) * 4;
      x_pass = 16L * p.ld32 * 4;
    }
    auto fetch_r = [&](int mi) {
      if constexpr (R32) {
#pragma unroll
        for (int j = 0; j < NCH / 2; ++j) {
          const haff_f32x4* src = reinterpret_cast<const haff_f32x4*>(x_lane + mi * x_pass + 128 * j);
          rnext32[j][0] = src[0];
          rnext32[j][1] = src[1];
        }
        return;
      }
      const int orow = out_row(mi);
#pragma unroll
      for (int j = 0; j < NCH / 2; ++j) {
        rnext[j] = uint4{0u, 0u, 0u, 0u};
        if (ALL || orow >= 0)
          rnext[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.resid) + (long)orow * p.ldr + n_wave_out +
                                                     32 * j + coff);
      }
    };
    if constexpr (RES) fetch_r(0);
    // head-major scatter (see GemmArgs::hm_d): the lane's 8 columns of chunk pair j lie inside ONE head (hm_d % 8 == 0); their
    // element offset is fixed for the tile, the row part (row_map[m] * hm_d) changes per pass
    long hm_col[NCH / 2 > 0 ? NCH / 2 : 1];
    if constexpr (!ALL && !RES && !SWIGLU && !OUT_F32) {
      if (has_hm) {
#pragma unroll
        for (int j = 0; j < NCH / 2; ++j) {
          const int col = n_wave_out + 32 * j + coff;
          const int part = (col >= p.hm_hd) + (col >= 2 * p.hm_hd);
          const int rem = col - part * p.hm_hd;
          const int h = (int)(((float)rem + 0.5f) * (1.0f / (float)p.hm_d));   // exact: rem < 2^16, hm_d >= 8
          hm_col[j] = (long)part * p.hm_part + (long)h * p.hm_head + (rem - h * p.hm_d);
        }
      }
    }
    HAFF_TRACE2(1);
    HAFF_TRACE3(10);
    // ALL: the lane's output address is affine in the pass index — one 64-bit base per tile and a scalar stride per pass
    // instead of a 64-bit multiply-add chain per store (the epilogue is instruction-bound: two waves per SIMD, ~60 VALU
    // per pass before this)
    const char* c_lane = reinterpret_cast<const char*>(p.C) +
                         ((long)(m_wave + fr) * p.ldc + n_wave_out + (OUT_F32 ? 4 * fh : coff)) * (OUT_F32 ? 4 : 2);
    const long c_pass = 16L * p.ldc * (OUT_F32 ? 4 : 2);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      // folded norm: x = rstd * (acc - mean * colsum) + bias as TWO explicit FMAs (acc - mean * colsum here, * rstd + bias where the
      // bias is added), so that every instance of this kernel rounds alike whatever hipcc's contraction pass would pick; without a
      // norm rstd_m = 1 and fma(acc, 1, bias) is the plain sum
      float rstd_m = 1.0f;
      if (has_ln) {
        const float2 st = *reinterpret_cast<const float2*>(sStat + 2 * (mi * 16 + fr));
        rstd_m = st.y;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          float cs[4];
          load4(sCsum + ni * 16 + fh * 4, cs);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[ni][mi][r] = __builtin_fmaf(-st.x, cs[r], acc[ni][mi][r]);
        }
      }
      if (mi == 0) HAFF_TRACE3(21);
      if (mi == 4) HAFF_TRACE3(25);
      float val[NCH][4];
      if constexpr (!SWIGLU) {
        auto act_side = [&](auto tag) {   // the activation is resolved ONCE per pass (wave-uniform switch)
          constexpr int ACT = decltype(tag)::value;
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) {
            if constexpr (ACT == HAFF_ACT_GELU) {
              const haff_f2 a = gelu_pair(haff_f2{__builtin_fmaf(acc[ni][mi][0], rstd_m, bias_r[ni][0]), __builtin_fmaf(acc[ni][mi][1], rstd_m, bias_r[ni][1])});
              const haff_f2 b = gelu_pair(haff_f2{__builtin_fmaf(acc[ni][mi][2], rstd_m, bias_r[ni][2]), __builtin_fmaf(acc[ni][mi][3], rstd_m, bias_r[ni][3])});
              val[ni][0] = a[0]; val[ni][1] = a[1]; val[ni][2] = b[0]; val[ni][3] = b[1];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) val[ni][r] = gemm_act<ACT>(__builtin_fmaf(acc[ni][mi][r], rstd_m, bias_r[ni][r]));
            }
          }
        };
        switch (act) {
          case HAFF_ACT_GELU: act_side(std::integral_constant<int, HAFF_ACT_GELU>{}); break;
          case HAFF_ACT_QUICK_GELU: act_side(std::integral_constant<int, HAFF_ACT_QUICK_GELU>{}); break;
          case HAFF_ACT_RELU: act_side(std::integral_constant<int, HAFF_ACT_RELU>{}); break;
          case HAFF_ACT_SILU: act_side(std::integral_constant<int, HAFF_ACT_SILU>{}); break;
          default: act_side(std::integral_constant<int, HAFF_ACT_NONE>{}); break;
        }
      } else {
#pragma unroll
        for (int nj = 0; nj < TN / 2; ++nj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g = __builtin_fmaf(acc[2 * nj][mi][r], rstd_m, bias_r[2 * nj][r]);
            const float u = __builtin_fmaf(acc[2 * nj + 1][mi][r], rstd_m, bias_r[2 * nj + 1][r]);
            val[nj][r] = g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)) * u;
          }
      }
      if (mi == 0) HAFF_TRACE3(22);
      if (mi == 4) HAFF_TRACE3(26);
      const int orow = out_row(mi);
      float st1 = 0.f, st2 = 0.f;   // row statistics of the final values (RES && ALL && p.stat_out)
      // RoPE + KV-cache append (see GemmArgs::rope_cs): rotate the lane's (c, c + 64) pairs, pick the destination row
      bf16_t* rope_dst = nullptr;
      if constexpr (!RES && !SWIGLU && !OUT_F32) {
        if (has_rope) {   // wave-uniform
          const int role = n0e / p.rope_hd;                               // 0 q, 1 k, 2 v: uniform over a 256-column tile
          const int col0 = n0e - role * p.rope_hd + (wn >> 1) * 128 + (wn & 1) * 32 + coff;   // lane's first column, j = 0
          const int m = (ALL || orow >= 0) ? orow : 0;
          const int b = (int)(((float)m + 0.5f) * (1.0f / (float)p.rope_t));   // exact for m < 2^22 (checked on the host)
          const int pos = p.rope_pos0 + (m - b * p.rope_t);
          if (role < 2) {
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
              float c4[4], s4[4];
              const float* csr = p.rope_cs + (long)pos * 128 + (wn & 1) * 32 + tb * 16 + 4 * fh;
              load4(csr, c4);
              load4(csr + 64, s4);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float x1 = val[tb][r], x2 = val[tb + 2][r];
                val[tb][r] = x1 * c4[r] - x2 * s4[r];
                val[tb + 2][r] = x2 * c4[r] + x1 * s4[r];
              }
            }
          }
          rope_dst = role == 0 ? reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + col0
                               : reinterpret_cast<bf16_t*>(role == 1 ? p.rope_k : p.rope_v) +
                                     ((long)b * p.rope_tmax + pos) * p.rope_hd + col0;
        }
      }
      if constexpr (OUT_F32) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (ALL || orow >= 0) {
            const long o = n_wave_out + 16 * c + 4 * fh;
            if (has_res) {
              float rr[4];
              load4(reinterpret_cast<const float*>(p.resid) + (long)orow * p.ldr + o, rr);
#pragma unroll
              for (int r = 0; r < 4; ++r) val[c][r] += rr[r];
            }
            haff_f32x4 q = {val[c][0], val[c][1], val[c][2], val[c][3]};
            float* dst = ALL ? reinterpret_cast<float*>(const_cast<char*>(c_lane) + mi * c_pass) + 16 * c
                             : reinterpret_cast<float*>(p.C) + (long)orow * p.ldc + o;
            *reinterpret_cast<haff_f32x4*>(dst) = q;
          }
        }
      } else {
        uint4 rcur[NCH / 2 > 0 ? NCH / 2 : 1];
        haff_f32x4 rcur32[R32 ? NCH / 2 : 1][2];
        if constexpr (RES) {
#pragma unroll
          for (int j = 0; j < NCH / 2; ++j) {
            if constexpr (R32) { rcur32[j][0] = rnext32[j][0]; rcur32[j][1] = rnext32[j][1]; }
            else rcur[j] = rnext[j];
          }
          if (mi + 1 < TM) fetch_r(mi + 1);
        }
#pragma unroll
        for (int j = 0; j < NCH / 2; ++j) {
          haff_u32x4 q;
          if constexpr (RES) {   // the residual is added in fp32: trade fp32 registers, add, round once
            float v8[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              unsigned tx = __builtin_bit_cast(unsigned, val[2 * j][r]), ty = __builtin_bit_cast(unsigned, val[2 * j + 1][r]);
              permlane16_swap(tx, ty);
              v8[r] = __builtin_bit_cast(float, tx);
              v8[4 + r] = __builtin_bit_cast(float, ty);
            }
            if constexpr (R32) {   // fp32 stream: add, write the fp32 sums back in place; the bf16 copy leaves below
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v8[e] += rcur32[j][0][e];
                v8[4 + e] += rcur32[j][1][e];
              }
              haff_f32x4* xd = reinterpret_cast<haff_f32x4*>(x_lane + mi * x_pass + 128 * j);
              xd[0] = haff_f32x4{v8[0], v8[1], v8[2], v8[3]};
              xd[1] = haff_f32x4{v8[4], v8[5], v8[6], v8[7]};
            } else {
            const unsigned w[4] = {rcur[j].x, rcur[j].y, rcur[j].z, rcur[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v8[2 * e] += __builtin_bit_cast(float, w[e] << 16);
              v8[2 * e + 1] += __builtin_bit_cast(float, w[e] & 0xffff0000u);
            }
            }
            q = haff_u32x4{pack_bf16x2(v8[0], v8[1]), pack_bf16x2(v8[2], v8[3]), pack_bf16x2(v8[4], v8[5]), pack_bf16x2(v8[6], v8[7])};
            if constexpr (ALL) {
              if (has_stat) {   // (wave-uniform) the lane's 8 columns of this row; the row's other columns sit in 3 more lanes
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  st1 += v8[e];
                  st2 = __builtin_fmaf(v8[e], v8[e], st2);
                }
              }
            }
          } else {
            unsigned x0 = pack_bf16x2(val[2 * j][0], val[2 * j][1]), y0 = pack_bf16x2(val[2 * j + 1][0], val[2 * j + 1][1]);
            unsigned x1 = pack_bf16x2(val[2 * j][2], val[2 * j][3]), y1 = pack_bf16x2(val[2 * j + 1][2], val[2 * j + 1][3]);
            permlane16_swap(x0, y0);
            permlane16_swap(x1, y1);
            q = haff_u32x4{x0, x1, y0, y1};
          }
#ifdef HAFF_EXP_NOSTORE   // timing experiment: everything but the global store
          asm volatile("" ::"v"(q));
#else
          if (ALL || orow >= 0) {
            bf16_t* dst = ALL ? reinterpret_cast<bf16_t*>(const_cast<char*>(c_lane) + mi * c_pass) + 32 * j
                              : reinterpret_cast<bf16_t*>(p.C) + (long)orow * p.ldc + n_wave_out + 32 * j + coff;
            if constexpr (!RES && !SWIGLU) {
              if (rope_dst) dst = rope_dst + 64 * j;   // logical columns: block pair j = 1 is the rotate-half partner half
            }
            if constexpr (!ALL && !RES && !SWIGLU) {
              if (has_hm) dst = reinterpret_cast<bf16_t*>(p.C) + (long)orow * p.hm_d + hm_col[j];
            }
            // plain stores: a lane writes HALF a 128-B line here and the other half with its next store; the L2 merges
            // them, a non-temporal store would send each half to memory on its own (measured -2...-8 %)
            *reinterpret_cast<haff_u32x4*>(dst) = q;
          }
#endif
        }
      }
      if (mi == 0) HAFF_TRACE3(23);
      if (mi == 4) HAFF_TRACE3(27);
      if constexpr (RES && ALL) {
        if (has_stat) {   // the four lanes of a row (same fr, fh = 0..3) -> one {sum, sum of squares} per (row, wave)
          st1 = quad_row_sum(st1);
          st2 = quad_row_sum(st2);
          if (fh == 0)
            *reinterpret_cast<float2*>(p.stat_out + ((long)orow * p.stat_slots + (n_wave_out >> 6)) * 2) = float2{st1, st2};
        }
      }
      HAFF_TRACE2(2 + mi);
      HAFF_TRACE3(11 + mi);
    }
    };   // reg_epilogue
    const bool all_rows = !has_map && (MFULL || m_wave + WROWS <= p.M);
    if (!OUT_F32 && has_res) {
      if (all_rows) reg_epilogue(std::true_type{}, std::true_type{});
      else reg_epilogue(std::false_type{}, std::true_type{});
    } else {
      if (all_rows) reg_epilogue(std::true_type{}, std::false_type{});
      else reg_epilogue(std::false_type{}, std::false_type{});
    }
  } else
#endif
  {
  const int rb_r = lane / LPR, rb_c = (lane % LPR) * 8;   // read-back row within a step / first column
  const long n_out = n_wave_out + rb_c;

  // residual of pass mi, prefetched one pass ahead (bf16 output only; the rare f32-out residual loads in place)
  constexpr bool PREFETCH_R = !OUT_F32;
  uint4 rres[STEPS];
#pragma unroll
  for (int st = 0; st < STEPS; ++st) rres[st] = uint4{0u, 0u, 0u, 0u};
  auto fetch_resid = [&](int mi) {
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int wr = mi * 16 + st * RPS + rb_r;
      const int orow = __shfl(orow_l[(mi * 16) >> 6], wr & 63);
      rres[st] = uint4{0u, 0u, 0u, 0u};
      if (orow >= 0)   // (non-temporal residual loads measured neutral)
        rres[st] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.resid) + (long)orow * p.ldr + n_out);
    }
  };
  const bool pre_r = PREFETCH_R && fast && has_res;
  if (pre_r) fetch_resid(0);

#ifdef HAFF_EXP_NOEPI   // timing experiment: the accumulators are kept alive, nothing is written
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) asm volatile("" ::"v"(acc[ni][mi]));
#else
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    float* row = sEp + fr * RS + fh * 4;
    float rstd_m = 1.0f;   // (see the register epilogue: the same two FMAs)
    if (has_ln) {
      const float2 st = *reinterpret_cast<const float2*>(sStat + 2 * (mi * 16 + fr));
      rstd_m = st.y;
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        float cs[4];
        load4(sCsum + ni * 16 + fh * 4, cs);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ni][mi][r] = __builtin_fmaf(-st.x, cs[r], acc[ni][mi][r]);
      }
    }
    if (!SWIGLU) {
      // the activation is resolved ONCE per pass (wave-uniform switch) so the per-element code is straight-line
      auto write_side = [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          float v[4];
#ifndef HAFF_GEMM_GELU_SCALAR
          if constexpr (ACT == HAFF_ACT_GELU) {
            const haff_f2 a = gelu_pair(haff_f2{__builtin_fmaf(acc[ni][mi][0], rstd_m, bias_r[ni][0]), __builtin_fmaf(acc[ni][mi][1], rstd_m, bias_r[ni][1])});
            const haff_f2 b = gelu_pair(haff_f2{__builtin_fmaf(acc[ni][mi][2], rstd_m, bias_r[ni][2]), __builtin_fmaf(acc[ni][mi][3], rstd_m, bias_r[ni][3])});
            v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
          } else
#endif
          {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gemm_act<ACT>(__builtin_fmaf(acc[ni][mi][r], rstd_m, bias_r[ni][r]));
          }
          store4(row + ni * 16, v);
        }
      };
      switch (act) {
        case HAFF_ACT_GELU: write_side(std::integral_constant<int, HAFF_ACT_GELU>{}); break;
        case HAFF_ACT_QUICK_GELU: write_side(std::integral_constant<int, HAFF_ACT_QUICK_GELU>{}); break;
        case HAFF_ACT_RELU: write_side(std::integral_constant<int, HAFF_ACT_RELU>{}); break;
        case HAFF_ACT_SILU: write_side(std::integral_constant<int, HAFF_ACT_SILU>{}); break;
        default: write_side(std::integral_constant<int, HAFF_ACT_NONE>{}); break;
      }
    } else {
#pragma unroll
      for (int nj = 0; nj < TN / 2; ++nj) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = __builtin_fmaf(acc[2 * nj][mi][r], rstd_m, bias_r[2 * nj][r]);
          const float u = __builtin_fmaf(acc[2 * nj + 1][mi][r], rstd_m, bias_r[2 * nj + 1][r]);
          v[r] = g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)) * u;
        }
        store4(row + nj * 16, v);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (fast) {
      // interior tile, 16-B aligned rows: straight-line read-back, 16-B (bf16) / 2x16-B (f32) stores per lane
      uint4 rcur[STEPS];
#pragma unroll
      for (int st = 0; st < STEPS; ++st) rcur[st] = rres[st];
      if (pre_r && mi + 1 < TM) fetch_resid(mi + 1);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int wr = mi * 16 + st * RPS + rb_r;
        const int orow = __shfl(orow_l[(mi * 16) >> 6], wr & 63);
        float v[8];
        load8(sEp + (st * RPS + rb_r) * RS + rb_c, v);
        if (orow >= 0) {
          if (OUT_F32) {
            if (p.resid) {
              float rr[8];
              load8(reinterpret_cast<const float*>(p.resid) + (long)orow * p.ldr + n_out, rr);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += rr[j];
            }
            store8_c(reinterpret_cast<float*>(p.C) + (long)orow * p.ldc + n_out, v, nt_out);
          } else {
            if (p.resid) {
              const unsigned int w[4] = {rcur[st].x, rcur[st].y, rcur[st].z, rcur[st].w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                v[2 * j] += __builtin_bit_cast(float, w[j] << 16);
                v[2 * j + 1] += __builtin_bit_cast(float, w[j] & 0xffff0000u);
              }
            }
#ifdef HAFF_EXP_NOSTORE   // timing experiment: everything but the global store
            asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
#else
            store8_c(reinterpret_cast<bf16_t*>(p.C) + (long)orow * p.ldc + n_out, v, nt_out);
#endif
          }
        }
      }
    } else {
      // ragged N edge or unaligned rows: element-wise, kept rolled (cold)
#pragma unroll 1
      for (int st = 0; st < STEPS; ++st) {
        const int wr = mi * 16 + st * RPS + rb_r;
        const int orow = __shfl(orow_l[(mi * 16) >> 6], wr & 63);
        if (orow < 0) continue;
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
          const long n = n_out + j;
          if (n >= n_total_out) break;
          float x = sEp[(st * RPS + rb_r) * RS + rb_c + j];
          if (OUT_F32) {
            if (p.resid) x += reinterpret_cast<const float*>(p.resid)[(long)orow * p.ldr + n];
            reinterpret_cast<float*>(p.C)[(long)orow * p.ldc + n] = x;
          } else {
            if (p.resid) x += bf16_to_f32(reinterpret_cast<const bf16_t*>(p.resid)[(long)orow * p.ldr + n]);
            reinterpret_cast<bf16_t*>(p.C)[(long)orow * p.ldc + n] = f32_to_bf16(x);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#endif
  }   // LDS-staged epilogue
  HAFF_TRACE(3);
#ifdef HAFF_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HAFF_TRACE(4);
#endif
  if (!has_next) break;
  HAFF_TRACE2(10);
  HAFF_TRACE3(19);
  if constexpr (PP) {
    __builtin_amdgcn_s_barrier();   // every wave is past its epilogue: its staging buffer takes K-tile 1
    if (!pf_next) {                 // single-K-tile products: the next tile's only K-tile is requested and awaited here
      tile_origin(tile_next, m0, n0);
      stage_coords(m0, n0);
      stage(ebuf ^ 1, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  HAFF_TRACE2(11);
  HAFF_TRACE3(20);
  HAFF_TRACE3_FLUSH();
  tile = tile_next;
  buf0 = ebuf ^ 1;
  }   // tile loop
}

// ---------------------------------------------------------------------------------------------------
// Skinny GEMM for M <= 64 (KV-cached decode steps, the [SEG] MLP, decoder token projections): the product is
// a stream over W — HBM-bound, 2*N*K bytes — so the grid is laid out for bandwidth, not for MFMA reuse:
// one workgroup per 16 weight rows (32 for SwiGLU pairs), its 4 waves split K into quarters, each wave streams its
// 16 x K/4 slab straight from HBM into MFMA A-fragments (16 B per lane), the activation rows (MT tiles of 16,
// L2-resident) are the B operand, and the four partial tiles meet in LDS. N = 4096 still gives 256 workgroups x
// 4 waves; the 128x128 tile launched 32 workgroups there (1.3 TB/s).
constexpr int skinny_batch(int nt, int mt) {   // k-steps per batch of loads: two register sets of W and x + accumulators
  const int u = (232 - 4 * nt * mt) / (8 * (nt + mt));
  return u >= 8 ? 8 : (u >= 4 ? 4 : 2);
}

template <int MT, int NT, bool SWIGLU, int KW = 4>
__global__ __launch_bounds__(64 * KW) void gemm_skinny_kernel(GemmArgs p) {
  static_assert(!SWIGLU || (NT % 2) == 0, "SwiGLU pairs a gate tile with an up tile");
  constexpr int U = skinny_batch(NT, MT);
  __shared__ float red[KW][MT][64][4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int fr = lane & 15, fh = lane >> 4;
  const int n0 = blockIdx.x * 16 * NT;
  const int KS = p.ksplit > 1 ? p.ksplit : 1;
  const int kq = p.K / (KW * KS);              // K % (32 * KW * KS) == 0: every wave gets whole 32-deep k-steps
  const int k_lo = (blockIdx.y * KW + wave) * kq;
  const bf16_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = min(mt * 16 + fr, p.M - 1);
    xrow[mt] = p.A + (long)(p.a_map ? p.a_map[m] : m) * p.lda + k_lo + fh * 8;
  }
  const bf16_t* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wrow[t] = p.W + (long)min(n0 + t * 16 + fr, p.N - 1) * p.ldw + k_lo + fh * 8;
  // folded RMSNorm from producer partials: thread (bl = tid >> 4, m = tid & 15) gathers partials b = bl, bl + 16, ...;
  // the loads go out before the weight stream and are only consumed after the K loop
  __shared__ float s_ssq[KW][16];
  // thread t fetches partial workgroups b = t and t + 256 of each row (<= 8 rows, <= 512 partials: haff_gemm_bf16_rms checks;
  // round 5: 8 rows instead of 4, as 16-byte loads of four rows each — configs[4] decodes 8 frames per GPU);
  // the values stay untouched in registers until after the K loop, so the weight stream starts without waiting for them
  // (summing them here — a dependent L2 round trip before the first weight load — made the step slower than the norm kernels)
  constexpr int SSQ_M = 8;
  float ssq_a[SSQ_M], ssq_b[SSQ_M];
  if (MT == 1 && p.ssq_in) {
    const int b0 = threadIdx.x, b1 = threadIdx.x + 64 * KW;
#pragma unroll
    for (int h4 = 0; h4 < SSQ_M / 4; ++h4) {
      float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
      if (4 * h4 < p.M) {   // (uniform) rows 4 h4 .. 4 h4 + 3 of a partial are 16 contiguous, 16-byte aligned bytes; rows >= M hold finite junk nobody reads
        if (b0 < p.ssq_n) load4(p.ssq_in + b0 * 16 + 4 * h4, va);
        if (b1 < p.ssq_n) load4(p.ssq_in + b1 * 16 + 4 * h4, vb);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { ssq_a[4 * h4 + e] = va[e]; ssq_b[4 * h4 + e] = vb[e]; }
    }
  }

  f32x4 acc[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Weight slab and activation fragments are streamed through two register sets each: the loads of batch b+1 (HBM for
  // W, L2 for x) are in flight while batch b is multiplied.
  uint4 wv[2][NT][U], xv[2][MT][U];
  auto load = [&](int set, int k) {
    // a 32-deep k-step takes 64 B of a row: the two k-steps that share a 128-B line are requested back to back, so
    // the second request meets the first in L1 instead of going to L2 again (PMC: L2 requests were 2x the bytes)
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      const int k0 = min(k + 32 * u, kq - 32), k1 = min(k + 32 * u + 32, kq - 32);   // tail: re-read, masked at the MFMA
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        wv[set][t][u] = *reinterpret_cast<const uint4*>(wrow[t] + k0);
        wv[set][t][u + 1] = *reinterpret_cast<const uint4*>(wrow[t] + k1);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        xv[set][mt][u] = *reinterpret_cast<const uint4*>(xrow[mt] + k0);
        xv[set][mt][u + 1] = *reinterpret_cast<const uint4*>(xrow[mt] + k1);
      }
    }
  };
  auto compute = [&](int set, int k) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k + 32 * u < kq) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[set][t][u]),
                                                                 __builtin_bit_cast(bf16x8, xv[set][mt][u]), acc[t][mt], 0, 0, 0);
      }
    }
  };
  constexpr int KB = 32 * U;
  load(0, 0);
  for (int k = 0; k < kq; k += 2 * KB) {
    if (k + KB < kq) load(1, k + KB);
    compute(0, k);
    if (k + KB < kq) {
      if (k + 2 * KB < kq) load(0, k + 2 * KB);
      compute(1, k + KB);
    }
  }
  if (MT == 1 && p.ssq_in) {   // lanes by butterfly, then the KW waves in index order (after the barrier below)
#pragma unroll
    for (int mm = 0; mm < SSQ_M; ++mm)
      if (mm < p.M) {
        float v = ssq_a[mm] + ssq_b[mm];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_ssq[wave][mm] = v;
      }
  }
  // the four K-quarters meet in LDS, one weight tile at a time; wave w then owns activation tile w of every weight
  // tile (lane holds D[n = 4*fh + r][m = 16*w + fr])
  float o[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float v[4] = {acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]};
      store4(&red[wave][mt][lane][0], v);
    }
    __syncthreads();
    if (wave < MT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < KW; ++w4) sum += red[w4][wave][lane][r];
        o[t][r] = sum;
      }
    }
    if (t + 1 < NT) __syncthreads();
  }
  if (wave >= MT) return;
  const int m = wave * 16 + fr;
  if (m >= p.M) return;
  if (p.ksplit > 1) {   // K slice of a split launch: raw fp32 partial sums, the epilogue runs in skinny_reduce_kernel
    float* wp = p.ws + ((long)blockIdx.y * p.M + m) * p.N;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + t * 16 + 4 * fh;
      if (n + 3 < p.N) store4(wp + n, o[t]);
      else
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) wp[n + r] = o[t][r];
    }
    return;
  }
  long orow = m;
  if (p.row_map) {
    orow = p.row_map[m];
    if (orow < 0) return;
  }
  const int n_total_out = SWIGLU ? (p.N >> 1) : p.N;
  if (MT == 1 && p.ssq_in) {
    float tot = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < KW; ++w4) tot += s_ssq[w4][fr];
    const float rstd = rsqrtf(tot / (float)p.K + p.ssq_eps);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t][r] *= rstd;
  }
  if (p.ln_stats) {
    const float mean = p.ln_stats[2 * m], rstd = p.ln_stats[2 * m + 1];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = min(n0 + t * 16 + 4 * fh + r, p.N - 1);
        o[t][r] = (o[t][r] - (p.ln_colsum ? mean * p.ln_colsum[n] : 0.f)) * rstd;
      }
  }
  constexpr int NOUT = SWIGLU ? NT / 2 : NT;     // 16-column output tiles of this workgroup
  float ssq_acc = 0.f;
#pragma unroll
  for (int j = 0; j < NOUT; ++j) {
    float val[4];
    const int nb = (SWIGLU ? (n0 >> 1) : n0) + 16 * j + 4 * fh;   // first of this lane's 4 consecutive output columns
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (SWIGLU) {
        const int ng = n0 + 32 * j + 4 * fh + r, nu = ng + 16;
        const float g = o[2 * j][r] + (p.bias ? p.bias[min(ng, p.N - 1)] : 0.f);
        const float u = o[2 * j + 1][r] + (p.bias ? p.bias[min(nu, p.N - 1)] : 0.f);
        val[r] = g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)) * u;
      } else {
        val[r] = apply_act(o[j][r] + (p.bias ? p.bias[min(nb + r, p.N - 1)] : 0.f), p.act);
      }
    }
    if (nb >= n_total_out) continue;
    const bool whole = nb + 3 < n_total_out;
    if (p.out_f32) {
      float* c = reinterpret_cast<float*>(p.C) + orow * p.ldc + nb;
      const float* rs = p.resid ? reinterpret_cast<const float*>(p.resid) + orow * p.ldr + nb : nullptr;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nb + r < n_total_out) c[r] = val[r] + (rs ? rs[r] : 0.f);
    } else {
      bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + orow * p.ldc + nb;
      const bf16_t* rs = p.resid ? reinterpret_cast<const bf16_t*>(p.resid) + orow * p.ldr + nb : nullptr;
      if (whole && ((reinterpret_cast<uintptr_t>(c) & 7) == 0) && (!rs || (reinterpret_cast<uintptr_t>(rs) & 7) == 0)) {
        if (rs) {
          const uint2 rv = *reinterpret_cast<const uint2*>(rs);
          val[0] += __builtin_bit_cast(float, rv.x << 16); val[1] += __builtin_bit_cast(float, rv.x & 0xffff0000u);
          val[2] += __builtin_bit_cast(float, rv.y << 16); val[3] += __builtin_bit_cast(float, rv.y & 0xffff0000u);
        }
        uint2 ov;
        ov.x = pack_bf16x2(val[0], val[1]);
        ov.y = pack_bf16x2(val[2], val[3]);
        *reinterpret_cast<uint2*>(c) = ov;
        const float q0 = __builtin_bit_cast(float, ov.x << 16), q1 = __builtin_bit_cast(float, ov.x & 0xffff0000u);
        const float q2 = __builtin_bit_cast(float, ov.y << 16), q3 = __builtin_bit_cast(float, ov.y & 0xffff0000u);
        ssq_acc += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb + r < n_total_out) {
            const bf16_t q = f32_to_bf16(val[r] + (rs ? bf16_to_f32(rs[r]) : 0.f));
            c[r] = q;
            ssq_acc += bf16_to_f32(q) * bf16_to_f32(q);
          }
      }
    }
  }
  if (MT == 1 && p.ssq_out && !p.out_f32) {
    // the 4 lanes that share output row m = fr (fh = 0..3) hold this workgroup's columns of it between them
    ssq_acc += __shfl_xor(ssq_acc, 16, 64);
    ssq_acc += __shfl_xor(ssq_acc, 32, 64);
    if (fh == 0) p.ssq_out[(long)blockIdx.x * 16 + fr] = ssq_acc;
  }
}

// Second half of a split-K weight-streaming product: out[m][n] = epi(sum over slices of ws[slice][m][n]), slices added
// in index order (deterministic), 4 consecutive columns per thread. bias -> act -> +resid, row map, bf16 / f32 output.
__global__ __launch_bounds__(256) void skinny_reduce_kernel(GemmArgs p) {
  if (p.swiglu) {
    // partial tiles hold the interleaved [gate x16 | up x16] columns; 4 consecutive OUTPUT columns per thread (they sit in
    // one 16-column group): out = silu(sum gate + bias_g) * (sum up + bias_u)
    const int nout = p.N >> 1, n4 = nout >> 2;   // N % 32 == 0
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
    const int ng = 32 * (n >> 4) + (n & 15);
    float g[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < p.ksplit; ++ks) {
      const float* wp = p.ws + ((long)ks * p.M + m) * p.N + ng;
      float a[4], b[4];
      load4(wp, a);
      load4(wp + 16, b);
#pragma unroll
      for (int r = 0; r < 4; ++r) { g[r] += a[r]; u[r] += b[r]; }
    }
    long orow = m;
    if (p.row_map) {
      orow = p.row_map[m];
      if (orow < 0) return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gg = g[r] + (p.bias ? p.bias[ng + r] : 0.f), uu = u[r] + (p.bias ? p.bias[ng + 16 + r] : 0.f);
      const float x = gg * __builtin_amdgcn_rcpf(1.0f + __expf(-gg)) * uu;
      if (p.out_f32) reinterpret_cast<float*>(p.C)[orow * p.ldc + n + r] = x;
      else reinterpret_cast<bf16_t*>(p.C)[orow * p.ldc + n + r] = f32_to_bf16(x);
    }
    return;
  }
  const int n4 = (p.N + 3) / 4;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * n4) return;
  const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const bool whole = n + 3 < p.N && (p.N & 3) == 0;
  for (int ks = 0; ks < p.ksplit; ++ks) {
    const float* wp = p.ws + ((long)ks * p.M + m) * p.N + n;
    if (whole) {
      float t[4];
      load4(wp, t);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += t[r];
    } else {
      for (int r = 0; r < 4; ++r)
        if (n + r < p.N) v[r] += wp[r];
    }
  }
  long orow = m;
  if (p.row_map) {
    orow = p.row_map[m];
    if (orow < 0) return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (n + r >= p.N) break;
    float x = apply_act(v[r] + (p.bias ? p.bias[n + r] : 0.f), p.act);
    if (p.out_f32) {
      if (p.resid) x += reinterpret_cast<const float*>(p.resid)[orow * p.ldr + n + r];
      reinterpret_cast<float*>(p.C)[orow * p.ldc + n + r] = x;
    } else {
      if (p.resid) x += bf16_to_f32(reinterpret_cast<const bf16_t*>(p.resid)[orow * p.ldr + n + r]);
      reinterpret_cast<bf16_t*>(p.C)[orow * p.ldc + n + r] = f32_to_bf16(x);
    }
  }
}

// Split-K launch for the 64-row decode products on narrow weights (N <= 8192: o_proj, down_proj): with 16 weight rows per
// workgroup every workgroup re-reads the whole activation matrix from L2 (4x the weight bytes at K = 4096, 1.4 MB per
// workgroup at K = 11008) and the fabric, not HBM, sets the time. 64 (or 32) rows per workgroup and 4 (or 2) K slices keep
// >= 192 workgroups while each reads only its slice of the activations. Returns false when no such split applies.
static bool launch_skinny_splitk(GemmArgs& p, hipStream_t s) {
  if (!p.ws || p.swiglu || p.ln_stats || p.M <= 32 || p.N > 8192) return false;
  const int tiles = (p.N + 15) / 16;
  int nt = 0, ks = 0;
  for (int cand_nt : {4, 2}) {
    for (int cand_ks : {4, 2}) {
      if (p.K % (128 * cand_ks)) continue;
      const int wgs = ((tiles + cand_nt - 1) / cand_nt) * cand_ks;
      if (wgs >= 192 && wgs <= 640) { nt = cand_nt; ks = cand_ks; break; }
    }
    if (nt) break;
  }
  if (!nt) return false;
  p.ksplit = ks;
  const dim3 b(256);
  if (nt == 4) hipLaunchKernelGGL((gemm_skinny_kernel<4, 4, false>), dim3((tiles + 3) / 4, ks), b, 0, s, p);
  else hipLaunchKernelGGL((gemm_skinny_kernel<4, 2, false>), dim3((tiles + 1) / 2, ks), b, 0, s, p);
  const long n_thr = (long)p.M * ((p.N + 3) / 4);
  hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)((n_thr + 255) / 256)), b, 0, s, p);
  return true;
}

template <int MT>
static void launch_skinny(const GemmArgs& p, int N, int swiglu, hipStream_t s) {
  // Weight rows per workgroup (16 * NT): every workgroup re-reads the activations (from L2), so the bytes a launch moves
  // are N*K*2 * (1 + MT/NT) and the fabric delivers ~7 TB/s of that mix: take the widest workgroup that still leaves
  // >= 192 of them (fewer, wider workgroups measured slower: N = 4096 stays at 256 x 16 rows). With one activation
  // tile (M <= 16) the narrow workgroups stream better (M = 1: 22 vs 28 us on qkv; M = 16 gate/up: 48 vs 55 us).
  const int tiles = (N + 15) / 16;
  const dim3 b(256);
  const bool nt4 = MT >= 2 && tiles / 4 >= 192;
  const bool nt2 = MT >= 2 && tiles / 2 >= 192;
  if (swiglu) {
    if (nt4) hipLaunchKernelGGL((gemm_skinny_kernel<MT, 4, true>), dim3((tiles + 3) / 4), b, 0, s, p);
    else hipLaunchKernelGGL((gemm_skinny_kernel<MT, 2, true>), dim3((tiles + 1) / 2), b, 0, s, p);
  } else {
    if (nt4) hipLaunchKernelGGL((gemm_skinny_kernel<MT, 4, false>), dim3((tiles + 3) / 4), b, 0, s, p);
    else if (nt2) hipLaunchKernelGGL((gemm_skinny_kernel<MT, 2, false>), dim3((tiles + 1) / 2), b, 0, s, p);
    else hipLaunchKernelGGL((gemm_skinny_kernel<MT, 1, false>), dim3(tiles), b, 0, s, p);
  }
}

}  // namespace

// Workgroups a persistent 8-wave launch takes (one per CU: 256 fills the chip). A caller that wants CUs left over for
// kernels of ANOTHER stream (HBM-bound decode steps beside the MFMA-bound encoder) lowers it for the launches it
// enqueues on ITS stream; multiples of 8 keep a workgroup's tiles on one XCD. The setting belongs to a stream (round 6;
// rounds 5's was one process-wide int: two models or two host threads in one process raced on it): a small table keyed by
// hipStream_t behind a mutex, entries at 256 are dropped, so it holds only streams that are capped right now.
namespace {
struct StreamCapTable {
  static constexpr int kSlots = 32;
  std::mutex mu;
  void* stream[kSlots];
  int cap[kSlots];
  int n = 0;
  int get(void* s) {
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < n; ++i)
      if (stream[i] == s) return cap[i];
    return 256;
  }
  // returns the previous cap of the stream, -1 when the table is full
  int set(void* s, int c) {
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < n; ++i)
      if (stream[i] == s) {
        const int old = cap[i];
        if (c == 256) { stream[i] = stream[n - 1]; cap[i] = cap[n - 1]; --n; }
        else cap[i] = c;
        return old;
      }
    if (c == 256) return 256;
    if (n == kSlots) return -1;
    stream[n] = s; cap[n] = c; ++n;
    return 256;
  }
};
StreamCapTable g_stream_caps;
}  // namespace
extern "C" int haff_gemm_stream_cap(void* stream, int cap) {
  if (!(cap >= 8 && cap <= 256 && (cap & 7) == 0)) return g_stream_caps.get(stream);   // not a valid cap: a query
  const int old = g_stream_caps.set(stream, cap);
  return old < 0 ? HAFF_ERR_UNSUPPORTED : old;
}

static bool haff_gemm_spec_enabled() {
#ifdef HAFF_TUNING   // HAFF_GEMM_NO_SPEC=1: every launch through the generic instance (A/B)
  static const bool off = [] { const char* e = getenv("HAFF_GEMM_NO_SPEC"); return e && atoi(e) != 0; }();
  return !off;
#else
  return true;
#endif
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(const GemmArgs& p, hipStream_t s, int nbatch = 1) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  int gx = tiles;
  if (WM * WN == 8) {   // persistent 8-wave tile: one workgroup per CU
    int cap = g_stream_caps.get((void*)s);
#ifdef HAFF_TUNING       // HAFF_GEMM_PERSIST: other cap, 0 = one tile per workgroup
    static const int cap_env = [] { const char* e = getenv("HAFF_GEMM_PERSIST"); return e ? atoi(e) : -1; }();
    if (cap_env >= 0) cap = cap_env;
#endif
    if (cap > 0 && gx > cap) gx = cap;
  }
  dim3 grid(gx, nbatch), block(64 * WM * WN);
  GemmArgs pd = p;
  if (WM * WN == 4) {
    long deep_max = 1024;   // 128x128 work items (tiles x slices) up to which the launch takes the two-K-tiles-in-flight loop
#ifdef HAFF_TUNING
    static const long dm_env = [] { const char* e = getenv("HAFF_GEMM_DEEP_MAX"); return e ? atol(e) : 1024L; }();
    deep_max = dm_env;
#endif
    pd.deep_k = (long)tiles * nbatch <= deep_max && ((p.K + BK - 1) / BK) >= 3;
  }
  const GemmArgs& pl = pd;
  if constexpr (BM == 256 && BN == 256 && WM * WN == 8) {
    // a specialised instance (see the GF_* flags) when the launch meets its host-side contract and its feature set has one
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool ok = nbatch == 1 && p.nb_inner == 0 && !p.out_f32 && (p.N % 256) == 0 && a16(p.C) && (p.ldc & 7) == 0 && a16(p.bias) &&
                    a16(p.resid) && (!p.resid || (p.ldr & 7) == 0) && a16(p.ln_stats) && a16(p.ln_colsum) && a16(p.row_map) &&
                    (!p.ln_colsum || p.ln_stats);
    if (ok && haff_gemm_spec_enabled()) {
      const unsigned f = (p.bias ? GF_BIAS : 0u) | (p.ln_stats ? GF_LN : 0u) | (p.ln_colsum ? GF_CSUM : 0u) |
                         ((p.resid || p.res32) ? GF_RES : 0u) | (p.res32 ? GF_RES32 : 0u) |
                         (p.stat_out ? GF_STAT : 0u) | (p.row_map ? GF_MAP : 0u) | (p.hm_d ? GF_HM : 0u) | (p.rope_cs ? GF_ROPE : 0u) |
                         ((unsigned)p.act << GF_ACT_SHIFT);
      const bool mfull = (p.M % 256) == 0;
#define HAFF_SPEC(FL, SW)                                                                                                   \
  if (f == ((FL) & ~(GF_SPEC | GF_RAGM)) && (p.swiglu != 0) == (SW) && (mfull || ((FL) & GF_RAGM))) {                       \
    hipLaunchKernelGGL((gemm_bf16_kernel<256, 256, 2, 4, false, SW, (FL) | GF_SPEC>), grid, block, 0, s, pl);               \
    return haff_check_launch();                                                                                             \
  }
      constexpr unsigned GELU_ = (unsigned)HAFF_ACT_GELU << GF_ACT_SHIFT, QGELU_ = (unsigned)HAFF_ACT_QUICK_GELU << GF_ACT_SHIFT;
      // the ViT-H blocks (131072 rows per 32 frames: always whole tiles): q|k|v global / windowed head-major, lin1, proj and lin2
      HAFF_SPEC(GF_BIAS | GF_LN | GF_CSUM, false)
      HAFF_SPEC(GF_BIAS | GF_LN | GF_CSUM | GF_MAP | GF_HM, false)
      HAFF_SPEC(GF_BIAS | GF_LN | GF_CSUM | GELU_, false)
      HAFF_SPEC(GF_BIAS | GF_RES | GF_STAT, false)
      HAFF_SPEC(GF_BIAS | GF_RES | GF_STAT | GF_RES32, false)   // ... on the fp32 residual stream (haff_gemm_bf16_rowstats32)
      // Llama prefill (64 x 291 rows: ragged last M-tile): q|k|v with RoPE, o_proj / down_proj, gate|up; CLIP and everything plain
      HAFF_SPEC(GF_ROPE | GF_RAGM, false)
      HAFF_SPEC(GF_RES | GF_RAGM, false)
      HAFF_SPEC(GF_RAGM, true)
      HAFF_SPEC(GF_RAGM, false)
      HAFF_SPEC(GF_BIAS | GF_RAGM, false)
      HAFF_SPEC(GF_BIAS | GF_RES | GF_RAGM, false)
      HAFF_SPEC(GF_BIAS | QGELU_ | GF_RAGM, false)
#undef HAFF_SPEC
    }
  }
  if constexpr (BM == 192 && BN == 256 && WM * WN == 8) {   // the fine-tune step's 2808-row products (forward, dX): plain / residual / SwiGLU
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool ok = nbatch == 1 && p.nb_inner == 0 && !p.out_f32 && (p.N % 256) == 0 && a16(p.C) && (p.ldc & 7) == 0 && a16(p.bias) &&
                    a16(p.resid) && (!p.resid || (p.ldr & 7) == 0) && !p.ln_stats && !p.ln_colsum && !p.row_map && !p.stat_out && !p.hm_d &&
                    !p.rope_cs && p.act == 0;
    if (ok && haff_gemm_spec_enabled()) {
      const unsigned f = (p.bias ? GF_BIAS : 0u) | (p.resid ? GF_RES : 0u);
#define HAFF_SPEC192(FL, SW)                                                                                               \
  if (f == ((FL) & ~(GF_SPEC | GF_RAGM)) && (p.swiglu != 0) == (SW)) {                                                       \
    hipLaunchKernelGGL((gemm_bf16_kernel<192, 256, 2, 4, false, SW, (FL) | GF_SPEC | GF_RAGM>), grid, block, 0, s, pl);     \
    return haff_check_launch();                                                                                             \
  }
      HAFF_SPEC192(0u, false)
      HAFF_SPEC192(GF_RES, false)
      HAFF_SPEC192(0u, true)
      HAFF_SPEC192(GF_BIAS, false)
      HAFF_SPEC192(GF_BIAS | GF_RES, false)
#undef HAFF_SPEC192
    }
  }
  if (p.res32) return HAFF_ERR_UNSUPPORTED;   // the fp32 residual stream exists in its specialised instance only
  if (p.swiglu) {
    if (p.out_f32) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, s, pl);
    else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, pl);
  } else {
    if (p.out_f32) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, s, pl);
    else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, s, pl);
  }
  return haff_check_launch();
}

#ifdef HAFF_GEMM_TRACE
extern "C" int haff_gemm_trace_read(unsigned long long* host, int n_words) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(haff_gemm_trace_buf), sizeof(unsigned long long) * n_words) == hipSuccess ? 0 : 1;
}
#endif

static int gemm_bf16_impl(const void* A, long lda, const int* a_map, long a_rows, const void* W, long ldw, void* C,
                          long ldc, const float* bias, const void* resid, long ldr, const int* row_map, int M, int N,
                          int K, int act, int out_f32, int swiglu, int tile_cfg, void* stream,
                          const float* ln_stats = nullptr, const float* ln_colsum = nullptr, void* workspace = nullptr,
                          long workspace_bytes = 0) {
  if (M <= 0 || N <= 0 || K <= 0) return HAFF_ERR_BAD_ARG;
  if ((K & 7) || (lda & 7) || (ldw & 7)) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return HAFF_ERR_BAD_ARG;
  if (swiglu && ((N & 31) || (ldc & 3) || resid)) return HAFF_ERR_BAD_ARG;
  GemmArgs p{reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(W), ldw, C, ldc,
             bias, resid, ldr, row_map, a_map, 8, ln_stats, ln_colsum, M, N, K, act, out_f32, swiglu, 0, 0, 0, 0, 0, 0, 0};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // weight-streaming kernel for decode-sized M; past 32 rows the 128x128 tile is faster again on very wide outputs
  // (M = 64: gate/up 53 vs 75 us, lm_head 68 vs 84 us; qkv 49 vs 42, o_proj 46 vs 24, down 113 vs 63)
  int skinny_max_m = 64;
#ifdef HAFF_TUNING   // HAFF_SKINNY_MAXM: rows up to which the weight-streaming kernel is taken (A/B against the split-K tile path)
  {
    static const int e = [] { const char* v = getenv("HAFF_SKINNY_MAXM"); return v ? atoi(v) : 64; }();
    skinny_max_m = e;
  }
#endif
  // 33..64 rows with a workspace: wide or deep weights go to the split-K tile path below (M = 64: qkv 31.7 vs 36.7 us, down_proj
  // 31.5 vs 42.4 us with 16 uneven K slices; o_proj stays here: 17.3 vs 20.6 us; tools/skinny64_ab.py)
  int tile_min_m = 32;
#ifdef HAFF_TUNING
  {
    static const int e = [] { const char* v = getenv("HAFF_TILE_MINM"); return v ? atoi(v) : 32; }();
    tile_min_m = e;
  }
#endif
  const bool tile_rows = M > tile_min_m && workspace && (K % BK) == 0 && !ln_stats && (N >= 8192 || K >= 8192);
  if (M <= skinny_max_m && (K % 128) == 0 && tile_cfg == 0 && !(M > 32 && N >= 16384) && !tile_rows) {
    if (M > 32 && workspace && workspace_bytes >= 4L * 4 * M * N && !a_map) {
      p.ws = reinterpret_cast<float*>(workspace);
      if (launch_skinny_splitk(p, s)) return haff_check_launch();
      p.ws = nullptr;
    }
    if (M <= 16) launch_skinny<1>(p, N, swiglu, s);
    else if (M <= 32) launch_skinny<2>(p, N, swiglu, s);
    else launch_skinny<4>(p, N, swiglu, s);
    return haff_check_launch();
  }
  // Few output tiles, long K (prefill-sized o_proj / down_proj, the CLIP tower at one frame): K is split over
  // blockIdx.y so that the 128x128 kernel's 512 resident-workgroup slots are used (96 tiles of 288x4096x4096 ran one
  // 64-step K loop each on 96 CUs: 62 us; 4 slices: 384 workgroups x 16 steps). Each slice is a "batch" of the batched
  // launch — operands offset by slice * K/ks along K, fp32 partial tile into ws[slice][M][N] — and skinny_reduce_kernel
  // adds the slices in index order and applies the epilogue: deterministic.
  if (tile_cfg == 0 && workspace && !ln_stats && M > (tile_min_m < 32 ? tile_min_m : 32) && (K % BK) == 0) {
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const int ksteps = K / BK;
    int ks = 0;
    int kchunk = 0;   // K-tiles per slice; the last slice may be shorter (K = 11008: 172 K-tiles = 15 x 11 + 7)
    if (t128 <= 256 && ksteps >= 32)   // (K = 1024: the second launch costs more than the shorter loop saves, 17.5 -> 19.8 us)
      for (int c : {16, 8, 4, 2}) {
        const int kc = (ksteps + c - 1) / c, n_sl = (ksteps + kc - 1) / kc;
        if (kc >= 4 && ksteps - (n_sl - 1) * kc >= 2 && t128 * n_sl <= 512 && 4L * n_sl * M * N <= workspace_bytes) { ks = n_sl; kchunk = kc; break; }
      }
    if (ks) {
      GemmArgs q = p;
      q.C = workspace; q.ldc = N; q.bias = nullptr; q.resid = nullptr; q.ldr = 0; q.row_map = nullptr;
      q.K = kchunk * BK; q.k_total = K; q.act = 0; q.out_f32 = 1; q.swiglu = 0;   // raw interleaved columns: the reduce kernel pairs them
      q.nb_inner = ks; q.sAo = 0; q.sWo = 0; q.sCo = 0; q.sAi = q.K; q.sWi = q.K; q.sCi = (long)M * N;
      const int rc = launch_gemm<128, 128, 2, 2>(q, s, ks);
      if (rc) return rc;
      p.ws = reinterpret_cast<float*>(workspace);
      p.ksplit = ks;
      const long n_thr = swiglu ? (long)M * (N / 8) : (long)M * ((N + 3) / 4);
      hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, s, p);
      return haff_check_launch();
    }
  }
  // the 8-wave tile addresses operands with 32-bit byte offsets and has no K-tail path
  const bool big_ok = (K % BK == 0) && ((long)(a_map ? a_rows : M) * lda * 2 < (1L << 32)) && ((long)N * ldw * 2 < (1L << 32));
  bool big = tile_cfg == 2 && big_ok;
  bool mid_tile = tile_cfg == 3 && big_ok;   // the 192 x 256 form of the 8-wave tile
  if (tile_cfg == 0 && big_ok) {
    // Pick the tile by wave-quantisation efficiency (tiles / slots rounded up) times the measured per-tile
    // advantage of the 256^2 kernel (tools/gemm_bench.py: ~1.2x at equal quantisation): 128^2 runs 2
    // workgroups per CU (512 slots), 256^2 one (256 slots).
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const double e128 = (double)t128 / (double)(((t128 + 511) / 512) * 512);
    const double e256 = (double)t256 / (double)(((t256 + 255) / 256) * 256) * 1.22;
    big = (M >= 256 && N >= 256 && e256 > e128);
    // 192-row tiles where they fill the chip's rounds better than 256-row ones by more than their smaller reuse costs (per
    // tile they move 7/8 of the bytes for 3/4 of the flops: x 0.93, tools/gemm_bench.py SHAPESET=train): the fine-tune step's
    // 2808 x 4096 outputs are 176 tiles = one round at 69 % with 256 rows, 240 = one round at 94 % with 192 (922 -> 980
    // TFLOP/s; K = 22016: 986 -> 1207); 5000 x 4096: 1010 -> 1115
    const long t192 = (long)((M + 191) / 192) * ((N + 255) / 256);
    const double e192 = (double)t192 / (double)(((t192 + 255) / 256) * 256) * 1.22 * 0.93;
    if (M >= 192 && N >= 256 && !ln_stats && e192 > 1.05 * (e256 > e128 ? e256 : e128)) { mid_tile = true; big = false; }
  }
  // Raster group depth: 8 M-tiles share their A panels across the N sweep; with a long K loop (>= 5120) and few N tiles
  // the concurrently running tiles drift apart and a 2-deep group keeps more of the sweep in the 4 MiB L2
  // (measured +5 % on 131072x1280x5120 and 18624x4096x11008, tools/gemm_variant.py).
  if (big || mid_tile) {
    // (sweep of 1..32 on the bench shapes, tools/gemm_variant.py with HAFF_GEMM_GROUP_M: <= 5 N-tiles: 1 (+4 % on
    // 131072x1280x1280), <= 16 N-tiles: 4 (+1.4 % on 131072x3840x1280), 20 N-tiles: 8; all within 3 % of each other)
    // (round 3, ring loop: <= 5 N-tiles 1; long K with <= 8 N-tiles 2; <= 16 N-tiles 4 (18624x4096x11008: 1351 vs 1315 at 2); 8)
    const int tn = (N + 255) / 256;
    if (tn <= 5) p.group_m = 1;
    else if (K >= 5120 && tn <= 8) p.group_m = 2;
    else if (tn <= 16) p.group_m = 4;
  }
#ifdef HAFF_TUNING
  {   // A/B override of the raster group depth (tools/gemm_variant.py)
    static const int gm_env = [] { const char* e = getenv("HAFF_GEMM_GROUP_M"); return e ? atoi(e) : 0; }();
    if (gm_env > 0) p.group_m = gm_env;
  }
#endif
#ifndef HAFF_GEMM_NO_NT
  p.nt_out = (long)M * (swiglu ? N / 2 : N) * (out_f32 ? 4 : 2) >= (64L << 20);
#endif
  if (mid_tile) return launch_gemm<192, 256, 2, 4>(p, s);
  return big ? launch_gemm<256, 256, 2, 4>(p, s) : launch_gemm<128, 128, 2, 2>(p, s);
}

// tile_cfg: 0 = auto, 1 = force the 128x128 tile, 2 = force the 256x256 tile, 3 = force the 192x256 tile (tests and A/B measurements)
extern "C" int haff_gemm_bf16_cfg(const void* A, long lda, const void* W, long ldw, void* C, long ldc,
                                  const float* bias, const void* resid, long ldr, const int* row_map,
                                  int M, int N, int K, int act, int out_f32, int swiglu, int tile_cfg, void* stream) {
  return gemm_bf16_impl(A, lda, nullptr, 0, W, ldw, C, ldc, bias, resid, ldr, row_map, M, N, K, act, out_f32, swiglu,
                        tile_cfg, stream);
}

extern "C" int haff_gemm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc,
                              const float* bias, const void* resid, long ldr, const int* row_map,
                              int M, int N, int K, int act, int out_f32, int swiglu, void* stream) {
  return gemm_bf16_impl(A, lda, nullptr, 0, W, ldw, C, ldc, bias, resid, ldr, row_map, M, N, K, act, out_f32, swiglu, 0,
                        stream);
}

// haff_gemm_bf16 with a caller-provided workspace (DEVICE memory, 16-B aligned, >= 16 * M * N bytes to be used): lets the
// weight-streaming kernel split K over workgroups for 33..64-row products on narrow weights (decode-step o_proj / down_proj
// at batch 64); fp32 partial tiles go through the workspace and are summed in a fixed order. Without a workspace (or when no
// split applies) this is haff_gemm_bf16.
extern "C" int haff_gemm_bf16_ws(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias,
                                 const void* resid, long ldr, const int* row_map, int M, int N, int K, int act, int out_f32,
                                 int swiglu, void* workspace, long workspace_bytes, void* stream) {
  if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15)) return HAFF_ERR_BAD_ARG;
  return gemm_bf16_impl(A, lda, nullptr, 0, W, ldw, C, ldc, bias, resid, ldr, row_map, M, N, K, act, out_f32, swiglu, 0,
                        stream, nullptr, nullptr, workspace, workspace_bytes);
}

// Decode-sized product (M <= 16; with ssq_in M <= 8 and ssq_n <= 512; K % 128 == 0; weight-streaming kernel) that carries Llama's RMSNorm between products
// without a norm kernel (LlamaDecoderLayer as reached from llava_llama.py:93-102: input_layernorm -> q/k/v,
// post_attention_layernorm -> gate/up):
//   ssq_in  != NULL: C = epi( rstd_m * (A . W^T) ), rstd_m = rsqrt(sum_{b < ssq_n} ssq_in[b][m] / K + eps), W = the weights with
//                    the norm's gamma folded into their columns (caller), A = the un-normalised residual stream;
//   ssq_out != NULL: (bf16 output) workgroup b also writes ssq_out[b][m] = sum over its output columns of bf16(C[m][n])^2,
//                    m < 16 — the partials the next product's ssq_in consumes; *n_parts_out = number of workgroups b.
// Both are fp32 [parts][16] device arrays. Sums run in a fixed order: results are bit-repeatable.
extern "C" int haff_gemm_bf16_rms(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias,
                                  const void* resid, long ldr, int M, int N, int K, int act, int out_f32, int swiglu,
                                  const float* ssq_in, int ssq_n, float eps, float* ssq_out, int* n_parts_out, void* stream) {
  if (M <= 0 || M > 16 || N <= 0 || K <= 0 || (K % 128) || (lda & 7) || (ldw & 7)) return HAFF_ERR_BAD_ARG;
  if (ssq_in && (M > 8 || ssq_n > 512)) return HAFF_ERR_BAD_ARG;   // consumer side: <= 8 rows, <= 512 producer workgroups
  if (ssq_in && (reinterpret_cast<uintptr_t>(ssq_in) & 15)) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return HAFF_ERR_BAD_ARG;
  if (swiglu && ((N & 31) || (ldc & 3) || resid)) return HAFF_ERR_BAD_ARG;
  if ((ssq_in && ssq_n <= 0) || (ssq_out && (out_f32 || swiglu))) return HAFF_ERR_BAD_ARG;
  GemmArgs p{reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(W), ldw, C, ldc,
             bias, resid, ldr, nullptr, nullptr, 8, nullptr, nullptr, M, N, K, act, out_f32, swiglu, 0, 0, 0, 0, 0, 0, 0};
  p.ssq_in = ssq_in; p.ssq_n = ssq_n; p.ssq_eps = eps; p.ssq_out = ssq_out;
  // launch_skinny<1>: 16 weight rows per workgroup (32 for SwiGLU pairs)
  if (n_parts_out) *n_parts_out = swiglu ? (N + 31) / 32 : (N + 15) / 16;
  launch_skinny<1>(p, N, swiglu, reinterpret_cast<hipStream_t>(stream));
  return haff_check_launch();
}

// Same with a gather on the A side: logical row m of the product reads A row a_map[m] (0 <= a_map[m] < a_rows).
// Used to run the window-unpartition projection only over real tokens (image_encoder.py:186-188,291-318: the padded
// window rows are dropped right after the projection, so they are never multiplied).
extern "C" int haff_gemm_bf16_gather(const void* A, long lda, const int* a_map, long a_rows, const void* W, long ldw,
                                     void* C, long ldc, const float* bias, const void* resid, long ldr,
                                     const int* row_map, int M, int N, int K, int act, int out_f32, int swiglu,
                                     void* stream) {
  if (!a_map || a_rows <= 0) return HAFF_ERR_BAD_ARG;
  return gemm_bf16_impl(A, lda, a_map, a_rows, W, ldw, C, ldc, bias, resid, ldr, row_map, M, N, K, act, out_f32, swiglu,
                        0, stream);
}

// Llama prefill q|k|v projection with rotate-half RoPE and the KV-cache append in the epilogue (transformers LlamaAttention.forward
// reached from llava_llama.py:93-102: q, k = apply_rotary_pos_emb(q_proj(x), k_proj(x)); cache append) — replaces haff_gemm_bf16 +
// haff_rope_cache on prefill-sized batches: q, k, v are never written un-rotated and read back (1.07 GB per layer at 64 x 291
// rows). A [M = B*T][K]; Wp [3*H*d][K] = the fused q|k|v weights with the rows of every 256-row tile PERMUTED (see
// GemmArgs::rope_cs; ops.rope_permute_rows builds it once); q_out [M][ldq] receives the rotated q (H*d columns);
// kcache / vcache [B][Tmax][H*d] rows pos0 .. pos0+T-1 receive the rotated k and v; cos_sin f32 [Tmax][128].
// d == 128, (H*d) % 256 == 0, K % 64 == 0, M < 2^22; otherwise HAFF_ERR_UNSUPPORTED.
extern "C" int haff_gemm_bf16_qkv_rope(const void* A, long lda, const void* Wp, long ldw, void* q_out, long ldq, void* kcache,
                                       void* vcache, const float* cos_sin, int B, int T, int Tmax, int pos0, int H, int d, int K,
                                       void* stream) {
  if (B <= 0 || T <= 0 || H <= 0 || K <= 0 || !q_out || !kcache || !vcache || !cos_sin || pos0 < 0 || pos0 + T > Tmax)
    return HAFF_ERR_BAD_ARG;
  if ((K & 7) || (lda & 7) || (ldw & 7) || (ldq & 7)) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(Wp) & 15) || (reinterpret_cast<uintptr_t>(q_out) & 15) ||
      (reinterpret_cast<uintptr_t>(kcache) & 15) || (reinterpret_cast<uintptr_t>(vcache) & 15) || (reinterpret_cast<uintptr_t>(cos_sin) & 15))
    return HAFF_ERR_BAD_ARG;
  const long M = (long)B * T;
  const int hd = H * d, N = 3 * hd;
  if (d != 128 || (hd % 256) || (K % BK) || M >= (1L << 22)) return HAFF_ERR_UNSUPPORTED;
  if (M * lda * 2 >= (1L << 32) || (long)N * ldw * 2 >= (1L << 32)) return HAFF_ERR_UNSUPPORTED;
  GemmArgs p{reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(Wp), ldw, q_out, ldq,
             nullptr, nullptr, 0, nullptr, nullptr, 8, nullptr, nullptr, (int)M, N, K, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  p.rope_cs = cos_sin; p.rope_k = kcache; p.rope_v = vcache;
  p.rope_t = T; p.rope_tmax = Tmax; p.rope_pos0 = pos0; p.rope_hd = hd;
  const int tn = N / 256;
  if (tn <= 5) p.group_m = 1;
  else if (K >= 5120 && tn <= 8) p.group_m = 2;
  else if (tn <= 16) p.group_m = 4;
  return launch_gemm<256, 256, 2, 4>(p, reinterpret_cast<hipStream_t>(stream));
}

// Residual product whose epilogue also emits the LayerNorm statistics of its OUTPUT rows (see GemmArgs::stat_out): proj and
// lin2 of a SAM block (image_encoder.py:186-193: x = shortcut + proj(attn); x = x + mlp(norm2(x))) hand the row sums of the new
// residual stream to haff_row_stats_finalize -> ln_stats of the next haff_gemm_bf16_ln, so neither a LayerNorm kernel nor a
// statistics pass reads the stream again. C = A.W^T + bias + resid in bf16 (C may alias resid); a_map optional (gather on
// the A side, as haff_gemm_bf16_gather). stat_out: f32 [M][N/64][2]. Needs whole 8-wave tiles: M % 256 == 0, N % 256 == 0,
// K % 64 == 0; otherwise HAFF_ERR_UNSUPPORTED (the caller keeps haff_row_stats).
extern "C" int haff_gemm_bf16_rowstats(const void* A, long lda, const int* a_map, long a_rows, const void* W, long ldw,
                                       void* C, long ldc, const float* bias, const void* resid, long ldr, int M, int N, int K,
                                       float* stat_out, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !resid || !stat_out) return HAFF_ERR_BAD_ARG;
  if ((K & 7) || (lda & 7) || (ldw & 7) || (ldc & 7) || (ldr & 7)) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(C) & 15) ||
      (reinterpret_cast<uintptr_t>(resid) & 15) || (reinterpret_cast<uintptr_t>(stat_out) & 7))
    return HAFF_ERR_BAD_ARG;
  if ((M % 256) || (N % 256) || (K % BK) || (a_map && a_rows <= 0)) return HAFF_ERR_UNSUPPORTED;
  if ((long)(a_map ? a_rows : M) * lda * 2 >= (1L << 32) || (long)N * ldw * 2 >= (1L << 32)) return HAFF_ERR_UNSUPPORTED;
  GemmArgs p{reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(W), ldw, C, ldc,
             bias, resid, ldr, nullptr, a_map, 8, nullptr, nullptr, M, N, K, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  p.stat_out = stat_out;
  p.stat_slots = N / 64;
  const int tn = N / 256;   // raster depth: the rule of gemm_bf16_impl
  if (tn <= 5) p.group_m = 1;
  else if (K >= 5120 && tn <= 8) p.group_m = 2;
  else if (tn <= 16) p.group_m = 4;
  return launch_gemm<256, 256, 2, 4>(p, reinterpret_cast<hipStream_t>(stream));
}

// haff_gemm_bf16_rowstats on an fp32 residual stream (round 6; the "fused fp32 stream" of DESIGN.md section 2): X32 f32 [M][ldx] is
// read, X32 + A.W^T + bias written back IN PLACE in fp32, C16 bf16 [M][ldc] receives the same values rounded once — the operand of
// the next haff_gemm_bf16_ln, whose folded LayerNorm takes the statistics emitted here (stat_out, from the fp32 values). The
// stream itself is never rounded between blocks. Same shape contract as haff_gemm_bf16_rowstats; ldx % 4 == 0.
extern "C" int haff_gemm_bf16_rowstats32(const void* A, long lda, const int* a_map, long a_rows, const void* W, long ldw,
                                         float* X32, long ldx, void* C16, long ldc, const float* bias, int M, int N, int K,
                                         float* stat_out, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !X32 || !C16 || !stat_out || !bias) return HAFF_ERR_BAD_ARG;
  if ((K & 7) || (lda & 7) || (ldw & 7) || (ldc & 7) || (ldx & 3)) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(C16) & 15) ||
      (reinterpret_cast<uintptr_t>(X32) & 15) || (reinterpret_cast<uintptr_t>(stat_out) & 7) || (reinterpret_cast<uintptr_t>(bias) & 15))
    return HAFF_ERR_BAD_ARG;
  if ((M % 256) || (N % 256) || (K % BK) || (a_map && a_rows <= 0)) return HAFF_ERR_UNSUPPORTED;
  if ((long)(a_map ? a_rows : M) * lda * 2 >= (1L << 32) || (long)N * ldw * 2 >= (1L << 32)) return HAFF_ERR_UNSUPPORTED;
  GemmArgs p{reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(W), ldw, C16, ldc,
             bias, nullptr, 0, nullptr, a_map, 8, nullptr, nullptr, M, N, K, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  p.stat_out = stat_out;
  p.stat_slots = N / 64;
  p.res32 = X32;
  p.ld32 = ldx;
  const int tn = N / 256;   // raster depth: the rule of gemm_bf16_impl
  if (tn <= 5) p.group_m = 1;
  else if (K >= 5120 && tn <= 8) p.group_m = 2;
  else if (tn <= 16) p.group_m = 4;
  return launch_gemm<256, 256, 2, 4>(p, reinterpret_cast<hipStream_t>(stream));   // always meets the specialised instance's contract
}

// Product with a LayerNorm / RMSNorm folded in (see GemmArgs::ln_stats): the normalised activations never exist in
// HBM. The caller scales W's columns by gamma, adds W.beta to the bias (haff side: sam.py / llava.py at load time) and
// passes the per-row {mean, rstd} from haff_row_stats. Replaces norm1->qkv and norm2->lin1 of the SAM blocks
// (image_encoder.py:179,191), input/post-attention RMSNorm -> qkv / gate-up of Llama, layer_norm1/2 of CLIP.
extern "C" int haff_gemm_bf16_ln(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias,
                                 const void* resid, long ldr, const int* row_map, const float* ln_stats,
                                 const float* ln_colsum, int M, int N, int K, int act, int out_f32, int swiglu,
                                 void* stream) {
  if (!ln_stats) return HAFF_ERR_BAD_ARG;
  return gemm_bf16_impl(A, lda, nullptr, 0, W, ldw, C, ldc, bias, resid, ldr, row_map, M, N, K, act, out_f32, swiglu, 0,
                        stream, ln_stats, ln_colsum);
}

// Product (optionally with a folded norm, as haff_gemm_bf16_ln) whose output is scattered HEAD-MAJOR: the windowed q|k|v
// projection of a ViT-H block (image_encoder.py:223-224 qkv Linear + :263-288 window_partition + :238-239 the reshape into heads)
// writes q, k and v of every (window, head) as ONE contiguous [tokens][d] block, so that the window attention kernel's K / V
// staging reads whole 128-B lines (round 4 measured the token-major layout's 160-B pieces at a 7680-B stride as what paces it).
//   product column n = part * (heads * d) + h * d + c, product row m  ->  C[part * part_stride + h * head_stride + row_map[m] * d + c]
// (bf16 elements). With row_map[m] = window * heads * n_tok + token, head_stride = n_tok * d and part_stride = (windows + 1) *
// heads * n_tok * d, C is [part][window][head][token][d] with one spare window for the pad token. row_map entries < 0 drop the row.
// bias f32 [N]; ln_stats / ln_colsum as haff_gemm_bf16_ln or both null. Whole 256 x 256 tiles only (M % 256 == 0, N % 256 == 0,
// K % 64 == 0), N == parts * heads * d with parts <= 3, d % 8 == 0: otherwise HAFF_ERR_UNSUPPORTED (-2) and the caller keeps the
// token-major layout.
extern "C" int haff_gemm_bf16_heads(const void* A, long lda, const void* W, long ldw, void* C, const float* bias, const int* row_map,
                                    const float* ln_stats, const float* ln_colsum, int M, int N, int K, int d, int heads,
                                    long part_stride, long head_stride, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || d <= 0 || heads <= 0 || !row_map || !C) return HAFF_ERR_BAD_ARG;
  if ((K & 7) || (lda & 7) || (ldw & 7) || (part_stride & 7) || (head_stride & 7)) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(C) & 15))
    return HAFF_ERR_BAD_ARG;
  const long hd = (long)heads * d;
  if ((d & 7) || d > 4096 || (M % 256) || (N % 256) || (K % BK) || N % hd != 0 || N / hd > 3 || hd >= (1L << 16)) return HAFF_ERR_UNSUPPORTED;
  if ((long)M * lda * 2 >= (1L << 32) || (long)N * ldw * 2 >= (1L << 32)) return HAFF_ERR_UNSUPPORTED;
  GemmArgs p{reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(W), ldw, C, (long)d,
             bias, nullptr, 0, row_map, nullptr, 8, ln_stats, ln_colsum, M, N, K, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  p.hm_d = d; p.hm_hd = (int)hd; p.hm_part = part_stride; p.hm_head = head_stride;
  const int tn = N / 256;   // raster depth: the rule of gemm_bf16_impl
  if (tn <= 5) p.group_m = 1;
  else if (K >= 5120 && tn <= 8) p.group_m = 2;
  else if (tn <= 16) p.group_m = 4;
  p.nt_out = (long)M * N * 2 >= (64L << 20);
  return launch_gemm<256, 256, 2, 4>(p, reinterpret_cast<hipStream_t>(stream));
}

// Batched C_z = A_z . W_z^T (no epilogue): z = zo * nb_inner + zi, operand offsets zo * s?o + zi * s?i (elements).
// Used by the training path for attention-shaped products over (batch, head) (scores, P.V and their gradients).
extern "C" int haff_gemm_bf16_batched(const void* A, long lda, long sAo, long sAi, const void* W, long ldw, long sWo,
                                      long sWi, void* C, long ldc, long sCo, long sCi, int nb_outer, int nb_inner,
                                      int M, int N, int K, int out_f32, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || nb_outer <= 0 || nb_inner <= 0) return HAFF_ERR_BAD_ARG;
  if ((K & 7) || (lda & 7) || (ldw & 7) || (sAo & 7) || (sAi & 7) || (sWo & 7) || (sWi & 7)) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return HAFF_ERR_BAD_ARG;
  GemmArgs p{reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(W), ldw, C, ldc,
             nullptr, nullptr, 0, nullptr, nullptr, 8, nullptr, nullptr, M, N, K, 0, out_f32, 0, nb_inner, sAo, sAi, sWo, sWi, sCo, sCi};
  return launch_gemm<128, 128, 2, 2>(p, reinterpret_cast<hipStream_t>(stream), nb_outer * nb_inner);
}
