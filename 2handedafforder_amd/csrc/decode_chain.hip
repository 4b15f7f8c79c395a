// KV-cached decode step of the Llama stack at <= 8 rows as ONE launch: every layer's five stages
//   q|k|v product (RMSNorm carried in) -> RoPE + cache append + attention -> o_proj (+residual) -> gate|up SwiGLU (RMSNorm carried
//   in) -> down_proj (+residual)
// (transformers LlamaDecoderLayer as reached from llava_llama.py:93-102 during LISA.py:443-450's greedy generate) are workgroup
// RANGES of one grid, chained by arrival counters instead of kernel boundaries.
//
// Why: at <= 8 rows a decode step is a stream over 13.2 GB (7B) of weights that the five-launches-per-layer form reads at 4.1 TB/s
// (98 us per layer against 65 us of pure stream; rocprofv3: q|k|v / o / down 18 us average, gate|up 35 us, attention 8.7 us):
// every launch boundary drains the memory pipe. Here a workgroup of stage s+1 is dispatched as soon as a slot frees up, requests the
// first two register sets of ITS weight slab — weights depend on nothing — and only then waits for stage s's counter.
// What it buys (MI355X, same box, hipGraph replay): 3.24 -> 3.10 ms per step at one row, 3.29 -> 3.15 at two, 3.37 -> 3.25 at four
// (-4 %); the rest of the distance to the stream bound is the hand-off itself: ~4.4 us per stage from the last producer's store to
// the first consumer's MFMA (store drain, shard add, replica adds, poll, sc1 loads of the activations) where a kernel boundary
// costs 1.2-1.9 us but lets nothing run ahead. (With the waits switched off — a bug of the first version, below — the same grid runs
// a step in 2.4 ms: that, not 3.1, is what the weight stream alone costs.)
//
// Three things the measurements forced:
//  * hand-offs are `sc1` (agent-scope, write-through) stores and `sc1` loads, drained before ONE lane signals — no release / acquire
//    fence per workgroup: the first version fenced (buffer_wbl2 sc1 / buffer_inv sc1) in each of the 64 000 workgroups of a step and
//    ran at 9.1 ms per step;
//  * arrival counters are SHARDED (8 shards + 8 replicas of the "shards complete" count per stage, a 128-byte line each): ~12 ns per
//    agent-scope add on one word made 768 arrivals 9 us, a tenth of a layer;
//  * o_proj / down_proj split K over TWO workgroups per 16-row tile: the dispatcher does not deal a stage's workgroups one per CU
//    (down_proj's 256 landed on 210 CUs, 46 of them carrying two) and a CU takes in ~22 GB/s whatever runs on it, so the stage ran
//    32 us for 90 MB; 512 half-K workgroups land two per CU on every CU (tools/chain_trace.py, -DCH_PLACE): 21 us;
//  * the counters are zeroed WRITE-THROUGH by a kernel in front of the launch: behind a hipMemsetAsync node the chained step was
//    bit-repeatable eagerly and a different result on every hipGraph replay (tools/chain_stress.py) — and 25 % faster, because the
//    waits saw the previous replay's counts and let stages start early.
//
// Deadlock freedom: a workgroup waits only for workgroups with a LOWER blockIdx.x (earlier stage of the same layer, or the last
// stage of the previous layer); the dispatcher hands out workgroups of a 1-D grid in index order per XCD, so everything a resident
// workgroup waits for is resident or finished, and stage 0 of layer 0 waits for nothing — on any number of free CUs (beside
// another stream's kernels too). Every spin is bounded anyway: a wait that runs out sets the sticky error word behind the counters,
// every later wait returns at once, the grid drains (the result is then garbage; haff_decode_chain_status reports it and
// LisaMI355.generate raises).
//
// Visibility between workgroups (per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by another CU's stores):
// every byte one stage hands to the next — the residual stream, q|k|v, the attention output, the SwiGLU output, the partial sums of
// squares, the K halves' partial tiles — is STORED write-through at agent scope (`sc1`: 8-byte / 4-byte relaxed agent atomics) and
// LOADED with `sc1` loads (buffer_load_dwordx4 / global_load_dwordx2 ... sc1 to registers), the storing wave drains its stores
// (s_waitcnt vmcnt(0)) before ONE lane adds to the stage's arrival counter, one lane of a consumer polls with relaxed agent loads and
// the workgroup barrier stands between the poll and every load of the handed-off bytes. Bytes no workgroup of the launch writes
// (weights, old cache rows, RoPE table, positions) are plain loads; the cache rows appended here are read by later launches only.
//
// Arithmetic: the product stage follows gemm_skinny_kernel<1, NT, SWIGLU, 4> of gemm_bf16.hip (same k-step order per accumulator,
// same LDS reduction over the four waves, same epilogue; o_proj / down_proj add their two K halves in the fixed order half 0 +
// half 1), the attention stage attn_decode_kernel<SPLIT, ROPE, NW = 4> of attention.hip. Against the five-launch layer the step
// agrees to a few bf16 ulps of the output scale (hipcc contracts the FMAs of the two translation units differently; at <= 4 rows
// the library splits a head over 16 waves); against ITSELF launched stage by stage (per_stage_launches: every wait satisfied by
// stream order) it is BIT-IDENTICAL — what tests/test_decode_chain_gpu.py and tests/test_fullsize_gpu.py hold it to.
#include "haff_common.h"

namespace {

constexpr int CH_MAXL = 48;      // layers per launch (7B: 32, 13B: 40)
constexpr int CH_STAGES = 5;
constexpr int CH_D = 128;        // head dim
constexpr int CH_KW = 4;         // waves per workgroup: K quarters of a product, key quarters of a head
constexpr int CH_U = 8;          // k-steps per register set (skinny_batch(1, 1) == skinny_batch(2, 1) == 8)
// experiment knobs (tools/build_chain_variant.sh -DCH_...): the product build takes the defaults below
#ifndef CH_PRE_TRIPS
#define CH_PRE_TRIPS 4
#endif
#ifndef CH_SLEEP
#define CH_SLEEP 16
#endif
#ifndef CH_W_NT
#define CH_W_NT 0               // 1: non-temporal loads on the weight stream
#endif
#ifndef CH_U_GU
#define CH_U_GU 8               // k-steps per register set of the gate|up stage (two weight tiles per workgroup)
#endif
constexpr int CH_PRE = CH_PRE_TRIPS;        // attention: trips of 16 key groups (64 keys each) whose K / V rows are requested BEFORE the wait (128 registers at 4 trips: 256 keys)
constexpr int CH_SPIN_LIMIT = 400000;
constexpr float CH_LOG2E = 1.4426950408889634f;

struct ChainLayer {
  const bf16_t *wqkv, *wo, *wgu, *wd;   // [3H][H] (gamma of input_layernorm folded in), [H][H], [2F][H] (16-row gate / up groups, gamma folded), [H][F]
  bf16_t *kc, *vc;                      // [B][tmax][H]
};

struct ChainArgs {
  ChainLayer L[CH_MAXL];
  bf16_t *x, *qkv, *att, *g;   // [M][H] residual stream (in / out), [M][3H], [M][H], [M][F] scratch
  float *ssq_a, *ssq_b;        // [H/16][16] partial sums of squares (o_proj -> gate|up, down_proj -> next q|k|v)
  float* ws;                   // [H/16][2][16][16] f32: the K halves' partial tiles of o_proj / down_proj
  const float* stats0;         // [M][2] {mean, rstd} of the incoming rows (layer 0's RMSNorm)
  const float* cos_sin;        // f32 [tmax][128]
  const int* nk_rows;          // device int32 [M]: position of the new token + 1
  unsigned* sync;              // [n_layers * 5][CH_STAGE_WORDS] arrival counters (zeroed by the launcher), then 1 sticky error word
  int n_layers, M, H, F, nh, tmax;
  float eps, scale;
  int nb[CH_STAGES];
  int per_layer;
  int block0;                  // first logical block of this launch (one launch per stage: tests, A/B)
};

// Arrival counters of one (layer, stage): 8 SHARDS (workgroup r of the stage adds to shard r & 7: ~12 ns per agent-scope add on one
// word — 768 arrivals on ONE word are 9 us, a tenth of a layer) and 8 REPLICAS of the "shards complete" count (the workgroup that
// completes a shard adds to every replica, 8 lanes of one instruction; a waiting workgroup polls ONE replica, so the pollers of a
// stage spread over 8 lines). Every word on a 128-byte line of its own.
constexpr int CH_SHARDS = 8;
#ifndef CH_LINE_WORDS
#define CH_LINE_WORDS 32
#endif
constexpr int CH_LINE = CH_LINE_WORDS;                       // words per counter line (32 = one 128-byte line each)
constexpr int CH_STAGE_WORDS = 2 * CH_SHARDS * CH_LINE;      // 8 shard lines + 8 replica lines

#ifdef CH_PLACE   // experiment builds only: where layer 2's workgroups ran (no atomics: the timing is the product build's)
__device__ unsigned ch_place_buf[CH_STAGES][1024];   // (xcc << 16) | HW_ID[15:0] of wave 0
__device__ unsigned long long ch_place_t[CH_STAGES][1024];
extern "C" int haff_decode_chain_place_read(unsigned* host, unsigned long long* t) {
  if (t && hipMemcpyFromSymbol(t, HIP_SYMBOL(ch_place_t), sizeof(unsigned long long) * CH_STAGES * 1024) != hipSuccess) return 1;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(ch_place_buf), sizeof(unsigned) * CH_STAGES * 1024) == hipSuccess ? 0 : 1;
}
#endif
#ifdef CH_TRACE   // experiment builds only: per (layer, stage) {first start, first wait satisfied, last wait satisfied, last end} (100 MHz clock)
__device__ unsigned long long ch_trace_buf[CH_MAXL * CH_STAGES][4];
#define CH_TRACE_MIN(slot, k) do { if (threadIdx.x == 0 && (blockIdx.x & 15) == 0) atomicMin(&ch_trace_buf[slot][k], (unsigned long long)wall_clock64()); } while (0)
#define CH_TRACE_MAX(slot, k) do { if (threadIdx.x == 0 && (blockIdx.x & 15) == 0) atomicMax(&ch_trace_buf[slot][k], (unsigned long long)wall_clock64()); } while (0)
extern "C" int haff_decode_chain_trace_read(unsigned long long* host, int reset) {
  if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(ch_trace_buf), sizeof(unsigned long long) * CH_MAXL * CH_STAGES * 4) != hipSuccess) return 1;
  if (reset) {
    static unsigned long long init[CH_MAXL * CH_STAGES][4];
    for (int i = 0; i < CH_MAXL * CH_STAGES; ++i) { init[i][0] = ~0ull; init[i][1] = ~0ull; init[i][2] = 0; init[i][3] = 0; }
    if (hipMemcpyToSymbol(HIP_SYMBOL(ch_trace_buf), init, sizeof(init)) != hipSuccess) return 1;
  }
  return 0;
}
#else
#define CH_TRACE_MIN(slot, k) do {} while (0)
#define CH_TRACE_MAX(slot, k) do {} while (0)
#endif

// nb_dep: workgroups of the stage waited for (its non-empty shards: min(nb_dep, 8) — the q|k|v / attention stages of a narrow model
// have fewer workgroups than shards)
__device__ __forceinline__ void chain_wait(unsigned* sync, int idx, int replica, unsigned* err, int nb_dep) {
  CH_TRACE_MIN(idx + 1, 0);
  if (idx < 0) return;
  const unsigned full = (unsigned)min(nb_dep, CH_SHARDS);
  if (threadIdx.x == 0) {
    const unsigned* w = sync + (long)idx * CH_STAGE_WORDS + (CH_SHARDS + replica) * CH_LINE;
    int spins = 0;
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < full) {
      __builtin_amdgcn_s_sleep(CH_SLEEP);
      ++spins;
      if ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      if (spins > CH_SPIN_LIMIT) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  CH_TRACE_MIN(idx + 1, 1);
  CH_TRACE_MAX(idx + 1, 2);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the compiler from moving the sc1 loads above the poll
}

// called by the ONE wave that stored the workgroup's results (every store of handed-off bytes is an sc1 store), after its stores;
// r = index of the workgroup within its stage, nb = workgroups of the stage
__device__ __forceinline__ void chain_signal(unsigned* sync, int idx, int r, int nb) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CH_TRACE_MAX(idx, 3);
  unsigned* base = sync + (long)idx * CH_STAGE_WORDS;
  const int lane = threadIdx.x & 63;
  const int shard = r & (CH_SHARDS - 1);
  const unsigned quota = (unsigned)((nb - shard + CH_SHARDS - 1) / CH_SHARDS);   // workgroups r' < nb with r' & 7 == shard
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(base + shard * CH_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __builtin_amdgcn_readfirstlane(old);
  if (old == quota - 1u && lane < CH_SHARDS)   // every other arrival of this shard drained its stores before its add
    __hip_atomic_fetch_add(base + (CH_SHARDS + lane) * CH_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// agent-scope (sc1) accesses of handed-off bytes
typedef unsigned long long ch_u64;
typedef __attribute__((ext_vector_type(4))) unsigned ch_u32x4;
__device__ __forceinline__ void st8_sc1(void* p, uint2 v) {
  __hip_atomic_store(reinterpret_cast<ch_u64*>(p), ((ch_u64)v.y << 32) | (ch_u64)v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint2 ld8_sc1(const void* p) {
  const ch_u64 x = __hip_atomic_load(reinterpret_cast<const ch_u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint2((unsigned)x, (unsigned)(x >> 32));
}
__device__ __forceinline__ void st4_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ch_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const ch_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);   // aux 16 = sc1
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// ---- product stage: C[M][N] = epi(X[M][K] . W[N][K]^T), M <= 8 rows, 16 * NT weight rows per workgroup ----------------------
// KS = 2 (o_proj, down_proj: H / 16 weight tiles, ONE per CU — and the dispatcher does not spread them one per CU: rocprof of the
// KS = 1 form found down_proj's 256 workgroups on 210 CUs, 46 of them carrying two, and a CU takes in ~22 GB/s whatever runs on it:
// 32 us for 90 MB): the K extent of a tile is split over TWO workgroups (2 bx = tile, half); each stores its fp32 partial tile
// write-through, draws a ticket, and the one that draws the second ticket adds the halves in the fixed order half 0 + half 1,
// runs the epilogue and signals the stage. 512 finer workgroups fill the chip evenly.
template <int NT, bool SWIGLU, int KS = 1>
__device__ __forceinline__ void chain_product(const bf16_t* __restrict__ W, int N, int K, const bf16_t* X, long ldx, int M, bf16_t* C,
                                              long ldc, const bf16_t* resid, long ldr, const float* ssq_in, int ssq_n, float eps,
                                              const float* ln_stats, float* ssq_out, int bx_in, int nb_stage, int nb_dep, unsigned* sync,
                                              int wait_idx, int signal_idx, unsigned* err, float* ws = nullptr, unsigned* tickets = nullptr) {
  static_assert(KS == 1 || (NT == 1 && !SWIGLU), "the K split serves the one-tile residual products");
  const int bx = bx_in / KS, half = bx_in - bx * KS;
  constexpr int U = NT == 2 ? CH_U_GU : CH_U, KW = CH_KW;
  __shared__ float red[KW][64][4];
  __shared__ float s_ssq[KW][16];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int fr = lane & 15, fh = lane >> 4;
  const int n0 = bx * 16 * NT;
  const int kq = K / (KW * KS);
  const int k_lo = (half * KW + wave) * kq;
  const __amdgpu_buffer_rsrc_t xr = ch_rsrc(X, (unsigned)(M * ldx * 2));
  const unsigned xoff = (unsigned)((min(fr, M - 1) * ldx + k_lo + fh * 8) * 2);   // bytes
  const bf16_t* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wrow[t] = W + (long)min(n0 + t * 16 + fr, N - 1) * K + k_lo + fh * 8;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 wv[2][NT][U], xv[2][U];
  auto load_w = [&](int set, int k) {
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      const int k0 = min(k + 32 * u, kq - 32), k1 = min(k + 32 * u + 32, kq - 32);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#if CH_W_NT
        {
          const ch_u32x4 w0 = __builtin_nontemporal_load(reinterpret_cast<const ch_u32x4*>(wrow[t] + k0));
          const ch_u32x4 w1 = __builtin_nontemporal_load(reinterpret_cast<const ch_u32x4*>(wrow[t] + k1));
          wv[set][t][u] = make_uint4(w0[0], w0[1], w0[2], w0[3]);
          wv[set][t][u + 1] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
        }
#else
        wv[set][t][u] = *reinterpret_cast<const uint4*>(wrow[t] + k0);
        wv[set][t][u + 1] = *reinterpret_cast<const uint4*>(wrow[t] + k1);
#endif
      }
    }
  };
  auto load_x = [&](int set, int k) {
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      const int k0 = min(k + 32 * u, kq - 32), k1 = min(k + 32 * u + 32, kq - 32);
#ifdef CH_EXP_PLAINX   // timing only: plain (L1-served) activation loads — stale data
      xv[set][u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(X) + xoff + 2 * k0);
      xv[set][u + 1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(X) + xoff + 2 * k1);
#else
      xv[set][u] = ld16_sc1(xr, xoff + 2 * k0);
      xv[set][u + 1] = ld16_sc1(xr, xoff + 2 * k1);
#endif
    }
  };
  auto compute = [&](int set, int k) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k + 32 * u < kq) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[set][t][u]),
                                                           __builtin_bit_cast(bf16x8, xv[set][u]), acc[t], 0, 0, 0);
      }
    }
  };
  constexpr int KB = 32 * U;
  // the weights depend on nobody: both register sets are on their way before this workgroup asks whether its inputs exist
  load_w(0, 0);
  if (KB < kq) load_w(1, KB);
  chain_wait(sync, wait_idx, bx_in & (CH_SHARDS - 1), err, nb_dep);

  constexpr int SSQ_M = 8;
  float ssq_a[SSQ_M], ssq_b[SSQ_M];
  if (ssq_in) {
    const int b0 = threadIdx.x, b1 = threadIdx.x + 64 * KW;
    const __amdgpu_buffer_rsrc_t sr = ch_rsrc(ssq_in, (unsigned)(ssq_n * 64));
#pragma unroll
    for (int h4 = 0; h4 < SSQ_M / 4; ++h4) {
      float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
      if (4 * h4 < M) {
        if (b0 < ssq_n) {
          const uint4 t = ld16_sc1(sr, (unsigned)(b0 * 64 + 16 * h4));
          va[0] = __builtin_bit_cast(float, t.x); va[1] = __builtin_bit_cast(float, t.y); va[2] = __builtin_bit_cast(float, t.z); va[3] = __builtin_bit_cast(float, t.w);
        }
        if (b1 < ssq_n) {
          const uint4 t = ld16_sc1(sr, (unsigned)(b1 * 64 + 16 * h4));
          vb[0] = __builtin_bit_cast(float, t.x); vb[1] = __builtin_bit_cast(float, t.y); vb[2] = __builtin_bit_cast(float, t.z); vb[3] = __builtin_bit_cast(float, t.w);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { ssq_a[4 * h4 + e] = va[e]; ssq_b[4 * h4 + e] = vb[e]; }
    }
  }
  load_x(0, 0);
  for (int k = 0; k < kq; k += 2 * KB) {
    if (k + KB < kq) {
      if (k > 0) load_w(1, k + KB);
      load_x(1, k + KB);
    }
    compute(0, k);
    if (k + KB < kq) {
      if (k + 2 * KB < kq) { load_w(0, k + 2 * KB); load_x(0, k + 2 * KB); }
      compute(1, k + KB);
    }
  }
  if (ssq_in) {
#pragma unroll
    for (int mm = 0; mm < SSQ_M; ++mm)
      if (mm < M) {
        float v = ssq_a[mm] + ssq_b[mm];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_ssq[wave][mm] = v;
      }
  }
  float o[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
    store4(&red[wave][lane][0], v);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < KW; ++w4) sum += red[w4][lane][r];
        o[t][r] = sum;
      }
    }
    if (t + 1 < NT) __syncthreads();
  }
  if (wave != 0) return;
  const int m = fr;
  if constexpr (KS == 2) {
    // partial tile [half][m][16 columns] f32, write-through; the ticket decides who combines
    float* mine = ws + (((long)bx * 2 + half) * 16 + m) * 16 + 4 * fh;
    if (m < M) {
      st8_sc1(mine, make_uint2(__builtin_bit_cast(unsigned, o[0][0]), __builtin_bit_cast(unsigned, o[0][1])));
      st8_sc1(mine + 2, make_uint2(__builtin_bit_cast(unsigned, o[0][2]), __builtin_bit_cast(unsigned, o[0][3])));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + bx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if ((ticket & 1u) == 0u) return;      // first of the pair: the other workgroup finishes the tile
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (m < M) {
      const __amdgpu_buffer_rsrc_t wr = ch_rsrc(ws, (unsigned)((long)nb_stage * 16 * 16 * 4));   // nb_stage = 2 * tiles
      const uint4 t = ld16_sc1(wr, (unsigned)(((((long)bx * 2 + (1 - half)) * 16 + m) * 16 + 4 * fh) * 4));
      const float other[4] = {__builtin_bit_cast(float, t.x), __builtin_bit_cast(float, t.y), __builtin_bit_cast(float, t.z), __builtin_bit_cast(float, t.w)};
#pragma unroll
      for (int r = 0; r < 4; ++r) o[0][r] = half == 0 ? o[0][r] + other[r] : other[r] + o[0][r];   // always half 0 + half 1
    }
  }
  if (m < M) {
    if (ssq_in) {
      float tot = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < KW; ++w4) tot += s_ssq[w4][fr];
      const float rstd = rsqrtf(tot / (float)K + eps);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][r] *= rstd;
    }
    if (ln_stats) {
      const float rstd = ln_stats[2 * m + 1];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][r] = (o[t][r] - 0.f) * rstd;
    }
    constexpr int NOUT = SWIGLU ? NT / 2 : NT;
    float ssq_acc = 0.f;
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
      float val[4];
      const int nb = (SWIGLU ? (n0 >> 1) : n0) + 16 * j + 4 * fh;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (SWIGLU) {
          const float gg = o[2 * j][r] + 0.f;
          const float uu = o[2 * j + 1][r] + 0.f;
          val[r] = gg * __builtin_amdgcn_rcpf(1.0f + __expf(-gg)) * uu;
        } else {
          val[r] = o[j][r] + 0.f;
        }
      }
      bf16_t* c = C + (long)m * ldc + nb;
      if (resid) {
        const uint2 rv = ld8_sc1(resid + (long)m * ldr + nb);
        val[0] += __builtin_bit_cast(float, rv.x << 16); val[1] += __builtin_bit_cast(float, rv.x & 0xffff0000u);
        val[2] += __builtin_bit_cast(float, rv.y << 16); val[3] += __builtin_bit_cast(float, rv.y & 0xffff0000u);
      }
      uint2 ov;
      ov.x = pack_bf16x2(val[0], val[1]);
      ov.y = pack_bf16x2(val[2], val[3]);
      st8_sc1(c, ov);
      const float q0 = __builtin_bit_cast(float, ov.x << 16), q1 = __builtin_bit_cast(float, ov.x & 0xffff0000u);
      const float q2 = __builtin_bit_cast(float, ov.y << 16), q3 = __builtin_bit_cast(float, ov.y & 0xffff0000u);
      ssq_acc += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
    }
    if (ssq_out) {
      ssq_acc += __shfl_xor(ssq_acc, 16, 64);
      ssq_acc += __shfl_xor(ssq_acc, 32, 64);
      if (fh == 0) st4_sc1(ssq_out + (long)bx * 16 + fr, ssq_acc);
    }
  }
  chain_signal(sync, signal_idx, bx, nb_stage / KS);
}

// ---- attention stage: one (row, head) per workgroup, its four waves take every fourth trip of 16 keys ------------------------
__device__ __forceinline__ float ch_row16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

__device__ __forceinline__ void ch_rope8(float (&x)[8], const float* cs_row, int c) {
  const int ci = (c & 7) * 8;
  float cv[8], sv[8];
  load8(cs_row + ci, cv);
  load8(cs_row + CH_D / 2 + ci, sv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float other = __shfl_xor(x[j], 8, 64);
    const float r = c < 8 ? x[j] * cv[j] - other * sv[j] : x[j] * cv[j] + other * sv[j];
    x[j] = bf16_to_f32(f32_to_bf16(r));
  }
}

__device__ __forceinline__ void chain_attention(const ChainArgs& a, const ChainLayer& L, int bh, unsigned* sync, int wait_idx,
                                                int signal_idx, unsigned* err) {
  constexpr int NW = CH_KW, UN = 4;
  __shared__ float s_ml[NW][2];
  __shared__ float s_o[NW][CH_D];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int b = bh / a.nh, h = bh - b * a.nh;
  const int Nk = min(a.nk_rows[b], a.tmax);
  const int g = lane >> 4, c = lane & 15;
  const long H = a.H, ld = 3L * a.H;
  const bf16_t* kb = L.kc + (long)b * a.tmax * H + (long)h * CH_D + c * 8;
  const bf16_t* vb = L.vc + (long)b * a.tmax * H + (long)h * CH_D + c * 8;
  const int n_it = (Nk + 3) / 4;

  // rows 0 .. Nk-2 of the caches were written by earlier steps: the first CH_PRE trips are requested before the wait
  uint4 kr[CH_PRE][UN], vr[CH_PRE][UN];
#pragma unroll
  for (int n = 0; n < CH_PRE; ++n) {
    const int it0 = wave * UN + n * NW * UN;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int key = min((it0 + u) * 4 + g, Nk - 1);
      const int kc = min(key, max(Nk - 2, 0));
      kr[n][u] = *reinterpret_cast<const uint4*>(kb + (long)kc * H);
      vr[n][u] = *reinterpret_cast<const uint4*>(vb + (long)kc * H);
    }
  }
  chain_wait(sync, wait_idx, bh & (CH_SHARDS - 1), err, a.nb[0]);

  float qv[8];
  const __amdgpu_buffer_rsrc_t qr = ch_rsrc(a.qkv, (unsigned)(a.M * ld * 2));
  const unsigned qoff = (unsigned)(((long)b * ld + (long)h * CH_D + c * 8) * 2);
  auto widen8 = [](uint4 r, float (&v)[8]) {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  };
  widen8(ld16_sc1(qr, qoff), qv);
  uint4 knew_bits, vnew_bits;
  {
    const float* cs_row = a.cos_sin + (long)(Nk - 1) * CH_D;
    ch_rope8(qv, cs_row, c);
    float kn[8];
    widen8(ld16_sc1(qr, qoff + (unsigned)(2 * H)), kn);
    ch_rope8(kn, cs_row, c);
    knew_bits.x = pack_bf16x2(kn[0], kn[1]); knew_bits.y = pack_bf16x2(kn[2], kn[3]);
    knew_bits.z = pack_bf16x2(kn[4], kn[5]); knew_bits.w = pack_bf16x2(kn[6], kn[7]);
    vnew_bits = ld16_sc1(qr, qoff + (unsigned)(4 * H));
    if (g == 0 && wave == 0) {
      *reinterpret_cast<uint4*>(const_cast<bf16_t*>(kb) + (long)(Nk - 1) * H) = knew_bits;
      *reinterpret_cast<uint4*>(const_cast<bf16_t*>(vb) + (long)(Nk - 1) * H) = vnew_bits;
    }
  }
  const float sl2 = a.scale * CH_LOG2E;
#pragma unroll
  for (int j = 0; j < 8; ++j) qv[j] *= sl2;

  float m_run = -1e30f, l_run = 0.f;
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto step = [&](int key, uint4 kq4, uint4 vq4) {
    if (key >= Nk - 1) { kq4 = knew_bits; vq4 = vnew_bits; }   // (the library tests the CLAMPED key: rows past the end take the new row too, masked below)
    const unsigned kw[4] = {kq4.x, kq4.y, kq4.z, kq4.w};
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s += qv[2 * j] * __builtin_bit_cast(float, kw[j] << 16);
      s += qv[2 * j + 1] * __builtin_bit_cast(float, kw[j] & 0xffff0000u);
    }
    s = ch_row16_sum(s);
    s = key < Nk ? s : -INFINITY;
    const float m_new = fmaxf(m_run, s);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    const float pr = __builtin_amdgcn_exp2f(s - m_new);
    m_run = m_new;
    l_run = l_run * alpha + pr;
    const unsigned vw[4] = {vq4.x, vq4.y, vq4.z, vq4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[2 * j] = o[2 * j] * alpha + pr * __builtin_bit_cast(float, vw[j] << 16);
      o[2 * j + 1] = o[2 * j + 1] * alpha + pr * __builtin_bit_cast(float, vw[j] & 0xffff0000u);
    }
  };
#pragma unroll
  for (int n = 0; n < CH_PRE; ++n) {
    const int it0 = wave * UN + n * NW * UN;
    if (it0 < n_it) {
#pragma unroll
      for (int u = 0; u < UN; ++u) step((it0 + u) * 4 + g, kr[n][u], vr[n][u]);
    }
  }
  for (int it0 = wave * UN + CH_PRE * NW * UN; it0 < n_it; it0 += NW * UN) {
    uint4 k2[UN], v2[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int key = min((it0 + u) * 4 + g, Nk - 1);
      const int kc = min(key, max(Nk - 2, 0));
      k2[u] = *reinterpret_cast<const uint4*>(kb + (long)kc * H);
      v2[u] = *reinterpret_cast<const uint4*>(vb + (long)kc * H);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) step((it0 + u) * 4 + g, k2[u], v2[u]);
  }
  float m_all = fmaxf(m_run, __shfl_xor(m_run, 16, 64));
  m_all = fmaxf(m_all, __shfl_xor(m_all, 32, 64));
  const float w = __builtin_amdgcn_exp2f(m_run - m_all);
  float l = l_run * w;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = o[j] * w;
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    o[j] = x;
  }
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s_o[wave][c * 8 + j] = o[j];
    if (c == 0) { s_ml[wave][0] = m_all; s_ml[wave][1] = l; }
  }
  __syncthreads();
  if (wave != 0) return;
  if (g == 0) {
    float Mx = s_ml[0][0];
#pragma unroll
    for (int w4 = 1; w4 < NW; ++w4) Mx = fmaxf(Mx, s_ml[w4][0]);
    float Ls = 0.f, accv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w4 = 0; w4 < NW; ++w4) {
      const float f = __builtin_amdgcn_exp2f(s_ml[w4][0] - Mx);
      Ls += s_ml[w4][1] * f;
#pragma unroll
      for (int j = 0; j < 8; ++j) accv[j] += s_o[w4][c * 8 + j] * f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) accv[j] /= Ls;
    bf16_t* op = a.att + (long)b * H + (long)h * CH_D + c * 8;
    st8_sc1(op, make_uint2(pack_bf16x2(accv[0], accv[1]), pack_bf16x2(accv[2], accv[3])));
    st8_sc1(op + 4, make_uint2(pack_bf16x2(accv[4], accv[5]), pack_bf16x2(accv[6], accv[7])));
  }
  chain_signal(sync, signal_idx, bh, a.nb[1]);
}

// Zeroes the arrival counters and the tickets WRITE-THROUGH (agent-scope stores): the chained launch reads and updates these words
// with agent-scope atomics, i.e. at the memory side. (Round 6: with hipMemsetAsync in front of the launch the chained step was
// bit-repeatable eagerly and NOT inside a hipGraph replay — every replay gave another result while the same kernel launched stage by
// stage stayed exact: the memset node's zeros were not where the atomics looked yet, waits saw the previous replay's counts and
// let stages start early. tools/chain_stress.py)
__global__ __launch_bounds__(256) void chain_zero_kernel(unsigned* a, long na, unsigned* b, long nb) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < na) __hip_atomic_store(a + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (i - na < nb) __hip_atomic_store(b + (i - na), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64 * CH_KW) void decode_chain_kernel(ChainArgs a) {
  const int bid = (int)blockIdx.x + a.block0;
  const int layer = bid / a.per_layer;
  int r = bid - layer * a.per_layer;
  int stage = 0;
  while (stage < CH_STAGES - 1 && r >= a.nb[stage]) { r -= a.nb[stage]; ++stage; }
  const ChainLayer& L = a.L[layer];
#ifdef CH_PLACE
  if (layer == 2 && threadIdx.x == 0 && r < 1024) {
    ch_place_buf[stage][r] = ((unsigned)__builtin_amdgcn_s_getreg(6164) << 16) | ((unsigned)__builtin_amdgcn_s_getreg(63492) & 0xffffu);
    ch_place_t[stage][r] = wall_clock64();
  }
#endif
  unsigned* err = a.sync + (long)a.n_layers * CH_STAGES * CH_STAGE_WORDS;
  const int me = layer * CH_STAGES + stage;
  const int dep = me - 1;   // stage 0 of layer l waits for stage 4 of layer l-1; (0, 0): dep = -1, no wait
  const int H = a.H, F = a.F, M = a.M, parts = H / 16;
  switch (stage) {
    case 0:
      chain_product<1, false>(L.wqkv, 3 * H, H, a.x, H, M, a.qkv, 3L * H, nullptr, 0, layer > 0 ? a.ssq_b : nullptr, parts, a.eps,
                              layer == 0 ? a.stats0 : nullptr, nullptr, r, a.nb[0], a.nb[4] / 2, a.sync, dep, me, err);
      break;
    case 1:
      chain_attention(a, L, r, a.sync, dep, me, err);
      break;
    case 2:
      chain_product<1, false, 2>(L.wo, H, H, a.att, H, M, a.x, H, a.x, H, nullptr, 0, a.eps, nullptr, a.ssq_a, r, a.nb[2], a.nb[1], a.sync, dep, me,
                                 err, a.ws, err + CH_LINE + (long)(layer * 2) * parts);
      break;
    case 3:
      chain_product<2, true>(L.wgu, 2 * F, H, a.x, H, M, a.g, F, nullptr, 0, a.ssq_a, parts, a.eps, nullptr, nullptr, r, a.nb[3], a.nb[2] / 2, a.sync,
                             dep, me, err);
      break;
    default:
      chain_product<1, false, 2>(L.wd, H, F, a.g, F, M, a.x, H, a.x, H, nullptr, 0, a.eps, nullptr, a.ssq_b, r, a.nb[4], a.nb[3], a.sync, dep, me,
                                 err, a.ws, err + CH_LINE + (long)(layer * 2 + 1) * parts);
      break;
  }
}

}  // namespace

// One KV-cached decode step of n_layers Llama layers at M <= 8 rows in ONE launch (see the head of this file).
//   layers: HOST array of n_layers {wqkv, wo, wgu, wd, kcache, vcache} DEVICE pointers (passed to the kernel by value):
//           wqkv [3H][H] with input_layernorm's gamma folded into its columns, wo [H][H], wgu [2F][H] in 16-row [gate | up] groups with
//           post_attention_layernorm's gamma folded in, wd [H][F] — the operands haff_gemm_bf16_rms takes on the 5-launch path;
//           kcache / vcache [M][tmax][H] bf16.
//   x [M][H] bf16: the residual stream, read and written in place; stats0 f32 [M][2]: {mean, rstd} of its rows on entry
//   (haff_row_stats, rms); qkv [M][3H], att [M][H], g [M][F] bf16 and ssq_a / ssq_b f32 [H/16][16]: scratch;
//   cos_sin f32 [tmax][128]; nk_rows device int32 [M] (position of the new token + 1);
//   sync: DEVICE uint32 [haff_decode_chain_sync_words(n_layers)], zeroed once by the caller: the launch re-zeroes the counters, the
//   last word is a sticky error flag (a wait ran out of patience: haff_decode_chain_status).
//   per_stage_launches != 0: the same kernel as n_layers * 5 launches, one per stage (tests / A-B: identical arithmetic, no chaining).
// hidden % 256 == 0, ffn % 256 == 0, hidden == heads * 128, hidden / 16 <= 512, M <= 8, n_layers <= 48; 16-B aligned pointers.
struct haff_chain_layer { const void *wqkv, *wo, *wgu, *wd; void *kcache, *vcache; };

extern "C" int haff_decode_chain_bf16(const haff_chain_layer* layers, int n_layers, int M, int hidden, int ffn, int heads, void* x,
                                      void* qkv, void* att, void* g, float* ssq_a, float* ssq_b, float* ws, const float* stats0, float eps,
                                      const float* cos_sin, const int* nk_rows, int tmax, float scale, unsigned* sync,
                                      int per_stage_launches, void* stream) {
  if (!layers || n_layers <= 0 || n_layers > CH_MAXL || M <= 0 || M > 8) return HAFF_ERR_BAD_ARG;
  if (hidden <= 0 || ffn <= 0 || (hidden % 256) || (ffn % 256) || hidden != heads * CH_D || hidden / 16 > 512) return HAFF_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 15)) return HAFF_ERR_BAD_ARG;
  if (!x || !qkv || !att || !g || !ssq_a || !ssq_b || !stats0 || !cos_sin || !nk_rows || !sync || tmax <= 0) return HAFF_ERR_BAD_ARG;
  const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(att) |
                       reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(ssq_a) | reinterpret_cast<uintptr_t>(ssq_b);
  if (al & 15) return HAFF_ERR_BAD_ARG;
  ChainArgs a;
  for (int i = 0; i < n_layers; ++i) {
    const haff_chain_layer& s = layers[i];
    if (!s.wqkv || !s.wo || !s.wgu || !s.wd || !s.kcache || !s.vcache) return HAFF_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(s.wqkv) | reinterpret_cast<uintptr_t>(s.wo) | reinterpret_cast<uintptr_t>(s.wgu) |
         reinterpret_cast<uintptr_t>(s.wd) | reinterpret_cast<uintptr_t>(s.kcache) | reinterpret_cast<uintptr_t>(s.vcache)) & 15)
      return HAFF_ERR_BAD_ARG;
    a.L[i] = ChainLayer{reinterpret_cast<const bf16_t*>(s.wqkv), reinterpret_cast<const bf16_t*>(s.wo), reinterpret_cast<const bf16_t*>(s.wgu),
                        reinterpret_cast<const bf16_t*>(s.wd), reinterpret_cast<bf16_t*>(s.kcache), reinterpret_cast<bf16_t*>(s.vcache)};
  }
  for (int i = n_layers; i < CH_MAXL; ++i) a.L[i] = ChainLayer{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a.x = reinterpret_cast<bf16_t*>(x); a.qkv = reinterpret_cast<bf16_t*>(qkv); a.att = reinterpret_cast<bf16_t*>(att);
  a.g = reinterpret_cast<bf16_t*>(g);
  a.ssq_a = ssq_a; a.ssq_b = ssq_b; a.ws = ws; a.stats0 = stats0; a.cos_sin = cos_sin; a.nk_rows = nk_rows; a.sync = sync;
  a.n_layers = n_layers; a.M = M; a.H = hidden; a.F = ffn; a.nh = heads; a.tmax = tmax; a.eps = eps; a.scale = scale;
  a.nb[0] = 3 * hidden / 16; a.nb[1] = M * heads; a.nb[2] = 2 * (hidden / 16); a.nb[3] = 2 * ffn / 32; a.nb[4] = 2 * (hidden / 16);
  a.per_layer = a.nb[0] + a.nb[1] + a.nb[2] + a.nb[3] + a.nb[4];
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // arrival counters, then (behind the sticky error line, which stays) the tickets of the K-split tiles
  {
    const long na = (long)n_layers * CH_STAGES * CH_STAGE_WORDS, nt = (long)n_layers * 2 * (hidden / 16);
    hipLaunchKernelGGL(chain_zero_kernel, dim3((unsigned)((na + nt + 255) / 256)), dim3(256), 0, s, sync, na, sync + na + CH_LINE, nt);
  }
  a.block0 = 0;
  if (!per_stage_launches) {
    hipLaunchKernelGGL(decode_chain_kernel, dim3((unsigned)(n_layers * a.per_layer)), dim3(64 * CH_KW), 0, s, a);
    return haff_check_launch();
  }
  // the SAME kernel, one launch per (layer, stage): every wait is already satisfied when its workgroup starts (stream order), the
  // arithmetic is the chained launch's statement for statement — what the chained launch is tested against bit for bit
  for (int l = 0; l < n_layers; ++l)
    for (int st = 0; st < CH_STAGES; ++st) {
      hipLaunchKernelGGL(decode_chain_kernel, dim3((unsigned)a.nb[st]), dim3(64 * CH_KW), 0, s, a);
      a.block0 += a.nb[st];
    }
  return haff_check_launch();
}

// 0: every wait of the launches so far was satisfied; 1: one ran out (the sticky word behind the counters). Synchronises the stream.
extern "C" int haff_decode_chain_status(const unsigned* sync, int n_layers, void* stream) {
  if (!sync || n_layers <= 0 || n_layers > CH_MAXL) return HAFF_ERR_BAD_ARG;
  unsigned v = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(&v, sync + (long)n_layers * CH_STAGES * CH_STAGE_WORDS, sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess)
    return HAFF_ERR_LAUNCH;
  if (hipStreamSynchronize(s) != hipSuccess) return HAFF_ERR_LAUNCH;
  return v ? 1 : 0;
}

// uint32 words of the sync buffer: 16 counter lines per (layer, stage), the sticky error word (a line of its own), one ticket per K-split tile
extern "C" int haff_decode_chain_sync_words(int n_layers, int hidden) {
  if (n_layers <= 0 || n_layers > CH_MAXL || hidden <= 0) return 0;
  return n_layers * CH_STAGES * CH_STAGE_WORDS + CH_LINE + n_layers * 2 * (hidden / 16);
}

// Workgroups per layer the chained launch uses (for sizing / reporting): 0 when the geometry is unsupported.
extern "C" int haff_decode_chain_supported(int M, int hidden, int ffn, int heads, int n_layers) {
  if (n_layers <= 0 || n_layers > CH_MAXL || M <= 0 || M > 8) return 0;
  if (hidden <= 0 || ffn <= 0 || (hidden % 256) || (ffn % 256) || hidden != heads * CH_D || hidden / 16 > 512) return 0;
  return 3 * hidden / 16 + M * heads + 2 * (hidden / 16) + 2 * ffn / 32 + 2 * (hidden / 16);
}
