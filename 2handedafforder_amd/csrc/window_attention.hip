// Fused SAM window attention for MI355X (gfx950): decomposed rel-pos bias computed IN the kernel, one pass over HBM.
//
//   out = softmax(scale * q.k^T + rel_h[q, kh] + rel_w[q, kw]) @ v      per (window, head), S x S tokens
//   rel_h[q, kh] = q . Rh[qh - kh + S - 1],  rel_w[q, kw] = q . Rw[qw - kw + S - 1]   (UNSCALED q)
//   (2Haff/model/segment_anything/modeling/image_encoder.py:235-260 Attention.forward, :354-392
//    add_decomposed_rel_pos, :322-351 get_rel_pos; the 28 windowed blocks of ViT-H: S = 14, d = 80)
//
// Why a dedicated kernel: at S*S = 196 tokens the generic flash kernel (attention.hip, 128-query blocks, rel-pos
// tables materialised by a separate kernel) reads K/V twice, pads 196 -> 256 on both axes and round-trips 112 B of
// fp32 tables per (query, head). This path is HBM-bound (98 flop/B), so the design goal is: q, k, v read once, out
// written once, nothing else touches HBM.
//
// An item is one (window, head). Its K and V are staged once into LDS (K rows 160 B since round 4 — see WinCfg::KSTR;
// V rows unpadded 160 B, which is conflict-free for ds_read_b64_tr_b16: 8 consecutive rows x 40 dwords
// hit 8 disjoint 8-bank groups). 7 waves share an item, each taking query-grid rows qh = wave and wave+7: a q-tile is
// ONE grid row (14 real queries + 2 masked lanes), keys are visited as 14 tiles = 14 grid rows of 16 virtual columns
// (kw >= 14 masked), so
//   * rel_w[q][kw] is tile-invariant (4 registers per lane) and rel_h[q][kh] is one scalar per key tile;
//   * both come from two small MFMAs T^T = Table . Q^T (27 table rows -> 2 m-tiles) whose result takes one trip
//     through a 2.3 KB wave-private LDS scratch to reach the lanes that need it;
//   * the whole 14 x 16 score row block of a q-tile lives in registers: plain (not online) softmax.
// Score MFMA is issued swapped (S^T = K . Q^T) as in attention.hip: lane = (query fr, 4 keys 4fh..4fh+3), P^T feeds
// the P.V MFMA from registers and V^T fragments come from transposed LDS reads with the same permuted k order.
#include "haff_common.h"

// Ablation hooks (counter / timing experiments: WRONG results) exist only in builds made with -DHAFF_TUNING
// (tools/build_window_variant.sh): HAFF_WIN_NOKREAD / _NOVREAD leave out the K / V^T fragment reads from LDS, HAFF_WIN_NOSCR the
// trips of the rel-pos terms through the wave-private scratch, HAFF_WIN_NODMA the HBM -> LDS staging.
#ifndef HAFF_TUNING
#undef HAFF_WIN_NOKREAD
#undef HAFF_WIN_NOVREAD
#undef HAFF_WIN_NOSCR
#undef HAFF_WIN_NODMA
#undef HAFF_WIN_RS
#undef HAFF_WIN_KLAST_OLD
#endif

namespace {

struct WinArgs {
  const bf16_t *q, *k, *v;
  bf16_t* o;
  long q_sb, q_sh, q_st;
  long k_sb, k_sh, k_st;
  long v_sb, v_sh, v_st;
  long o_sb, o_sh, o_st;
  int B, H;  // B = number of windows
  float scale;
  const bf16_t *tab_h, *tab_w;  // [2S-1][D] bf16, contiguous
  // window padding (image_encoder.py:263-288): windows tile a grid_h x grid_w token grid, wps per side; window
  // tokens beyond the grid are pads whose q/k/v are those of token row `pad_token` (relative to the base pointers).
  // grid_h == 0: every token is real.
  int grid_h, grid_w, wps;
  long pad_token;
};

#ifdef HAFF_WIN_TRACE  // phase timestamps (100 MHz wall clock) of workgroup 0 / wave 0, for tools/window_attn_trace.py
__device__ unsigned long long haff_win_trace_buf[64 * 16];
#define WTRACE(i) do { if (blockIdx.x == 0 && tid == 0 && it < 64) haff_win_trace_buf[it * 16 + (i)] = wall_clock64(); } while (0)
#else
#define WTRACE(i) do {} while (0)
#endif

constexpr float WLOG2E = 1.4426950408889634f;
typedef __attribute__((address_space(3))) bf16x4* wlds_v4_ptr;
typedef const __attribute__((address_space(1))) void* wgptr_t;
typedef __attribute__((address_space(3))) void* wlptr_t;
__device__ __attribute__((aligned(16))) unsigned int haff_win_zero_page[4];  // source of the K-row pad slot

template <int D, int S>
struct WinCfg {
  static constexpr int N = S * S;
  static constexpr int NKD = (D + 31) / 32;     // 32-deep k-steps over the head dim (last one zero-padded in Q)
  static constexpr int ND = D / 16;             // output d-tiles
  static constexpr int CPR = D / 8;             // 16-B chunks per row
  // K rows (round 4): the fragment read — lane (key column fr, d-chunk fh): 16 B at row fr, column 16 fh — runs at the LDS's
  // full rate exactly for row strides of 32 modulo 64 bytes (tools/probes/lds_b128_pattern.hip: 160, 224, 288 -> 18 clocks per
  // read and wave, 176 -> 32). d = 80 needs no pad chunk at all: the 176-byte rows of rounds 1-3 made every K read a 2-way bank
  // conflict. KPC pad chunks (0 for d = 80, 2 otherwise) re-read chunk 0 of their row: finite, multiplied by zero Q columns. The
  // padded third k-step of d = 80 reads the first chunks of the NEXT row (or of V behind the last K row) there instead: finite
  // data against the same zero Q columns.
  static constexpr int KPC = (D * 2) % 64 == 32 ? 0 : 2;
  static constexpr int KSTR = D * 2 + 16 * KPC;   // bytes
  static constexpr int VSTR = D * 2;
  static constexpr int NKS = (S + 1) / 2;       // P.V k-steps: two key tiles (grid rows) each
#ifdef HAFF_WIN_RS
  static constexpr int RS = HAFF_WIN_RS;
#else
  // scratch row stride in floats: 52 since round 6 (36 before). The scratch's 16-B writes at a 144-B row stride and the rel_w
  // reads behind them were half of the kernel's LDS bank-conflict cycles (tools/pmc_window_lds.sh: 6.45e6 of 12.5e6 per 32-frame
  // launch; 40 -> 19.8e6, 44 -> 6.6e6, 52 -> 6.3e6 in all, 68 = 36: profiles/r6_pmc_window_attn_lds_ablation.txt)
  static constexpr int RS = 52;
#endif
  static constexpr int QPW = 2;                 // q-tiles (query-grid rows) per wave per item
  static constexpr int NWAVES = (S + QPW - 1) / QPW;   // 7 waves cover the 14 grid rows exactly
  static constexpr int NTHREADS = 64 * NWAVES;
  static constexpr int KSLOTS = N * (CPR + KPC), VSLOTS = N * CPR;   // 16-B DMA slots
  static constexpr int K_BYTES = N * KSTR, V_BYTES = N * VSTR;
  static constexpr int PAD_SLOTS = (KSLOTS + VSLOTS + 63) / 64 * 64;   // whole DMA instructions; the tail slots hold zeros
  static constexpr int BUF_BYTES = PAD_SLOTS * 16;
  static constexpr int SC_BYTES = NWAVES * 16 * RS * 4;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES + SC_BYTES;
};

// Persistent: one workgroup of 7 waves per CU walks items (window, head) = blockIdx.x, += gridDim.x. While the waves
// compute item i out of LDS buffer i&1, the K/V rows and the Q fragments of item i+1 are already in flight from HBM
// into registers; they are written to the other buffer after the compute, then ONE barrier per item. The load stream
// never stops, which is what an HBM-bound kernel needs (the non-persistent version serialised load -> compute per
// workgroup and reached 2.3 TB/s; see tools/window_attn_bench.py).
template <int D, int S>
__global__ __launch_bounds__((WinCfg<D, S>::NTHREADS)) void window_attn_kernel(WinArgs p) {
  using C = WinCfg<D, S>;
  static_assert(D % 16 == 0 && D <= 96 && S <= 16 && 2 * S - 1 <= 32, "window kernel geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  float* sScr = reinterpret_cast<float*>(wsm + 2 * C::BUF_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fh = lane >> 4;
  const int n_items = p.B * p.H;

  // item -> (window, head). Ids equal mod 8 share an XCD (gridDim.x is a multiple of 8 or covers every item): give
  // one XCD all heads of a window, so the partial 128-B lines a head's 160-B row slices leave behind are consumed
  // from that XCD's L2 by its neighbours.
  auto decode = [&](int id, int& win, int& h) {
    if ((p.B & 7) == 0) {
      const int x = id & 7, kseq = id >> 3;
      win = (kseq / p.H) * 8 + x;
      h = kseq - (kseq / p.H) * p.H;
    } else {
      win = id / p.H;
      h = id - win * p.H;
    }
  };

  // Staging is HBM -> LDS DMA (global_load_lds_dwordx4: no VGPR round trip, nothing to write back after the
  // compute): the LDS image is slot-linear (lane i of an instruction lands at base + 16*i), each lane picks the source
  // chunk of its slot. K rows are CPR + KPC slots, V rows CPR slots.
  constexpr int TOTAL_SLOTS = C::KSLOTS + C::VSLOTS;
  constexpr int NDMA = (C::PAD_SLOTS + C::NTHREADS - 1) / C::NTHREADS;
  static_assert((C::K_BYTES % 16) == 0 && C::KSTR == (C::CPR + C::KPC) * 16 && C::VSTR == C::CPR * 16 && C::KSTR % 64 == 32, "slot layout");
  // byte offset of each slot's source relative to the item's K view (V = K + a constant for the fused qkv layout the
  // caller passes; checked on the host). Pad / tail slots re-read chunk 0 of a K row: they only have to be finite
  // (the matching Q columns are zero).
  unsigned slot_off[NDMA];
  unsigned slot_pad[NDMA];   // [31:28] token row in window, [27:24] token column, [23] V slot, [7:0] byte offset of the chunk inside the pad token's K / V row
  const long v_minus_k = (p.v - p.k);   // elements; same for every (window, head) because strides match
  const unsigned vmk_bytes = (unsigned)(v_minus_k * 2);   // (< 4 GiB: checked on the host)
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int sl = tid + i * C::NTHREADS;
    long off;
    unsigned pad = 0;
    if (sl >= TOTAL_SLOTS) {
      off = 0;
    } else if (sl < C::KSLOTS) {
      const int row = sl / (C::CPR + C::KPC), c = sl - row * (C::CPR + C::KPC);
      off = (long)row * p.k_st + (c < C::CPR ? c * 8 : 0);
      pad = ((unsigned)(row / S) << 28) | ((unsigned)(row % S) << 24) | (unsigned)((c < C::CPR ? c : 0) * 16);
    } else {
      const int sv = sl - C::KSLOTS;
      const int row = sv / C::CPR, c = sv - row * C::CPR;
      off = v_minus_k + (long)row * p.v_st + c * 8;
      pad = ((unsigned)(row / S) << 28) | ((unsigned)(row % S) << 24) | (1u << 23) | (unsigned)(c * 16);   // (+ v_minus_k at issue time: round 5's
      //                                                  head-major planes put V hundreds of MB behind K, past any packed field)
    }
    slot_off[i] = (unsigned)(off * 2);
    slot_pad[i] = pad;
  }
  uint4 qnext[C::QPW][C::NKD];
  const int qcol = min(fr, S - 1);       // the lane's query column inside a grid row (masked lanes duplicate S-1)
  // state of the prefetch in flight (the next item's K/V view), set by begin_loads, used by issue_dma
  const bf16_t* nx_kb = p.k;
  int nx_pad_h0 = S, nx_pad_w0 = S;
  unsigned nx_pad_base = 0, nx_lds0 = 0;
  auto begin_loads = [&](int id, int buf) {
    int win, h;
    decode(id, win, h);
    const bf16_t* qb = p.q + (long)win * p.q_sb + (long)h * p.q_sh;
    const bf16_t* kb = p.k + (long)win * p.k_sb + (long)h * p.k_sh;
    // first padded token row / column of this window (>= S: none)
    int pad_h0 = S, pad_w0 = S;
    if (p.grid_h > 0) {
      const int wi = win % (p.wps * p.wps);
      pad_h0 = p.grid_h - (wi / p.wps) * S;
      pad_w0 = p.grid_w - (wi % p.wps) * S;
    }
    const unsigned pad_base = (unsigned)((p.pad_token * p.k_st - (long)win * p.k_sb) * 2);   // bytes from kb
    const bf16_t* qpad = p.q + p.pad_token * p.q_st + (long)h * p.q_sh;
#pragma unroll
    for (int t = 0; t < C::QPW; ++t) {
      const int qh_t = min(wave + t * C::NWAVES, S - 1);
      const bf16_t* qsrc = (qh_t >= pad_h0 || qcol >= pad_w0) ? qpad : qb + (long)(qh_t * S + qcol) * p.q_st;
#pragma unroll
      for (int kd = 0; kd < C::NKD; ++kd) {
        const int col = kd * 32 + fh * 8;
        qnext[t][kd] = make_uint4(0, 0, 0, 0);
        if (col < D) qnext[t][kd] = *reinterpret_cast<const uint4*>(qsrc + col);
      }
    }
    nx_kb = kb;
    nx_pad_h0 = pad_h0;
    nx_pad_w0 = pad_w0;
    nx_pad_base = pad_base;
    nx_lds0 = (unsigned)(uintptr_t)(wlptr_t)(wsm + buf * C::BUF_BYTES) + wave * 1024;  // hardware adds lane*16
  };
  // DMA instructions [i0, i1) of the item begin_loads prepared. Inline asm, not the builtin: the compiler's LDS-DMA
  // tracking puts a vmcnt(0) in front of the next LDS access (it cannot prove the scratch / other buffer do not alias),
  // which would drain the prefetch at once. Completion is waited for by hand (vmcnt(0) before the end-of-item barrier).
  // M0 is otherwise unused here. The 10 instructions of an item are NOT issued in one go: the CU accepts requests at the
  // rate HBM serves them (~24 GB/s per CU), so a wave that issues its whole 9 KB share sits in the issue stage for
  // ~4 us of an ~11 us item (tools/window_attn_trace.py); spread over the compute phases the requests queue behind
  // the MFMA / softmax work instead.
  auto issue_dma = [&](int i0, int i1) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      if (i < i0 || i >= i1) continue;
      const unsigned m0v = nx_lds0 + i * C::NTHREADS * 16;
      const bool is_pad = (int)(slot_pad[i] >> 28) >= nx_pad_h0 || (int)((slot_pad[i] >> 24) & 15u) >= nx_pad_w0;
      const unsigned off = is_pad ? nx_pad_base + (slot_pad[i] & 0xffu) + ((slot_pad[i] >> 23) & 1u) * vmk_bytes : slot_off[i];
#ifdef HAFF_WIN_NODMA
      asm volatile("" ::"v"(off), "s"(m0v));
      continue;
#endif
      if ((i + 1) * C::NTHREADS <= C::PAD_SLOTS || (i * C::NTHREADS + wave * 64) < C::PAD_SLOTS)   // wave-uniform
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(nx_kb), "s"(m0v)
                     : "memory");
    }
  };
  constexpr int DMA_Q1 = (NDMA + 3) / 4, DMA_Q2 = (NDMA + 1) / 2, DMA_Q3 = (3 * NDMA + 3) / 4;

  // The DMA is invisible to the compiler's vmcnt bookkeeping, so its own counted waits must never be needed while a
  // prefetch is in flight: after each hand-placed vmcnt(0) the prefetched Q registers are passed through an empty
  // asm, which makes the compiler consider them loaded from then on (no later, miscounted wait on them).
  auto settle_q = [&]() {
#pragma unroll
    for (int t = 0; t < C::QPW; ++t)
#pragma unroll
      for (int kd = 0; kd < C::NKD; ++kd)
        asm volatile("" : "+v"(qnext[t][kd].x), "+v"(qnext[t][kd].y), "+v"(qnext[t][kd].z), "+v"(qnext[t][kd].w));
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  begin_loads(item, 0);
  issue_dma(0, NDMA);

  // ---- rel-pos table fragments (A operand: lane = (table row fr, d-chunk fh)); constant for the workgroup ----
  bf16x8 th[2][C::NKD], tw[2][C::NKD];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int r = min(mt * 16 + fr, 2 * S - 2);
#pragma unroll
    for (int kd = 0; kd < C::NKD; ++kd) {
      const int col = kd * 32 + fh * 8;
      uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
      if (col < D) {
        a = *reinterpret_cast<const uint4*>(p.tab_h + r * D + col);
        b = *reinterpret_cast<const uint4*>(p.tab_w + r * D + col);
      }
      th[mt][kd] = __builtin_bit_cast(bf16x8, a);
      tw[mt][kd] = __builtin_bit_cast(bf16x8, b);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  settle_q();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int kd = 0; kd < C::NKD; ++kd) {
      asm volatile("" : "+v"(th[mt][kd]));
      asm volatile("" : "+v"(tw[mt][kd]));
    }
  __syncthreads();

  const float sl2 = p.scale * WLOG2E;
  float* scr = sScr + wave * (16 * C::RS);
  const int tr_q = fr >> 2, tr_p = fr & 3;
  bool kw_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) kw_ok[j] = (4 * fh + j) < S;

  for (int it = 0;; ++it) {
    const int buf = it & 1;
    const unsigned char* sK = wsm + buf * C::BUF_BYTES;
    const unsigned char* sV = sK + C::K_BYTES;
    int win, h;
    decode(item, win, h);
    bf16_t* ob = p.o + (long)win * p.o_sb + (long)h * p.o_sh;
    // this item's Q fragments arrived with the previous prefetch; move them out of the prefetch registers
    uint4 qcur[C::QPW][C::NKD];
#pragma unroll
    for (int t = 0; t < C::QPW; ++t)
#pragma unroll
      for (int kd = 0; kd < C::NKD; ++kd) qcur[t][kd] = qnext[t][kd];
    const int next = item + gridDim.x;
    const bool has_next = next < n_items;
    WTRACE(0);
    if (has_next) {   // HBM -> the other LDS buffer (last read one item ago), in flight during the compute
      begin_loads(next, buf ^ 1);
      issue_dma(0, DMA_Q1);
    }
    WTRACE(1);

#pragma unroll
    for (int t = 0; t < C::QPW; ++t) {
      const int qh = wave + t * C::NWAVES;   // < S: the waves tile the grid rows exactly (static_assert below)
      static_assert(C::QPW * C::NWAVES == S, "q-tiles must divide evenly over the waves");
      // ---- Q fragments of this grid row: raw (for the rel-pos terms) and pre-scaled by scale*log2e ----
      bf16x8 qraw[C::NKD], qs[C::NKD];
#pragma unroll
      for (int kd = 0; kd < C::NKD; ++kd) {
        const uint4 r = qcur[t][kd];
        uint4 rs = make_uint4(0, 0, 0, 0);
        if (kd * 32 + fh * 8 < D) {
          float qv[8];
          const unsigned w4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            qv[2 * j] = __builtin_bit_cast(float, w4[j] << 16) * sl2;
            qv[2 * j + 1] = __builtin_bit_cast(float, w4[j] & 0xffff0000u) * sl2;
          }
          rs.x = pack_bf16x2(qv[0], qv[1]); rs.y = pack_bf16x2(qv[2], qv[3]);
          rs.z = pack_bf16x2(qv[4], qv[5]); rs.w = pack_bf16x2(qv[6], qv[7]);
        }
        qraw[kd] = __builtin_bit_cast(bf16x8, r);
        qs[kd] = __builtin_bit_cast(bf16x8, rs);
      }

      // ---- rel_w: T^T[r][q] = Rw[r] . q -> scratch[q][r] -> lane (q, fh) picks r = qw - kw + S-1, kw = 4fh+j ----
      float relw4[4], relh_t[S];
      {
        f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = t0;
#pragma unroll
        for (int kd = 0; kd < C::NKD; ++kd) {
          t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tw[0][kd], qraw[kd], t0, 0, 0, 0);
          t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tw[1][kd], qraw[kd], t1, 0, 0, 0);
        }
        float a[4] = {t0[0], t0[1], t0[2], t0[3]}, b[4] = {t1[0], t1[1], t1[2], t1[3]};
#ifdef HAFF_WIN_NOSCR
#pragma unroll
        for (int j = 0; j < 4; ++j) relw4[j] = kw_ok[j] ? (a[j] + b[j]) * WLOG2E : -INFINITY;
#else
        store4(scr + fr * C::RS + 4 * fh, a);
        store4(scr + fr * C::RS + 16 + 4 * fh, b);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = min(max(qcol - (4 * fh + j) + S - 1, 0), 2 * S - 2);
          relw4[j] = kw_ok[j] ? scr[fr * C::RS + r] * WLOG2E : -INFINITY;   // masked key columns stay -inf through the MFMAs
        }
        __builtin_amdgcn_wave_barrier();
#endif
      }
      // ---- rel_h: same through the same scratch; r = qh - kh + S-1 is wave-uniform per key tile ----
      {
        f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = t0;
#pragma unroll
        for (int kd = 0; kd < C::NKD; ++kd) {
          t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th[0][kd], qraw[kd], t0, 0, 0, 0);
          t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th[1][kd], qraw[kd], t1, 0, 0, 0);
        }
        float a[4] = {t0[0], t0[1], t0[2], t0[3]}, b[4] = {t1[0], t1[1], t1[2], t1[3]};
#ifdef HAFF_WIN_NOSCR
#pragma unroll
        for (int kt = 0; kt < S; ++kt) relh_t[kt] = (a[kt & 3] + b[kt & 3]) * WLOG2E;
#else
        store4(scr + fr * C::RS + 4 * fh, a);
        store4(scr + fr * C::RS + 16 + 4 * fh, b);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kt = 0; kt < S; ++kt) relh_t[kt] = scr[fr * C::RS + (qh - kt + S - 1)] * WLOG2E;
        __builtin_amdgcn_wave_barrier();
#endif
      }

      WTRACE(2 + 5 * t);
      // ---- scores S^T[key][q] for all S key tiles, accumulators start at the (log2-domain) bias ----
      f32x4 sacc[S];
#pragma unroll
      for (int kt = 0; kt < S; ++kt)
        sacc[kt] = f32x4{relw4[0] + relh_t[kt], relw4[1] + relh_t[kt], relw4[2] + relh_t[kt], relw4[3] + relh_t[kt]};
      // groups of 7 key tiles, k-step outer inside a group: 7 independent accumulators between two uses of the
      // same one, 7 K fragments in flight
      constexpr int KG = (S + 1) / 2;
      // the zero-padded last k-step (d = 80: columns 64..95) would read 32 bytes PAST an unpadded row for fh >= 2 — the next key's
      // first chunks, or V behind the last K row: finite only if the neighbour is (Inf / NaN x a zero Q column = NaN in ANOTHER
      // key's score). Those lanes re-read the row's own last chunk instead (the global kernel's k_lane2 does the same).
#ifdef HAFF_WIN_KLAST_OLD
      const int koff_last = min(fh * 16 + (C::NKD - 1) * 64, D * 2 - 16);
#else
      // (round 6) ... and the two lanes groups past the row's end (fh >= 2, zero Q columns) take the two chunks IN FRONT of the last
      // k-step's, so that the four lanes of a key still cover 64 contiguous bytes of its row (rotated): the same bank pattern as the
      // full k-steps. Three lanes on one chunk (rounds 4-5) made this read a 2-way bank conflict (tools/pmc_window_lds.sh).
      const int koff_last = (D * 2 - (C::NKD - 1) * 64 == 32) ? (C::NKD - 1) * 64 - 32 + ((fh + 2) & 3) * 16
                                                              : min(fh * 16 + (C::NKD - 1) * 64, D * 2 - 16);
#endif
#pragma unroll
      for (int g0 = 0; g0 < S; g0 += KG) {
#pragma unroll
        for (int kd = 0; kd < C::NKD; ++kd) {
#pragma unroll
          for (int kt = g0; kt < g0 + KG && kt < S; ++kt) {
            // A operand row = key (kh = kt, kw = fr)
#ifdef HAFF_WIN_NOKREAD
            const bf16x8 kf = qs[(kd + 1) % C::NKD];
#else
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (kt * S + qcol) * C::KSTR + (kd == C::NKD - 1 ? koff_last : fh * 16 + kd * 64));
#endif
            sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qs[kd], sacc[kt], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }

      WTRACE(3 + 5 * t);
      if (has_next) {
        if (t == 0) issue_dma(DMA_Q1, DMA_Q2);
        else issue_dma(DMA_Q3, NDMA);
      }
      // ---- softmax over the whole row block (lane owns query column fr; its 4 keys per tile are kw = 4fh+j) ----
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < S; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, sacc[kt][j]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < S; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = __builtin_amdgcn_exp2f(sacc[kt][j] - mx);
          sacc[kt][j] = e;
          psum += e;
        }
      psum += __shfl_xor(psum, 16, 64);
      psum += __shfl_xor(psum, 32, 64);

      WTRACE(4 + 5 * t);
      // ---- O^T += V^T . P^T: k-step ks covers key tiles 2ks (slots 0-3) and 2ks+1 (slots 4-7) ----
      f32x4 oacc[C::ND];
#pragma unroll
      for (int dt = 0; dt < C::ND; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int vkw = min(4 * fh + tr_q, S - 1);  // key column this lane's transposed read fetches
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) {
        constexpr int dummy = 0; (void)dummy;
        const int t0 = 2 * ks, t1 = (2 * ks + 1 < S) ? 2 * ks + 1 : 2 * ks;
        uint4 u;
        u.x = pack_bf16x2(sacc[t0][0], sacc[t0][1]);
        u.y = pack_bf16x2(sacc[t0][2], sacc[t0][3]);
        u.z = (2 * ks + 1 < S) ? pack_bf16x2(sacc[t1][0], sacc[t1][1]) : 0u;
        u.w = (2 * ks + 1 < S) ? pack_bf16x2(sacc[t1][2], sacc[t1][3]) : 0u;
        const bf16x8 pf = __builtin_bit_cast(bf16x8, u);
        const unsigned char* v0 = sV + (t0 * S + vkw) * C::VSTR + 8 * tr_p;
        const unsigned char* v1 = sV + (t1 * S + vkw) * C::VSTR + 8 * tr_p;
#pragma unroll
        for (int dt = 0; dt < C::ND; ++dt) {
#ifdef HAFF_WIN_NOVREAD
          const bf16x8 vf = qs[dt % C::NKD];
          asm volatile("" ::"v"(v0), "v"(v1));
#else
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wlds_v4_ptr)(v0 + 32 * dt));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wlds_v4_ptr)(v1 + 32 * dt));
          const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
        }
      }

      WTRACE(5 + 5 * t);
      if (has_next && t == 0) issue_dma(DMA_Q2, DMA_Q3);
      // ---- out[q][16dt + 4fh + j] = O^T / sum ----
      const float inv = 1.0f / psum;
      if (fr < S) {
        bf16_t* orow = ob + (long)(qh * S + fr) * p.o_st + 4 * fh;
#pragma unroll
        for (int dt = 0; dt < C::ND; ++dt) {
          float v[4] = {oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv};
          store4(orow + 16 * dt, v);
        }
      }
    }

    WTRACE(12);
    if (!has_next) break;
    item = next;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my share of the next item has landed in LDS
    settle_q();
    WTRACE(13);
    __syncthreads();
    WTRACE(14);
  }
}

}  // namespace

#ifdef HAFF_WIN_TRACE
extern "C" int haff_win_trace_read(unsigned long long* host, int n_words) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(haff_win_trace_buf), sizeof(unsigned long long) * n_words) == hipSuccess ? 0 : 1;
}
#endif

// Fused window attention with in-kernel decomposed rel-pos (bf16). q/k/v/o: strides in elements (window, head,
// token); n_tokens = S*S per window. grid_h/grid_w > 0: the windows tile images of grid_h x grid_w tokens
// (ceil(grid/S)^2 windows per image, window-major as window_partition orders them) and tokens beyond the grid are
// PADS that were never written: their q/k/v are read from token row `pad_token` (relative to the base pointers),
// which the caller fills with the projection of a zero token, i.e. the qkv bias (pad after norm1,
// image_encoder.py:179-183). grid_h == 0: every row of every window is real. tab_h/tab_w: bf16 [2S-1][d] contiguous (image_encoder.py:322-351 with
// q_size == k_size: no interpolation). Supported geometry: S == 14, d == 80 (SAM ViT-H windowed blocks);
// anything else returns HAFF_ERR_UNSUPPORTED and the caller uses haff_relpos_tables + haff_attention_bf16.
extern "C" int haff_window_attention_bf16(const void* q, long q_sb, long q_sh, long q_st,
                                          const void* k, long k_sb, long k_sh, long k_st,
                                          const void* v, long v_sb, long v_sh, long v_st,
                                          void* o, long o_sb, long o_sh, long o_st,
                                          int n_windows, int H, int S, int d, float scale,
                                          const void* tab_h, const void* tab_w, int grid_h, int grid_w,
                                          long pad_token, void* stream) {
  if (n_windows <= 0 || H <= 0 || S <= 0 || d <= 0 || !tab_h || !tab_w) return HAFF_ERR_BAD_ARG;
  if ((q_st & 7) || (k_st & 7) || (v_st & 7) || (o_st & 3) || (q_sh & 7) || (k_sh & 7) || (v_sh & 7) || (o_sh & 3) ||
      (q_sb & 7) || (k_sb & 7) || (v_sb & 7) || (o_sb & 3))
    return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(q) & 15) || (reinterpret_cast<uintptr_t>(k) & 15) ||
      (reinterpret_cast<uintptr_t>(v) & 15) || (reinterpret_cast<uintptr_t>(o) & 7) ||
      (reinterpret_cast<uintptr_t>(tab_h) & 15) || (reinterpret_cast<uintptr_t>(tab_w) & 15))
    return HAFF_ERR_BAD_ARG;
  if (S != 14 || d != 80) return HAFF_ERR_UNSUPPORTED;
  // the staging DMA addresses K and V of an item from one base with 32-bit offsets: same (window, head, token)
  // strides for both (true for the fused qkv buffer) and an item span under 4 GiB
  if (k_sb != v_sb || k_sh != v_sh || k_st != v_st) return HAFF_ERR_UNSUPPORTED;
  {
    const long dvk = reinterpret_cast<const bf16_t*>(v) - reinterpret_cast<const bf16_t*>(k);
    const long span = (dvk < 0 ? -dvk : dvk) + (long)S * S * k_st + d;
    if (dvk < 0 || span * 2 >= (1L << 32)) return HAFF_ERR_UNSUPPORTED;
  }
  WinArgs p{reinterpret_cast<const bf16_t*>(q), reinterpret_cast<const bf16_t*>(k), reinterpret_cast<const bf16_t*>(v),
            reinterpret_cast<bf16_t*>(o), q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st,
            n_windows, H, scale, reinterpret_cast<const bf16_t*>(tab_h), reinterpret_cast<const bf16_t*>(tab_w),
            grid_h, grid_w, grid_h > 0 ? (grid_h + S - 1) / S : 0, pad_token};
  if (grid_h > 0) {
    // padded windows: square window grid, whole images, pad row addressable with the 32-bit staging offsets
    // ... and at or after every window's base: the kernel reaches the pad row through an UNSIGNED 32-bit byte offset from
    // that base ((pad_token * k_st - win * k_sb) * 2), which would wrap for a pad row placed before a window
    if (grid_w != grid_h || (n_windows % (p.wps * p.wps)) != 0 || pad_token < 0 || q_st != k_st ||
        (pad_token + 1) * k_st * 2 >= (1L << 32) || pad_token * k_st < (long)(n_windows - 1) * k_sb)
      return HAFF_ERR_BAD_ARG;
    // the V pad row is reached as pad base + (v - k) in ONE unsigned 32-bit byte offset (issue_dma: nx_pad_base + chunk +
    // vmk_bytes): the SUM must fit too — on head-major planes both terms are ~ part * 2 bytes and wrap together from ~171
    // frames per pass on; the caller keeps the token-major layout then
    const long dvk = reinterpret_cast<const bf16_t*>(v) - reinterpret_cast<const bf16_t*>(k);
    if ((pad_token * k_st + dvk + d) * 2 >= (1L << 32)) return HAFF_ERR_UNSUPPORTED;
  }
  using C = WinCfg<80, 14>;
  // the attribute is per device and this entry point keeps no state: set it on every call (a host-side table write)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&window_attn_kernel<80, 14>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
    return HAFF_ERR_LAUNCH;
  // persistent: one 7-wave workgroup per CU (150 KB LDS); a multiple of 8 workgroups keeps the item -> XCD mapping
  const int n_items = n_windows * H;
  dim3 grid(n_items < 256 ? n_items : 256), block(C::NTHREADS);
  hipLaunchKernelGGL((window_attn_kernel<80, 14>), grid, block, C::LDS_BYTES, reinterpret_cast<hipStream_t>(stream), p);
  return haff_check_launch();
}
