// Prototype of a ONE-WAVE-PER-SIMD 256x256 bf16 GEMM tile for MI355X (gfx950): 4 waves per workgroup, each wave owns 128x128 of the
// tile with its 256 fp32 accumulators in the accumulator half of the 512-entry register file, and the finished tile leaves as
// bf16 from 128 PARKED registers during the NEXT tile's K loop (one 1-KiB store per K-tile) — the 8-wave tile of gemm_bf16.hip has
// no registers to park a tile in, so its stores (>= 4.3 us per tile through the CU's store path) sit between two K loops.
//   C[m][n] = sum_k A[m][k] * W[n][k] + bias[n]      (A: M x K, W: N x K, both K-contiguous bf16; C bf16)
// K loop: K-tiles of 32 in a 4-stage LDS ring (A 256 x 64 B | W 256 x 64 B per stage, 128 KiB), operands by LDS-DMA four K-tiles
// ahead behind ONE counted wait and ONE workgroup barrier per K-tile; 64 v_mfma_f32_16x16x32_bf16 per K-tile and wave in four
// quadrants of 16, every quadrant reloading ONE 64-row operand half (4 ds_read_b128) for a later quadrant ("snake": 64 fragment
// registers instead of 128 for a double buffer).
// hipcc --offload-arch=gfx950 -O3 -o gemm_w4 gemm_w4.hip && ./gemm_w4 [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <type_traits>
#include "../../2handedafforder_amd/csrc/haff_common.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct W4Args {
  const bf16_t* A; long lda;
  const bf16_t* W; long ldw;
  bf16_t* C; long ldc;
  const float* bias;
  int M, N, K;
  int group_m;
  unsigned long long* stamps;
};

namespace {
constexpr int BM = 256, BN = 256, BK = 32, NST = 4;
constexpr int A_BYTES = BM * BK * 2;          // 16 KiB
constexpr int STAGE_BYTES = 2 * A_BYTES;      // 32 KiB
constexpr int XTRA = NST * STAGE_BYTES;       // bias (1 KiB) behind the ring

__device__ __forceinline__ void permlane16_swap(unsigned& x, unsigned& y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
template <int N> using IC = std::integral_constant<int, N>;
}  // namespace

// 16 MFMAs with 4 fragment reads between them: 2 | r 4 | r 4 | r 4 | r 2
#define W4_SCHED_QUAD() do { \
  __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); \
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); \
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); \
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); \
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); } while (0)
#ifndef W4_EXPLICIT_ZERO
#define W4_EXPLICIT_ZERO 0
#endif
#ifdef W4_NOSCHED
#undef W4_SCHED_QUAD
#define W4_SCHED_QUAD() do {} while (0)
#endif
#ifdef W4_LOOSEWAIT
#define W4_WAIT_IMM 0x8070   // vmcnt(32) lgkmcnt(0): timing experiment (operands may not have landed)
#else
#define W4_WAIT_IMM 0x4070   // vmcnt(16) lgkmcnt(0) expcnt(7)
#endif
#ifndef W4_SPREAD
#define W4_SPREAD 0
#endif
#ifndef W4_DIRECT_MI
#define W4_DIRECT_MI 0
#endif
#ifndef W4_DIRECT
#define W4_DIRECT 0
#endif
#ifndef W4_NOSTORE_OVERLAP
#define W4_NOSTORE_OVERLAP 0
#endif

__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(W4Args p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE_BYTES + 1024];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fh = lane >> 4;

  const int tiles_m = p.M / BM, tiles_n = p.N / BN;
  const int nwg = tiles_m * tiles_n;
  auto tile_origin = [&](int t, int& tm0, int& tn0) {
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    const int per_group = p.group_m * tiles_n;
    const int g = lin / per_group;
    const int first_m = g * p.group_m;
    const int gsz = min(tiles_m - first_m, p.group_m);
    const int in_g = lin - g * per_group;
    tm0 = (first_m + in_g % gsz) * BM;
    tn0 = (in_g / gsz) * BN;
  };
  int tile = blockIdx.x;
  int m0, n0;
  tile_origin(tile, m0, n0);

  // ---- staging coordinates: 16-B chunks; LDS position pos = i*256 + tid; row = pos >> 2; physical chunk = pos & 3 holds logical
  // chunk (pos & 3) ^ sw(row), sw(row) = (-(row >> 2)) & 3: with 64-byte rows this makes the 16x16x32 fragment read (lane (fr, fh):
  // row fr, logical chunk fh) conflict-free in every one of ds_read_b128's four 16-lane groups
  unsigned a_off[4], w_off[4];
  auto stage_coords = [&](int tm0, int tn0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#ifdef W4_FATROWS   // timing experiment (wrong operands): every request covers 8 rows x 128 B instead of 16 rows x 64 B
      const int pos = i * 256 + tid, row = (i * 256 + tid) >> 3;
      const int lch = pos & 7;
#else
      const int pos = i * 256 + tid, row = pos >> 2;
      const int lch = (pos & 3) ^ ((-(row >> 2)) & 3);
#endif
      a_off[i] = ((unsigned)(tm0 + row) * (unsigned)p.lda + lch * 8) * 2u;
      w_off[i] = ((unsigned)(tn0 + row) * (unsigned)p.ldw + lch * 8) * 2u;
    }
  };
  stage_coords(m0, n0);
  auto dma_a = [&](int stage, int k0, int i0, int i1) {
    const bf16_t* base = p.A + k0;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < i0 || i >= i1) continue;
#ifdef W4_HALFDMA
      if (i & 1) continue;
#endif
      unsigned o = a_off[i];
      asm volatile("" : "+v"(o));
#ifndef W4_NODMA
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(base) + o),
                                       (lptr_t)(smem + stage * STAGE_BYTES + (i * 256 + wave * 64) * 16), 16, 0, 0);
#endif
    }
  };
  auto dma_w = [&](int stage, int k0, int i0, int i1) {
    const bf16_t* base = p.W + k0;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < i0 || i >= i1) continue;
#ifdef W4_HALFDMA
      if (i & 1) continue;
#endif
      unsigned o = w_off[i];
      asm volatile("" : "+v"(o));
#ifndef W4_NODMA
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(base) + o),
                                       (lptr_t)(smem + stage * STAGE_BYTES + A_BYTES + (i * 256 + wave * 64) * 16), 16, 0, 0);
#endif
    }
  };

  // one request of the K loop, in the fewest issue slots: m0 <- LDS address (scalar add), one wait state, the load with an SGPR
  // base and the lane's 32-bit offset. (Through the builtin hipcc spent 6 slots per request — register copies around the
  // address, the m0 move, a nop — and a one-wave-per-SIMD loop pays for every slot its 16-cycle MFMAs do not cover.)
  const unsigned wave_lds = (unsigned)(uintptr_t)(lptr_t)smem + wave * 1024;   // LDS byte address of this wave's slice of request 0, stage 0, A
  auto dma1 = [&](unsigned voff, const void* base, auto lds_tag) {
    constexpr int LDS = decltype(lds_tag)::value;
    (void)wave_lds;
#ifdef W4_ONEWAVE_DMA   // timing experiment (wrong operands): only one wave of the workgroup issues its requests
    if (wave != 0) return;
#endif
#ifndef W4_NODMA
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(base), "s"(wave_lds), "i"(LDS) : "memory", "scc");
#endif
  };

  const int nk = p.K / BK;        // multiple of 4 (host)
  const int nq = nk >> 2;

  // fragment read addresses (bytes inside a stage)
  const unsigned sw = (unsigned)((fh ^ ((-(fr >> 2)) & 3)) << 4);
  const unsigned rd_a = wm * (128 * 64) + fr * 64 + sw;
  const unsigned rd_w = A_BYTES + wn * (128 * 64) + fr * 64 + sw;

  f32x4 acc[8][8];        // [ni][mi]
  bf16x8 af[2][4], bf[2][4];
  u32x4 parked[8][4];     // [mi][chunk pair]: 8 bf16 = 16 B per lane
  unsigned park_off = 0;      // lane's first output byte of the parked tile, from p.C (outputs stay under 4 GiB: host)
  const unsigned c_pass = 16u * (unsigned)p.ldc * 2u;

  auto read_half = [&](bf16x8 (&dst)[4], unsigned rd, int stage, int half) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      dst[t] = *reinterpret_cast<const bf16x8*>(smem + stage * STAGE_BYTES + rd + (half * 4 + t) * 1024);
  };
  auto store_one = [&](auto idx_tag) {
    constexpr int IDX = decltype(idx_tag)::value;
    constexpr int mi = IDX >> 2, j = IDX & 3;
    // SGPR base + 32-bit lane offset, formed at the store (as 64-bit lane addresses hipcc computed all 32 ahead of the K loop
    // and kept them in 64 registers)
    unsigned o = park_off + mi * c_pass + j * 64;
    asm volatile("" : "+v"(o));
    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.C) + o) = parked[mi][j];
  };
  auto store_all = [&]() {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned o = park_off + mi * c_pass + j * 64;
        asm volatile("" : "+v"(o));
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.C) + o) = parked[mi][j];
      }
  };

  // ---- prologue: K-tiles 0..3 of the first tile, fragments A0 | B0 of K-tile 0
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    dma_a(s, s * BK, 0, 4);
    dma_w(s, s * BK, 0, 4);
  }
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_half(af[0], rd_a, 0, 0);
  read_half(bf[0], rd_w, 0, 0);

  bool have_parked = false;
  for (;;) {
    const int tile_next = tile + (int)gridDim.x;
    const bool has_next = tile_next < nwg;
    const int m0e = m0, n0e = n0;

    // one K-tile: stage S = kt & 3, parity P = kt & 1. F = the A half the K-tile starts with, X = the other one.
    // ZERO: first K-tile of the tile (accumulators start from zero). last: one of the tile's last four K-tiles — its requests are
    // the next tile's first K-tiles (the staging coordinates have moved on), if there is a next tile
    // ST: the parked tile's stores 2 * ST and 2 * ST + 1 leave behind this K-tile (-1: none) — two 1-KiB stores per wave and
    // K-tile are 8 B/clk per CU, under the ~14 B/clk its store path takes
    auto ktile = [&](int kq, bool last, auto s_tag, auto zero_tag, auto st_tag) {
      constexpr int S = decltype(s_tag)::value, P = S & 1, F = P, X = 1 - P, SN = (S + 1) & 3;
      constexpr bool ZERO = decltype(zero_tag)::value;
      constexpr int ST = decltype(st_tag)::value;
      const int kt = kq * 4 + S;
      // MFMAs [I0, I1) of the 16 of quadrant (A half AH, B half BH), as inline asm with the accumulator tied in place in the
      // accumulator registers: with the builtin hipcc wrote every result to a NEW register quad and shuffled accumulators
      // through v_accvgpr_read / write (256 accumulators leave the allocator no slack in the 256-entry accumulator half)
      auto mfmas = [&](auto ah_tag, auto bh_tag, auto i0_tag, auto i1_tag) {
        constexpr int AH = decltype(ah_tag)::value, BH = decltype(bh_tag)::value;
        constexpr int I0 = decltype(i0_tag)::value, I1 = decltype(i1_tag)::value;
        (void)acc; (void)af; (void)bf;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
          const int t = i >> 2, u = i & 3;
          if constexpr (ZERO)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[BH * 4 + u][AH * 4 + t]) : "v"(bf[BH][u]), "v"(af[AH][t]));
          else
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[BH * 4 + u][AH * 4 + t]) : "v"(bf[BH][u]), "v"(af[AH][t]));
        }
      };
      auto read1 = [&](bf16x8& dst, unsigned rd, int stage, int tile16) {
#ifdef W4_NOREAD
        asm volatile("" : "+v"(dst));
#else
        dst = *reinterpret_cast<const bf16x8*>(smem + stage * STAGE_BYTES + rd + tile16 * 1024);
#endif
      };
#define SB() __builtin_amdgcn_sched_barrier(0)
      // one quadrant: 16 MFMAs, the 4 fragment reads of a LATER quadrant and up to 4 operand requests between them
      auto quadrant = [&](auto ah_tag, auto bh_tag, bf16x8 (&dst)[4], unsigned rd, int stage, int half, auto&& req) {
        // one fragment read or one request per MFMA gap (a 16-cycle MFMA covers about two issue slots of its own wave); reads
        // first: the next quadrant's first MFMA needs them and a ds_read_b128 takes 100+ cycles to come back
        mfmas(ah_tag, bh_tag, IC<0>{}, IC<1>{}); SB();
        read1(dst[0], rd, stage, half * 4 + 0); SB();
        mfmas(ah_tag, bh_tag, IC<1>{}, IC<2>{}); SB();
        read1(dst[1], rd, stage, half * 4 + 1); SB();
        mfmas(ah_tag, bh_tag, IC<2>{}, IC<3>{}); SB();
        read1(dst[2], rd, stage, half * 4 + 2); SB();
        mfmas(ah_tag, bh_tag, IC<3>{}, IC<4>{}); SB();
        read1(dst[3], rd, stage, half * 4 + 3); SB();
        mfmas(ah_tag, bh_tag, IC<4>{}, IC<6>{}); SB();
        req(IC<0>{}); SB();
        mfmas(ah_tag, bh_tag, IC<6>{}, IC<8>{}); SB();
        req(IC<1>{}); SB();
        mfmas(ah_tag, bh_tag, IC<8>{}, IC<10>{}); SB();
        req(IC<2>{}); SB();
        mfmas(ah_tag, bh_tag, IC<10>{}, IC<12>{}); SB();
        req(IC<3>{}); SB();
        mfmas(ah_tag, bh_tag, IC<12>{}, IC<16>{}); SB();
      };
      auto no_req = [&](auto) {};
      // Q0: (F, B0); reload A half X of THIS K-tile.  Q1: (X, B0); reload B half 1 of this K-tile
      quadrant(IC<F>{}, IC<0>{}, af[X], rd_a, S, X, no_req);
      quadrant(IC<X>{}, IC<0>{}, bf[1], rd_w, S, 1, no_req);
      // every read of stage S is done (mine), my share of K-tile kt + 1 has landed (K-tiles kt + 2, kt + 3 may be in flight)
      // (the builtin, not inline asm: hipcc's own wait insertion reads it and knows the fragment reads are complete; behind an
      // asm wait it put an lgkmcnt(0) after the first new ds_read of Q2 — LDS-DMA counts as a flat access, after which it
      // only ever waits for zero — and exposed that read's latency once per K-tile)
      __builtin_amdgcn_s_waitcnt(W4_WAIT_IMM);   // vmcnt(16) [or 32: W4_LOOSEWAIT] lgkmcnt(0)
#ifndef W4_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
      SB();
      // stage S is free: K-tile kt + 4 of this tile, or K-tile S of the next one (of this one again if there is no next tile:
      // requests nobody reads, so that the counted wait and the instruction stream stay the same to the end)
      const int k4 = last ? S * BK : (kt + 4) * BK;
      // Q2: (X, B1); reload B half 0 of the NEXT K-tile; request A.  Q3: (F, B1); reload A half X of the next K-tile; request W
      const bf16_t* a_base = p.A + k4;
      const bf16_t* w_base = p.W + k4;
      quadrant(IC<X>{}, IC<1>{}, bf[0], rd_w, SN, 0, [&](auto i) { constexpr int I = decltype(i)::value; dma1(a_off[I], a_base, IC<S * STAGE_BYTES + I * 4096>{}); });
      quadrant(IC<F>{}, IC<1>{}, af[X], rd_a, SN, X, [&](auto i) { constexpr int I = decltype(i)::value; dma1(w_off[I], w_base, IC<S * STAGE_BYTES + A_BYTES + I * 4096>{}); });
#undef SB
#if !W4_NOSTORE_OVERLAP
      if constexpr (ST >= 0) {
        if (have_parked) {
          store_one(IC<2 * (ST >= 0 ? ST : 0)>{});
          store_one(IC<2 * (ST >= 0 ? ST : 0) + 1>{});
        }
      }
#endif
    };

#if W4_EXPLICIT_ZERO
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    auto next_coords = [&](bool last) {
      if (last && has_next) {   // the last four K-tiles request the next tile's first four
        tile_origin(tile_next, m0, n0);
        stage_coords(m0, n0);
      }
    };
    using NOZ = std::false_type;
#ifdef W4_STAMP
    unsigned long long t_k0 = 0, r_k0 = 0;
    if (tile == (int)blockIdx.x + 2 * (int)gridDim.x) { t_k0 = __builtin_amdgcn_s_memtime(); r_k0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    // the first 16 K-tiles carry the parked tile's 32 stores (nq >= 4: host)
    ktile(0, false, IC<0>{}, std::true_type{}, IC<0>{});
    if (p.bias && wave == 0) {   // this tile's 256 bias values -> LDS (behind K-tile 0's barrier: every wave has left the last epilogue)
      const float* b = p.bias + n0e;
      asm volatile("" : "+s"(b));
      unsigned o = (unsigned)lane * 16u;
      asm volatile("" : "+v"(o));
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(b) + o), (lptr_t)(smem + XTRA), 16, 0, 0);
    }
    ktile(0, false, IC<1>{}, NOZ{}, IC<1>{});
    ktile(0, false, IC<2>{}, NOZ{}, IC<2>{});
    ktile(0, false, IC<3>{}, NOZ{}, IC<3>{});
    ktile(1, false, IC<0>{}, NOZ{}, IC<4>{});
    ktile(1, false, IC<1>{}, NOZ{}, IC<5>{});
    ktile(1, false, IC<2>{}, NOZ{}, IC<6>{});
    ktile(1, false, IC<3>{}, NOZ{}, IC<7>{});
    ktile(2, false, IC<0>{}, NOZ{}, IC<8>{});
    ktile(2, false, IC<1>{}, NOZ{}, IC<9>{});
    ktile(2, false, IC<2>{}, NOZ{}, IC<10>{});
    ktile(2, false, IC<3>{}, NOZ{}, IC<11>{});
    {
      const bool last = nq == 4;
      next_coords(last);
      ktile(3, last, IC<0>{}, NOZ{}, IC<12>{});
      ktile(3, last, IC<1>{}, NOZ{}, IC<13>{});
      ktile(3, last, IC<2>{}, NOZ{}, IC<14>{});
      ktile(3, last, IC<3>{}, NOZ{}, IC<15>{});
    }
    for (int kq = 4; kq < nq; ++kq) {
      const bool last = kq == nq - 1;
      next_coords(last);
      ktile(kq, last, IC<0>{}, NOZ{}, IC<-1>{});
      ktile(kq, last, IC<1>{}, NOZ{}, IC<-1>{});
      ktile(kq, last, IC<2>{}, NOZ{}, IC<-1>{});
      ktile(kq, last, IC<3>{}, NOZ{}, IC<-1>{});
    }
#ifdef W4_STAMP
    if (tile == (int)blockIdx.x + 2 * (int)gridDim.x && lane == 0) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      p.stamps[(blockIdx.x * 4 + wave) * 2] = t1 - t_k0;
      p.stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r_k0;
    }
#endif
#if W4_NOSTORE_OVERLAP
    if (have_parked) store_all();
#endif
    // ---- epilogue: bias, bf16, pair two 4-column chunks per lane (v_permlane16_swap) -> park
    {
      const float* sBias = reinterpret_cast<const float*>(smem + XTRA) + wn * 128;
      const int coff = 16 * (fh & 1) + 4 * (fh & 2);
      park_off = ((unsigned)(m0e + wm * 128 + fr) * (unsigned)p.ldc + n0e + wn * 128 + coff) * 2u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float b0[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          load4(sBias + (2 * j) * 16 + fh * 4, b0);
          load4(sBias + (2 * j + 1) * 16 + fh * 4, b1);
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const f32x4 a = acc[2 * j][mi], b = acc[2 * j + 1][mi];
          unsigned x0 = pack_bf16x2(a[0] + b0[0], a[1] + b0[1]), y0 = pack_bf16x2(b[0] + b1[0], b[1] + b1[1]);
          unsigned x1 = pack_bf16x2(a[2] + b0[2], a[3] + b0[3]), y1 = pack_bf16x2(b[2] + b1[2], b[3] + b1[3]);
          permlane16_swap(x0, y0);
          permlane16_swap(x1, y1);
          if (W4_DIRECT || mi < W4_DIRECT_MI)
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.C) + (park_off + mi * c_pass + j * 64)) = u32x4{x0, x1, y0, y1};
          else
            parked[mi][j] = u32x4{x0, x1, y0, y1};
          __builtin_amdgcn_sched_barrier(0);   // one (row tile, chunk pair) at a time: hoisted accumulator reads cost 90 registers
        }
      }
#if !W4_DIRECT
      have_parked = true;
#endif
    }
    if (!has_next) break;
    tile = tile_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no request may be in flight to this workgroup's LDS when it ends
  // the last tile's stores
  store_all();
}

// ---- reference: one thread per output, fp32 accumulate
__global__ void ref_kernel(W4Args p, float* out, int m_lo, int m_hi) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long rows = m_hi - m_lo;
  if (idx >= rows * p.N) return;
  const int m = m_lo + (int)(idx / p.N), n = (int)(idx % p.N);
  float s = 0.f;
  for (int k = 0; k < p.K; ++k) s += bf16_to_f32(p.A[(long)m * p.lda + k]) * bf16_to_f32(p.W[(long)n * p.ldw + k]);
  out[idx] = s + (p.bias ? p.bias[n] : 0.f);
}

static bf16_t h_bf16(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1);
  return (bf16_t)(u >> 16);
}

int main(int argc, char** argv) {
  int M = 131072, N = 3840, K = 1280;
  if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
  const int iters = argc >= 5 ? atoi(argv[4]) : 10;
  if (M % 256 || N % 256 || K % 128 || K < 512) { printf("shape must be multiples of 256 x 256 x 128\n"); return 1; }
  std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
  std::vector<float> hb(N);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hA) v = h_bf16(rnd());
  for (auto& v : hW) v = h_bf16(rnd() * 0.2f);
  for (auto& v : hb) v = rnd();
  bf16_t *dA, *dW, *dC;
  float *db, *dref;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 2); hipMalloc(&db, N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
  hipMemset(dC, 0xff, (size_t)M * N * 2);
  unsigned long long* dst;
  hipMalloc(&dst, 256 * 4 * 2 * 8);
  hipMemset(dst, 0, 256 * 4 * 2 * 8);
  W4Args p{dA, K, dW, K, dC, N, db, M, N, K, 0, dst};
  const int tiles_n = N / 256, tiles_m = M / 256;
  p.group_m = tiles_n >= 24 ? 4 : (K >= 4096 ? 2 : 8);
  if (getenv("GROUP_M")) p.group_m = atoi(getenv("GROUP_M"));
  const int nwg = tiles_m * tiles_n;
  const int grid = nwg < 256 ? nwg : 256;
  hipLaunchKernelGGL(gemm_w4_kernel, dim3(grid), dim3(256), 0, 0, p);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
  // check rows [0, 512) and the last 256 rows against the reference kernel
  const int chk_rows = 512;
  hipMalloc(&dref, (size_t)chk_rows * N * 4);
  std::vector<float> href((size_t)chk_rows * N);
  std::vector<bf16_t> hC((size_t)chk_rows * N);
  double worst = 0;
  long bad = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const int lo = pass == 0 ? 0 : M - chk_rows, hi = lo + chk_rows;
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)(((long)chk_rows * N + 255) / 256)), dim3(256), 0, 0, p, dref, lo, hi);
    hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hC.data(), dC + (size_t)lo * N, hC.size() * 2, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < href.size(); ++i) {
      unsigned u = (unsigned)hC[i] << 16;
      float c;
      memcpy(&c, &u, 4);
      const double d = fabs((double)c - href[i]), tol = 0.02 + 0.01 * fabs(href[i]);
      if (!(d <= tol)) { if (bad < 5) printf("  bad [%zu][%zu]: %g vs %g\n", lo + i / N, i % N, c, href[i]); ++bad; }
      if (d > worst) worst = d;
    }
  }
  printf("check: worst abs err %.4g, bad %ld of %zu\n", worst, bad, 2 * href.size());
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_w4_kernel, dim3(grid), dim3(256), 0, 0, p);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_w4_kernel, dim3(grid), dim3(256), 0, 0, p);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  printf("%d x %d x %d  group_m %d: %.1f us  %.0f TFLOP/s\n", M, N, K, p.group_m, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
#ifdef W4_STAMP
  {
    std::vector<unsigned long long> hs(256 * 4 * 2);
    hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int i = 0; i < 256 * 4; ++i) if (hs[2 * i]) { cyc.push_back((double)hs[2 * i]); clk.push_back(hs[2 * i] / (hs[2 * i + 1] * 10.0)); }
    if (!cyc.empty()) {
      std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
      const double c = cyc[cyc.size() / 2];
      printf("K loop of a workgroup's third tile: median %.0f cycles = %.1f per K-tile (1024 = MFMA-bound), clock %.2f GHz (median), %zu waves\n",
             c, c / (K / 32), clk[clk.size() / 2] / 1000.0 * 1000.0, cyc.size());
    }
  }
#endif
  return bad ? 2 : 0;
}
