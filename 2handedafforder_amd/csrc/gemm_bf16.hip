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
// Design (gfx950): 128x128x64 tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave, 4x4 tiles of
// v_mfma_f32_16x16x32_bf16). Both operands are staged HBM->LDS with global_load_lds_dwordx4 (no VGPR round
// trip); the LDS image is lane-linear and the XOR swizzle (chunk ^= row&7) is applied on the per-lane SOURCE
// address and again on the ds_read_b128 address, which makes the fragment reads bank-conflict free.
// Double-buffered with a counted s_waitcnt vmcnt(8) + raw s_barrier so the next K-tile's loads stay in
// flight across the barrier while the current tile is multiplied.
// The MFMA is issued "swapped" (W rows as the A operand, activation rows as the B operand): each lane then
// holds 4 CONSECUTIVE output columns of one output row, so bias/residual loads and the C stores are 8-byte
// (bf16) / 16-byte (f32) vectors and SwiGLU pairs (gate, up) land in the same lane.
// Workgroup ids are remapped XCD-aware (ids that share an XCD get neighbouring tiles) and grouped 8 M-tiles
// deep so the 4 MiB per-XCD L2 holds the A and W panels the concurrently running tiles share.
#include "haff_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int TILE_ELEMS = BM * BK;  // per operand per buffer (BM == BN)
constexpr int GROUP_M = 8;

__device__ __attribute__((aligned(16))) unsigned int haff_zero_page[8];  // 32 B of zeros for K-tail chunks

struct GemmArgs {
  const bf16_t* A; long lda;
  const bf16_t* W; long ldw;
  void* C; long ldc;
  const float* bias;
  const void* resid; long ldr;
  const int* row_map;
  int M, N, K;
  int act, out_f32, swiglu;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <bool OUT_F32, bool SWIGLU>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 2 * TILE_ELEMS];  // [buf][A|W][128][64] = 64 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware + grouped tile mapping (speed only; any mapping is correct) ----
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int lin;
  {
    const int orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int tm, tn;
  {
    const int per_group = GROUP_M * tiles_n;
    const int g = lin / per_group;
    const int first_m = g * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_g = lin - g * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-thread staging coordinates: 4 chunks of 16 B per operand per K-tile ----
  // LDS position pos = i*256 + tid (lane-linear); row = pos>>3; logical chunk = (pos&7) ^ (row&7)
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
  int kcol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pos = i * NTHREADS + tid;
    const int row = pos >> 3;
    const int c = (pos & 7) ^ (row & 7);
    kcol[i] = c * 8;
    const int am = min(m0 + row, p.M - 1);
    const int wn_ = min(n0 + row, p.N - 1);
    a_src[i] = p.A + (long)am * p.lda + c * 8;
    w_src[i] = p.W + (long)wn_ * p.ldw + c * 8;
  }
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(haff_zero_page);

  auto stage = [&](int buf, int k0) {
    bf16_t* sA = smem + buf * 2 * TILE_ELEMS;
    bf16_t* sW = sA + TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool in_k = (k0 + kcol[i]) < p.K;
      const bf16_t* ga = in_k ? a_src[i] + k0 : zero_src;
      const bf16_t* gw = in_k ? w_src[i] + k0 : zero_src;
      // wave-uniform LDS base; hardware adds lane*16
      bf16_t* la = sA + (i * NTHREADS + wave * 64) * 8;
      bf16_t* lw = sW + (i * NTHREADS + wave * 64) * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)la, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)gw, (lptr_t)lw, 16, 0, 0);
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fh = lane >> 4;

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      stage(cur ^ 1, (kt + 1) * BK);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    const bf16_t* sA = smem + cur * 2 * TILE_ELEMS;
    const bf16_t* sW = sA + TILE_ELEMS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4], af[4];
      const int c = ks * 4 + fh;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rw = wn * 64 + t * 16 + fr;
        const int ra = wm * 64 + t * 16 + fr;
        wf[t] = *reinterpret_cast<const bf16x8*>(sW + rw * BK + ((c ^ (rw & 7)) << 3));
        af[t] = *reinterpret_cast<const bf16x8*>(sA + ra * BK + ((c ^ (ra & 7)) << 3));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: lane holds D[n = 4*fh + reg][m = fr] of each 16x16 tile ----
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + fr;
    if (m >= p.M) continue;
    long orow = m;
    if (p.row_map) {
      const int mapped = p.row_map[m];
      if (mapped < 0) continue;
      orow = mapped;
    }
    if (!SWIGLU) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + fh * 4;
        if (n >= p.N) continue;
        float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
        const bool full = (n + 4 <= p.N);
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (full || n + r < p.N) v[r] += p.bias[n + r];
        }
        if (p.act != HAFF_ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
        }
        if (OUT_F32) {
          float* crow = reinterpret_cast<float*>(p.C) + orow * p.ldc + n;
          if (p.resid) {
            const float* rrow = reinterpret_cast<const float*>(p.resid) + orow * p.ldr + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (full || n + r < p.N) v[r] += rrow[r];
          }
          if (full && ((p.ldc & 3) == 0)) {
            store4(crow, v);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) crow[r] = v[r];
          }
        } else {
          bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + orow * p.ldc + n;
          if (p.resid) {
            const bf16_t* rrow = reinterpret_cast<const bf16_t*>(p.resid) + orow * p.ldr + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (full || n + r < p.N) v[r] += bf16_to_f32(rrow[r]);
          }
          if (full && ((p.ldc & 3) == 0)) {
            store4(crow, v);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) crow[r] = f32_to_bf16(v[r]);
          }
        }
      }
    } else {
      // W rows are interleaved in 16-row groups [gate x16 | up x16]; output width N/2
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) {
        const int n_in = n0 + wn * 64 + nj * 32 + fh * 4;  // gate column (interleaved index)
        if (n_in >= p.N) continue;
        const int n_out = ((n0 + wn * 64) >> 1) + nj * 16 + fh * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float g = acc[2 * nj][mi][r];
          float u = acc[2 * nj + 1][mi][r];
          if (p.bias) { g += p.bias[n_in + r]; u += p.bias[n_in + 16 + r]; }
          v[r] = (g / (1.0f + __expf(-g))) * u;
        }
        if (OUT_F32) store4(reinterpret_cast<float*>(p.C) + orow * p.ldc + n_out, v);
        else store4(reinterpret_cast<bf16_t*>(p.C) + orow * p.ldc + n_out, v);
      }
    }
  }
}

}  // namespace

extern "C" int haff_gemm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc,
                              const float* bias, const void* resid, long ldr, const int* row_map,
                              int M, int N, int K, int act, int out_f32, int swiglu, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return HAFF_ERR_BAD_ARG;
  if ((K & 7) || (lda & 7) || (ldw & 7)) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return HAFF_ERR_BAD_ARG;
  if (swiglu && ((N & 31) || (ldc & 3) || resid)) return HAFF_ERR_BAD_ARG;
  GemmArgs p{reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(W), ldw, C, ldc,
             bias, resid, ldr, row_map, M, N, K, act, out_f32, swiglu};
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(tiles), block(NTHREADS);
  if (swiglu) {
    if (out_f32) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, s, p);
  } else {
    if (out_f32) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, s, p);
  }
  return haff_check_launch();
}
