// fp32 GEMM with the same fused epilogues as haff_gemm_bf16 — the PARITY-MODE twin.
//
//   C[M,N] = epi( A[M,K] · W[N,K]^T ), everything fp32, k-ordered fmaf accumulation.
//
// BASELINE.json's north_star asks for mask logits within 1e-3 of the reference's fp32 CPU forward; a bf16
// pipeline cannot promise that across 32+32 transformer layers, so the host side can run the whole path in
// fp32 ("parity mode") through this kernel, while the bf16 MFMA kernel is the throughput mode. This one is a
// plain LDS-tiled VALU kernel (64x64x16 tile, 4x4 outputs per thread, strided so stores coalesce and SwiGLU
// gate/up pairs land in one thread); it is not a performance path.
#include "haff_common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

struct GemmF32Args {
  const float* A; long lda;
  const float* W; long ldw;
  float* C; long ldc;
  const float* bias;
  const float* resid; long ldr;
  const int* row_map;
  int M, N, K;
  int act, swiglu;
  int nb_inner;
  long sAo, sAi, sWo, sWi, sCo, sCi;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args p) {
  __shared__ float sA[TK][TM + 1];
  __shared__ float sW[TK][TN + 1];
  if (p.nb_inner > 0) {
    const int zo = blockIdx.y / p.nb_inner, zi = blockIdx.y - zo * p.nb_inner;
    p.A += zo * p.sAo + zi * p.sAi;
    p.W += zo * p.sWo + zi * p.sWi;
    p.C += zo * p.sCo + zi * p.sCi;
  }
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int tiles_n = (p.N + TN - 1) / TN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  const int am = min(m0 + lrow, p.M - 1);
  const int wn = min(n0 + lrow, p.N - 1);
  for (int k0 = 0; k0 < p.K; k0 += TK) {
    float4 a = make_float4(0, 0, 0, 0), w = make_float4(0, 0, 0, 0);
    if (k0 + lk < p.K) {
      a = *reinterpret_cast<const float4*>(p.A + (long)am * p.lda + k0 + lk);
      w = *reinterpret_cast<const float4*>(p.W + (long)wn * p.ldw + k0 + lk);
    }
    __syncthreads();
    sA[lk + 0][lrow] = a.x; sA[lk + 1][lrow] = a.y; sA[lk + 2][lrow] = a.z; sA[lk + 3][lrow] = a.w;
    sW[lk + 0][lrow] = w.x; sW[lk + 1][lrow] = w.y; sW[lk + 2][lrow] = w.z; sW[lk + 3][lrow] = w.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      float av[4], wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = sA[k][ty + 16 * i]; wv[i] = sW[k][tx + 16 * i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
    }
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 16 * i;
    if (m >= p.M) continue;
    long orow = m;
    if (p.row_map) {
      const int mapped = p.row_map[m];
      if (mapped < 0) continue;
      orow = mapped;
    }
    if (!p.swiglu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + tx + 16 * j;
        if (n >= p.N) continue;
        float v = acc[i][j];
        if (p.bias) v += p.bias[n];
        v = apply_act(v, p.act);
        if (p.resid) v += p.resid[orow * p.ldr + n];
        p.C[orow * p.ldc + n] = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int n_in = n0 + tx + 16 * j;  // gate column; up at n_in + 16
        if (n_in >= p.N) continue;
        float g = acc[i][j], u = acc[i][j + 1];
        if (p.bias) { g += p.bias[n_in]; u += p.bias[n_in + 16]; }
        const int n_out = (n0 >> 1) + (j >> 1) * 16 + tx;
        p.C[orow * p.ldc + n_out] = (g / (1.0f + expf(-g))) * u;
      }
    }
  }
}

}  // namespace

extern "C" int haff_gemm_f32(const float* A, long lda, const float* W, long ldw, float* C, long ldc,
                             const float* bias, const float* resid, long ldr, const int* row_map, int M, int N, int K,
                             int act, int swiglu, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 3) || (lda & 3) || (ldw & 3)) return HAFF_ERR_BAD_ARG;
  if (swiglu && ((N & 31) || resid)) return HAFF_ERR_BAD_ARG;
  GemmF32Args p{A, lda, W, ldw, C, ldc, bias, resid, ldr, row_map, M, N, K, act, swiglu, 0, 0, 0, 0, 0, 0, 0};
  const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return haff_check_launch();
}

extern "C" int haff_gemm_f32_batched(const float* A, long lda, long sAo, long sAi, const float* W, long ldw, long sWo,
                                     long sWi, float* C, long ldc, long sCo, long sCi, int nb_outer, int nb_inner, int M,
                                     int N, int K, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || nb_outer <= 0 || nb_inner <= 0) return HAFF_ERR_BAD_ARG;
  if ((K & 3) || (lda & 3) || (ldw & 3) || (sAo & 3) || (sAi & 3) || (sWo & 3) || (sWi & 3)) return HAFF_ERR_BAD_ARG;
  GemmF32Args p{A, lda, W, ldw, C, ldc, nullptr, nullptr, 0, nullptr, M, N, K, 0, 0, nb_inner, sAo, sAi, sWo, sWi, sCo, sCi};
  const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles, nb_outer * nb_inner), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return haff_check_launch();
}
