// TN product of the fine-tune step: C[i][j] = sum_m A[m][i] * B[m][j] — the weight gradient dW = dY^T . X of a trainable Linear
// (torch.nn.Linear under autograd; the reference's trainable set: text_hidden_fcs and both mask decoders, train_ds.py:232-244)
// WITHOUT transposed copies of dY and X: both operands are read as they lie in HBM (row = the contracted index m) and the
// MFMA fragments, whose k index must run along m, come out of LDS through ds_read_b64_tr_b16 (the transposing read of gfx950).
//
// 128 x 128 output tile, 4 waves (2 x 2) of 64 x 64 = 4 x 4 MFMA tiles; the contraction is walked in slabs of 64 rows
// ([64][128] bf16 of each operand per stage, LDS rows padded to 272 B, which the transposing read takes without bank
// conflicts), the next slab's global loads in flight in registers while the current one is multiplied. The contraction is
// usually long and the output small (65 536 image-token rows against 256 x 256 weights), so blockIdx.z splits m across
// workgroups; partial tiles go to an fp32 workspace and gemm_tn_reduce_kernel adds them in index order (no atomics:
// repeatable to the bit) and rounds once.
#include "haff_common.h"

namespace {

constexpr int TT = 128;          // output tile edge
constexpr int TR = 64;           // contraction rows per stage
constexpr int TSTR = TT + 8;     // LDS row stride in elements (272 B)

typedef __attribute__((address_space(3))) bf16x4* lds4_ptr;

// fragment of a 16x16x32 MFMA whose k index runs over the ROWS of an LDS image [k][c]: c = c0 + fr; the lane's 8 k slots are
// rows k0 + 4fh + (0..3) and k0 + 16 + 4fh + (0..3) — the same permutation for both operands, so the sum over k is complete
__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* img, int k0, int c0, int fr, int fh) {
  const bf16_t* a0 = img + (k0 + 4 * fh + (fr >> 2)) * TSTR + c0 + 4 * (fr & 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_ptr)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_ptr)(a0 + 16 * TSTR));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct TnArgs {
  const bf16_t *A, *B;
  long lda, ldb;
  long M;
  int N1, N2;
  long rows_per_split;   // multiple of TR
  float* part;           // [splits][N1][N2]
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t sA[TR * TSTR];
  __shared__ __attribute__((aligned(16))) bf16_t sB[TR * TSTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fh = lane >> 4;
  const int wi = wave >> 1, wj = wave & 1;
  const int i0 = blockIdx.x * TT, j0 = blockIdx.y * TT;
  const long m_lo = (long)blockIdx.z * p.rows_per_split;
  long m_hi = m_lo + p.rows_per_split;
  if (m_hi > p.M) m_hi = p.M;
  // staging: thread t moves 16-byte chunk (t & 15) of rows (t >> 4) + 16 q, q = 0..3, of both operands
  const int sc = (tid & 15) * 8, sr = tid >> 4;
  const bool a_ok = i0 + sc < p.N1, b_ok = j0 + sc < p.N2;   // N1, N2 % 8 == 0: a chunk is whole or absent
  uint4 ra[4], rb[4];
  auto fetch = [&](long m0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long m = m0 + sr + 16 * q;
      const bool in = m < m_hi;
      ra[q] = (in && a_ok) ? *reinterpret_cast<const uint4*>(p.A + m * p.lda + i0 + sc) : uint4{0u, 0u, 0u, 0u};
      rb[q] = (in && b_ok) ? *reinterpret_cast<const uint4*>(p.B + m * p.ldb + j0 + sc) : uint4{0u, 0u, 0u, 0u};
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (m_lo < m_hi) fetch(m_lo);
  for (long m0 = m_lo; m0 < m_hi; m0 += TR) {
    __syncthreads();   // every wave is done reading the previous slab
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<uint4*>(sA + (sr + 16 * q) * TSTR + sc) = ra[q];
      *reinterpret_cast<uint4*>(sB + (sr + 16 * q) * TSTR + sc) = rb[q];
    }
    __syncthreads();
    if (m0 + TR < m_hi) fetch(m0 + TR);   // in flight under the MFMAs below
#pragma unroll
    for (int ks = 0; ks < TR / 32; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = frag_tr(sA, 32 * ks, 64 * wi + 16 * t, fr, fh);
        fb[t] = frag_tr(sB, 32 * ks, 64 * wj + 16 * t, fr, fh);
      }
      // column operand first: the lane ends up with C[i = fr][j = 4fh .. 4fh+3] of every 16 x 16 tile
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
    }
  }
  float* dst = p.part + (long)blockIdx.z * p.N1 * p.N2;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = i0 + 64 * wi + 16 * a + fr;
    if (i >= p.N1) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j0 + 64 * wj + 16 * b + 4 * fh;
      if (j >= p.N2) continue;   // N2 % 8 == 0: the 4 columns are in or out together
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      store4(dst + (long)i * p.N2 + j, v);
    }
  }
}

template <typename TO>
__global__ void gemm_tn_reduce_kernel(const float* part, int splits, long n, TO* out) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z) {
      float v[4];
      load4(part + (long)z * n + i, v);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += v[r];
    }
    store4(out + i, s);
  }
}

// contraction rows per split (multiple of TR): enough workgroups for the chip, at most 64 splits
void tn_geometry(long M, int N1, int N2, long& rows_per_split, int& splits) {
  const long tiles = (long)((N1 + TT - 1) / TT) * ((N2 + TT - 1) / TT);
  long want = (512 + tiles - 1) / tiles;
  if (want > 64) want = 64;
  if (want < 1) want = 1;
  long rps = ((M + want - 1) / want + TR - 1) / TR * TR;
  if (rps < 4 * TR) rps = 4 * TR;   // at least 256 contraction rows per workgroup
  rows_per_split = rps;
  splits = (int)((M + rps - 1) / rps);
}

}  // namespace

extern "C" int haff_gemm_tn_workspace_elems(long M, int N1, int N2) {
  if (M <= 0 || N1 <= 0 || N2 <= 0) return HAFF_ERR_BAD_ARG;
  long rps;
  int splits;
  tn_geometry(M, N1, N2, rps, splits);
  const long n = (long)splits * N1 * N2;
  return n > 0x7fffffffL ? HAFF_ERR_UNSUPPORTED : (int)n;
}

// out [N1][N2] (contiguous; bf16 or f32) = A^T . B with A [M][lda] (N1 columns), B [M][ldb] (N2 columns), both bf16.
// N1, N2, lda, ldb multiples of 8 and 16-byte aligned bases (anything else: HAFF_ERR_UNSUPPORTED, the caller transposes).
extern "C" int haff_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, long M, int N1, int N2, float* workspace,
                                 long workspace_elems, void* out, int out_f32, void* stream) {
  if (M <= 0 || N1 <= 0 || N2 <= 0 || !A || !B || !workspace || !out || lda < N1 || ldb < N2) return HAFF_ERR_BAD_ARG;
  if ((N1 & 7) || (N2 & 7) || (lda & 7) || (ldb & 7) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15))
    return HAFF_ERR_UNSUPPORTED;
  long rps;
  int splits;
  tn_geometry(M, N1, N2, rps, splits);
  const long n = (long)N1 * N2;
  if (workspace_elems < (long)splits * n) return HAFF_ERR_BAD_ARG;
  TnArgs p{(const bf16_t*)A, (const bf16_t*)B, lda, ldb, M, N1, N2, rps, workspace};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((N1 + TT - 1) / TT, (N2 + TT - 1) / TT, splits), dim3(256), 0, s, p);
  long g = (n / 4 + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  if (out_f32) hipLaunchKernelGGL((gemm_tn_reduce_kernel<float>), dim3((unsigned)g), dim3(256), 0, s, workspace, splits, n, (float*)out);
  else hipLaunchKernelGGL((gemm_tn_reduce_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, s, workspace, splits, n, (bf16_t*)out);
  return hipGetLastError() == hipSuccess ? HAFF_OK : HAFF_ERR_LAUNCH;
}
