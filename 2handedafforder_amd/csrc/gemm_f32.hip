// fp32 GEMM with the same fused epilogues as haff_gemm_bf16 — exact-f32 twin on the f32-input matrix cores.
//
//   C[M,N] = epi( A[M,K] · W[N,K]^T ), everything fp32: v_mfma_f32_16x16x4_f32 is bit-for-bit an f32 fmaf chain
//   (one rounding per product, no wider accumulator), at the f32 vector peak (64 FLOP/clk/SIMD).
//
// Two users: (1) the fp32 "parity mode" of the whole path (mask logits within 1e-3 of the fp32 CPU forward,
// BASELINE.json north_star); (2) the fp32 DECODER TAIL of the bf16 throughput mode — text_hidden_fcs, both two-way
// mask decoders, hypernetwork / IoU / taxonomy MLPs and the first transposed conv run in fp32 on fp32 image
// embeddings (7 GFLOP per frame; SURVEY section 7, hard part 3).
//
// 128x128x16 tile, 4 waves (2x2), 64x64 per wave = 4x4 MFMA tiles. Operands are staged through registers into LDS
// rows of 16 floats padded to 20 (80 B: the 16 rows of a fragment start on 16 different 16-B bank groups); a
// fragment is ONE ds_read_b128 per lane — lane (r = l & 15, g = l >> 4) reads k = 4g .. 4g+3 of row r — and the four
// components feed four MFMAs: MFMA c sums k = {c, 4+c, 8+c, 12+c}, the same permutation on both operands, so the 16
// k of the tile are covered once. Swapped orientation (W is the MFMA's A operand): a lane's 4 accumulator registers
// are 4 CONSECUTIVE output columns of one row -> 16-B epilogue loads/stores, SwiGLU (gate, up) pairs in one lane.
#include "haff_common.h"

namespace {

constexpr int TM = 128, TN = 128, TK = 16, RS = 20;

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
  __shared__ __attribute__((aligned(16))) float sA[2][TM * RS];
  __shared__ __attribute__((aligned(16))) float sW[2][TN * RS];
  if (p.nb_inner > 0) {
    const int zo = blockIdx.y / p.nb_inner, zi = blockIdx.y - zo * p.nb_inner;
    p.A += zo * p.sAo + zi * p.sAi;
    p.W += zo * p.sWo + zi * p.sWi;
    p.C += zo * p.sCo + zi * p.sCi;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int tiles_n = (p.N + TN - 1) / TN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: thread -> rows (tid >> 2) and (tid >> 2) + 64, k chunk (tid & 3) * 4
  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  const float* a0 = p.A + (long)min(m0 + lrow, p.M - 1) * p.lda + lk;
  const float* a1 = p.A + (long)min(m0 + lrow + 64, p.M - 1) * p.lda + lk;
  const float* w0 = p.W + (long)min(n0 + lrow, p.N - 1) * p.ldw + lk;
  const float* w1 = p.W + (long)min(n0 + lrow + 64, p.N - 1) * p.ldw + lk;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ra0, ra1, rw0, rw1;
  auto fetch = [&](int k0) {
    const bool ok = k0 + lk < p.K;
    ra0 = ok ? *reinterpret_cast<const float4*>(a0 + k0) : z4;
    ra1 = ok ? *reinterpret_cast<const float4*>(a1 + k0) : z4;
    rw0 = ok ? *reinterpret_cast<const float4*>(w0 + k0) : z4;
    rw1 = ok ? *reinterpret_cast<const float4*>(w1 + k0) : z4;
  };
  auto stash = [&](int buf) {
    *reinterpret_cast<float4*>(&sA[buf][lrow * RS + lk]) = ra0;
    *reinterpret_cast<float4*>(&sA[buf][(lrow + 64) * RS + lk]) = ra1;
    *reinterpret_cast<float4*>(&sW[buf][lrow * RS + lk]) = rw0;
    *reinterpret_cast<float4*>(&sW[buf][(lrow + 64) * RS + lk]) = rw1;
  };
  const int nk = (p.K + TK - 1) / TK;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch((kt + 1) * TK);  // in flight under the MFMAs below
    float4 wf[4], af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wf[i] = *reinterpret_cast<const float4*>(&sW[buf][(wn * 64 + i * 16 + fr) * RS + fg * 4]);
      af[i] = *reinterpret_cast<const float4*>(&sA[buf][(wm * 64 + i * 16 + fr) * RS + fg * 4]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].x, af[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].y, af[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].z, af[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].w, af[j].w, acc[i][j], 0, 0, 0);
      }
    if (kt + 1 < nk) {
      stash(buf ^ 1);  // the other buffer was last read in iteration kt-1, behind the barrier below
      __syncthreads();
    }
  }

  // epilogue: acc[i][j][r] = C[m0 + wm*64 + j*16 + fr][n0 + wn*64 + i*16 + fg*4 + r]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + fr;
    if (m >= p.M) continue;
    long orow = m;
    if (p.row_map) {
      const int mapped = p.row_map[m];
      if (mapped < 0) continue;
      orow = mapped;
    }
    if (!p.swiglu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + fg * 4;
        if (n >= p.N) continue;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        float* cp = p.C + orow * p.ldc + n;
        const float* rp = p.resid ? p.resid + orow * p.ldr + n : nullptr;
        const bool vec = n + 3 < p.N && ((reinterpret_cast<uintptr_t>(cp) & 15) == 0) &&
                         (!rp || (reinterpret_cast<uintptr_t>(rp) & 15) == 0) &&
                         (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
        if (vec) {
          if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
          if (rp) {
            const float4 q = *reinterpret_cast<const float4*>(rp);
            v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
          }
          *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (n + r >= p.N) break;
            float x = v[r];
            if (p.bias) x += p.bias[n + r];
            x = apply_act(x, p.act);
            if (rp) x += rp[r];
            cp[r] = x;
          }
        }
      }
    } else {
      // W rows interleaved [gate x16 | up x16]: tile i even = gate, i odd = up, same lane
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const int n_in = n0 + wn * 64 + i * 16 + fg * 4;  // gate rows; up at n_in + 16
        if (n_in >= p.N) continue;
        const int n_out = (n0 >> 1) + wn * 32 + (i >> 1) * 16 + fg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float g = acc[i][j][r], u = acc[i + 1][j][r];
          if (p.bias) { g += p.bias[n_in + r]; u += p.bias[n_in + 16 + r]; }
          p.C[orow * p.ldc + n_out + r] = (g / (1.0f + expf(-g))) * u;
        }
      }
    }
  }
}

// Few rows (M <= 16: the [SEG] MLP, the two-way decoders' token-side projections and MLPs, hypernetwork / IoU / taxonomy
// heads at a handful of prompts): the product is a stream over W, and a 128x128 tile would run it on N/128 workgroups with
// one exposed memory round trip per 16-deep K step (1 x 4096 x 4096: 314 us, 4 x 256 x 2048: 160 us, the 256-wide token
// projections 24 us each). Here a workgroup owns NC output columns, its 4 waves take the 256-float K chunks round robin
// (lane l: floats 4l..4l+3 of a chunk — 1 KiB coalesced per row), every lane keeps MR x NC fp32 partial sums (fmaf chain),
// and the partials meet in a fixed order: lanes by xor-butterfly, then waves 0..3. No tile, no MFMA: HBM/L2-latency work.
template <int MR, int NC>
__global__ __launch_bounds__(256) void gemm_f32_skinny_kernel(GemmF32Args p) {
  __shared__ float red[4][MR][NC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * NC;
  const float* wrow[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) wrow[c] = p.W + (long)min(n0 + c, p.N - 1) * p.ldw;
  const float* arow[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) arow[m] = p.A + (long)min(m, p.M - 1) * p.lda;
  float acc[MR][NC];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[m][c] = 0.f;
  for (int k = wave * 256 + lane * 4; k < p.K; k += 1024) {
    float4 w[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) w[c] = *reinterpret_cast<const float4*>(wrow[c] + k);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const float4 a = *reinterpret_cast<const float4*>(arow[m] + k);
#pragma unroll
      for (int c = 0; c < NC; ++c)
        acc[m][c] = fmaf(a.w, w[c].w, fmaf(a.z, w[c].z, fmaf(a.y, w[c].y, fmaf(a.x, w[c].x, acc[m][c]))));
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float v = acc[m][c];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) red[wave][m][c] = v;
    }
  __syncthreads();
  const int t = threadIdx.x;
  if (t >= MR * NC) return;
  const int m = t / NC, c = t - m * NC, n = n0 + c;
  if (m >= p.M || n >= p.N) return;
  long orow = m;
  if (p.row_map) {
    orow = p.row_map[m];
    if (orow < 0) return;
  }
  float x = ((red[0][m][c] + red[1][m][c]) + red[2][m][c]) + red[3][m][c];
  if (p.bias) x += p.bias[n];
  x = apply_act(x, p.act);
  if (p.resid) x += p.resid[orow * p.ldr + n];
  p.C[orow * p.ldc + n] = x;
}

template <int MR>
static void launch_f32_skinny(const GemmF32Args& p, hipStream_t s) {
  // enough workgroups to cover the CUs on narrow outputs, 4 columns each once there are plenty (fewer A re-reads from L2)
  if (p.N >= 2048) hipLaunchKernelGGL((gemm_f32_skinny_kernel<MR, 4>), dim3((p.N + 3) / 4), dim3(256), 0, s, p);
  else if (p.N >= 512) hipLaunchKernelGGL((gemm_f32_skinny_kernel<MR, 2>), dim3((p.N + 1) / 2), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_f32_skinny_kernel<MR, 1>), dim3(p.N), dim3(256), 0, s, p);
}

}  // namespace

extern "C" int haff_gemm_f32(const float* A, long lda, const float* W, long ldw, float* C, long ldc,
                             const float* bias, const float* resid, long ldr, const int* row_map, int M, int N, int K,
                             int act, int swiglu, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 3) || (lda & 3) || (ldw & 3)) return HAFF_ERR_BAD_ARG;
  if (swiglu && ((N & 31) || resid)) return HAFF_ERR_BAD_ARG;
  GemmF32Args p{A, lda, W, ldw, C, ldc, bias, resid, ldr, row_map, M, N, K, act, swiglu, 0, 0, 0, 0, 0, 0, 0};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool al16 = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
  // every workgroup re-reads all M rows of A (L2): past 8 rows only while the weight is small enough not to care
  if (!swiglu && al16 && (M <= 8 || (M <= 16 && (long)N * K <= (1L << 21)))) {
    if (M <= 1) launch_f32_skinny<1>(p, s);
    else if (M <= 4) launch_f32_skinny<4>(p, s);
    else if (M <= 8) launch_f32_skinny<8>(p, s);
    else launch_f32_skinny<16>(p, s);
    return haff_check_launch();
  }
  const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, s, p);
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
