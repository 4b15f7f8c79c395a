// Rank-r adapter kernels of the LoRA fine-tune path (reference: peft LoraConfig on q_proj / v_proj, 2Haff/train_ds.py:192-230;
// the adapted projections feed LlamaAttention's rotate-half RoPE, model/llava/model/language_model/llava_llama.py:76-101).
//
//   forward    q = rope(x.Wq^T + s.(x.Aq^T).Bq^T),  k = rope(x.Wk^T),  v = x.Wv^T + s.(x.Av^T).Bv^T
//
// The frozen product x.[Wq;Wk;Wv]^T is the tile GEMM's; everything a rank-8 update adds is HBM-bound row traffic, and the
// generic route (four N = 8 / K = 8 products on the 128 x 128 tile, scale, add, two RoPE passes, and in backward three
// transposes per dW plus the zero-fill / copy / add chain of autograd's slice adjoints) cost ~0.9 ms per Llama layer and
// step — 14 % of the configs[3] step. Here:
//   * t^T = A2.x^T [16][M] (A2 = [Aq; Av]) is ONE weight-streaming launch of gemm_bf16.hip with the roles swapped (the
//     activation rows are the streamed operand);
//   * lora_qkv_rope_fwd_kernel reads q|k|v once, adds the rank-r update with one 75 %-empty MFMA per 16 x 16 tile (k = the
//     adapter index: the matrix pipe is idle on this path anyway), rotates q and k, writes the three attention operands;
//   * lora_qkv_rope_bwd_kernel assembles d(qkv) = [rope^T dq | rope^T dk | dv] in one pass (no slice adjoints);
//   * dt^T = B^T.dq^T is again the swapped weight-streaming product; lora_dx_kernel adds s.keep.(dt.A2) into the d(x) the
//     frozen product's adjoint wrote; lora_tn_kernel is the one contraction over ROWS (dA = dt^T.x, dB = dq^T.t): rank rows
//     in SGPRs, the big operand streamed once, per-row-block partials summed in index order (no atomics, deterministic).
// bf16 storage, fp32 arithmetic, head dim 128, rank <= 8 per adapter.
#include "haff_common.h"

namespace {

constexpr int HD = 128;   // head dim (Llama 7B / 13B)

__device__ __forceinline__ bf16x8 zero_frag() { return bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }

// the 16 x 32 operand tile of t (rows = activation rows row0.., k = adapter index 0..15, 16..31 empty) out of t^T [16][ldt]
__device__ __forceinline__ bf16x8 load_t_frag(const bf16_t* tT, long ldt, long row, int fh) {
  bf16x8 f = zero_frag();
  if (fh < 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (short)tT[(long)(8 * fh + i) * ldt + row];
  }
  return f;
}

struct LoraFwdArgs {
  const bf16_t* qkv; long ld_qkv;
  const bf16_t* tT; long ldt;
  const bf16_t *Bq, *Bv; int ldb;
  const float* cs;
  bf16_t *qo, *ko, *vo; long ldo;
  long M; int H, T; float scale;
};

// one wave = one head x 16-row tiles. The head's 128 columns are four groups of 32 (groups 2, 3 = the rotate-half partners of
// groups 0, 1); a group takes TWO MFMAs whose column operands are loaded in a permuted order (MFMA h, operand row i = column
// 8 * (i / 4) + 4 * h + i % 4), so that lane (fr, fh) ends up with row fr, columns 8fh .. 8fh+7 of the group: 16-byte
// loads and stores, 64 contiguous bytes per row and instruction (the natural operand order leaves 4 columns per lane:
// 8-byte accesses in 32-byte runs, which held this kernel at 2.4 TB/s)
__global__ __launch_bounds__(256, 3) void lora_qkv_rope_fwd_kernel(LoraFwdArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fh = lane >> 4;
  const int head = blockIdx.x;
  const long n_rt = (p.M + 15) / 16;
  bf16x8 bq[4][2], bv[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long col = (long)head * HD + 32 * g + 8 * (fr >> 2) + 4 * h + (fr & 3);
      bq[g][h] = fh == 0 ? *reinterpret_cast<const bf16x8*>(p.Bq + col * p.ldb) : zero_frag();
      bv[g][h] = fh == 1 ? *reinterpret_cast<const bf16x8*>(p.Bv + col * p.ldb) : zero_frag();
    }
  for (long rt = (long)blockIdx.y * 4 + wave; rt < n_rt; rt += (long)gridDim.y * 4) {
    const long row = rt * 16 + fr;
    const bool valid = row < p.M;
    const long rc = valid ? row : p.M - 1;
    const int pos = (int)(rc % p.T);
    const bf16x8 tt = load_t_frag(p.tT, p.ldt, rc, fh);
    const bf16_t* src = p.qkv + rc * p.ld_qkv + (long)head * HD + 8 * fh;
    bf16_t* q_o = p.qo + rc * p.ldo + (long)head * HD + 8 * fh;
    bf16_t* k_o = p.ko + rc * p.ldo + (long)head * HD + 8 * fh;
    bf16_t* v_o = p.vo + rc * p.ldo + (long)head * HD + 8 * fh;
    const float* csr = p.cs + (long)pos * HD + 8 * fh;
#pragma unroll
    for (int g = 0; g < 2; ++g) {   // column group g and its rotate-half partner g + 2
      float qa[8], qb[8], ka[8], kb[8], va[8], vb[8], co[8], si[8];
      load8(src + 32 * g, qa); load8(src + 64 + 32 * g, qb);
      load8(src + p.H + 32 * g, ka); load8(src + p.H + 64 + 32 * g, kb);
      load8(src + 2 * (long)p.H + 32 * g, va); load8(src + 2 * (long)p.H + 64 + 32 * g, vb);
      load8(csr + 32 * g, co); load8(csr + 64 + 32 * g, si);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 dq0[2], dq1[2], dv0[2], dv1[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        dq0[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[g][h], tt, z, 0, 0, 0);
        dq1[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[g + 2][h], tt, z, 0, 0, 0);
        dv0[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[g][h], tt, z, 0, 0, 0);
        dv1[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[g + 2][h], tt, z, 0, 0, 0);
      }
      float q1[8], q2[8], k1[8], k2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ql = qa[e] + p.scale * dq0[e >> 2][e & 3], qh = qb[e] + p.scale * dq1[e >> 2][e & 3];
        q1[e] = ql * co[e] - qh * si[e];
        q2[e] = qh * co[e] + ql * si[e];
        k1[e] = ka[e] * co[e] - kb[e] * si[e];
        k2[e] = kb[e] * co[e] + ka[e] * si[e];
        va[e] += p.scale * dv0[e >> 2][e & 3];
        vb[e] += p.scale * dv1[e >> 2][e & 3];
      }
      if (valid) {
        store8(q_o + 32 * g, q1); store8(q_o + 64 + 32 * g, q2);
        store8(k_o + 32 * g, k1); store8(k_o + 64 + 32 * g, k2);
        store8(v_o + 32 * g, va); store8(v_o + 64 + 32 * g, vb);
      }
    }
  }
}

// d(qkv) [M][3H] = [rope^T dq | rope^T dk | dv]: thread = (row, head, 8 low columns + their partners)
__global__ __launch_bounds__(256) void lora_qkv_rope_bwd_kernel(const bf16_t* dq, const bf16_t* dk, const bf16_t* dv, long ld_in,
                                                              const float* cs, bf16_t* dqkv, long ld_out, long M, int H, int T) {
  const int nh = H / HD;
  const long total = M * nh * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i & 7);
    const int head = (int)((i >> 3) % nh);
    const long row = (i >> 3) / nh;
    const int pos = (int)(row % T);
    const long col = (long)head * HD + 8 * ch;
    float co[8], si[8];
    load8(cs + (long)pos * HD + 8 * ch, co);
    load8(cs + (long)pos * HD + 64 + 8 * ch, si);
    float a[8], b[8], x1[8], x2[8];
    load8(dq + row * ld_in + col, a);
    load8(dq + row * ld_in + col + 64, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) { x1[e] = a[e] * co[e] + b[e] * si[e]; x2[e] = b[e] * co[e] - a[e] * si[e]; }
    store8(dqkv + row * ld_out + col, x1);
    store8(dqkv + row * ld_out + col + 64, x2);
    load8(dk + row * ld_in + col, a);
    load8(dk + row * ld_in + col + 64, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) { x1[e] = a[e] * co[e] + b[e] * si[e]; x2[e] = b[e] * co[e] - a[e] * si[e]; }
    store8(dqkv + row * ld_out + H + col, x1);
    store8(dqkv + row * ld_out + H + col + 64, x2);
    *reinterpret_cast<uint4*>(dqkv + row * ld_out + 2 * (long)H + col) = *reinterpret_cast<const uint4*>(dv + row * ld_in + col);
    *reinterpret_cast<uint4*>(dqkv + row * ld_out + 2 * (long)H + col + 64) = *reinterpret_cast<const uint4*>(dv + row * ld_in + col + 64);
  }
}

// dx[row][col] (+)= scale * keep[row][col] * sum_j dt[row][j] * A2[j][col]: wave = 128 columns x 16-row tiles
struct LoraDxArgs {
  const bf16_t* dtT; long ldt;
  const bf16_t* A2; long lda;
  const bf16_t* keep; long ldk;
  bf16_t* dx; long ldx;
  long M; int K; int accumulate; float scale;
  const bf16_t* keep_v;   // two masks (haff_lora_dx2): `keep` gates the q adapter's ranks (rows 0-7 of A2), keep_v the v adapter's (8-15)
};
__global__ __launch_bounds__(256) void lora_dx_kernel(LoraDxArgs p) {   // column operands permuted as in the forward kernel
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fh = lane >> 4;
  const long c0 = (long)blockIdx.x * 128;
  const long n_rt = (p.M + 15) / 16;
  bf16x8 af[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      af[g][h] = zero_frag();
      if (fh < 2) {
        const long col = c0 + 32 * g + 8 * (fr >> 2) + 4 * h + (fr & 3);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[g][h][i] = (short)p.A2[(long)(8 * fh + i) * p.lda + col];
      }
    }
  for (long rt = (long)blockIdx.y * 4 + wave; rt < n_rt; rt += (long)gridDim.y * 4) {
    const long row = rt * 16 + fr;
    const bool valid = row < p.M;
    const long rc = valid ? row : p.M - 1;
    const bf16x8 tt = load_t_frag(p.dtT, p.ldt, rc, fh);
    bf16_t* dst = p.dx + rc * p.ldx + c0 + 8 * fh;
    const bf16_t* kpr = p.keep ? p.keep + rc * p.ldk + c0 + 8 * fh : nullptr;
    const bf16_t* kvr = p.keep_v ? p.keep_v + rc * p.ldk + c0 + 8 * fh : nullptr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float old[8], kp[8], kv[8];
      if (p.accumulate) load8(dst + 32 * g, old);
      if (kpr) load8(kpr + 32 * g, kp);
      if (kvr) load8(kvr + 32 * g, kv);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 d[2];
      float v[8];
      if (kvr) {   // (wave-uniform) two masks: the q adapter's ranks sit in the fh = 0 lanes of the A fragment, the v adapter's in fh = 1
        const bf16x8 zf = zero_frag();
        f32x4 dv[2];
        d[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh == 0 ? af[g][0] : zf, tt, z, 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh == 0 ? af[g][1] : zf, tt, z, 0, 0, 0);
        dv[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh == 1 ? af[g][0] : zf, tt, z, 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh == 1 ? af[g][1] : zf, tt, z, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = p.scale * (d[e >> 2][e & 3] * kp[e] + dv[e >> 2][e & 3] * kv[e]);
          if (p.accumulate) v[e] += old[e];
        }
      } else {
        d[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g][0], tt, z, 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g][1], tt, z, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = p.scale * d[e >> 2][e & 3];
          if (kpr) v[e] *= kp[e];
          if (p.accumulate) v[e] += old[e];
        }
      }
      if (valid) store8(dst + 32 * g, v);
    }
  }
}

// part[rb][j][n] = sum over the row block's rows m of sT[j][m] * big[m][n]: workgroup = 128 columns (2 per lane) x one row
// block; its 4 waves take 8-row groups in turn (the R x 8 rank values of a group are wave-uniform: scalar loads), and meet in
// LDS in wave order
template <int R>
__global__ __launch_bounds__(256) void lora_tn_kernel(const bf16_t* sT, long lds, const bf16_t* big, long ldb, long M, int N,
                                                      int rows_per_block, float* part) {
  __shared__ float red[4][R][128];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long col = (long)blockIdx.x * 128 + 2 * lane;
  const long colc = col < N ? col : 0;   // N % 2 == 0 (checked by the launcher); columns past N are computed on column 0 and dropped
  const long r_lo = (long)blockIdx.y * rows_per_block;
  long r_hi = r_lo + rows_per_block;
  if (r_hi > M) r_hi = M;
  float acc[R][2];
#pragma unroll
  for (int j = 0; j < R; ++j) acc[j][0] = acc[j][1] = 0.f;
  // 16 rows per step (r_lo % 64 == 0, lds % 8 == 0 and lds >= roundup(M, 16): 32-B aligned rank rows), the next step's 16
  // loads of the big operand in flight while this step's 2 * 16 * R FMAs run: a wave has only a handful of steps, and
  // with one memory round trip per step the launch ran at a quarter of the HBM rate
  auto load_rows = [&](long m0, unsigned (&bw)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {   // rows past the block re-read its last row (their rank value is forced to 0 below)
      const long m = m0 + i < r_hi ? m0 + i : r_hi - 1;
      bw[i] = *reinterpret_cast<const unsigned*>(big + m * ldb + colc);
    }
  };
  unsigned bw[16], nx[16];
  long m0 = r_lo + 16 * wave;
  if (m0 < r_hi) load_rows(m0, bw);
  for (; m0 < r_hi; m0 += 64) {
    const bool more = m0 + 64 < r_hi;
    if (more) load_rows(m0 + 64, nx);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint4 s0 = *reinterpret_cast<const uint4*>(sT + (long)j * lds + m0);
      const uint4 s1 = *reinterpret_cast<const uint4*>(sT + (long)j * lds + m0 + 8);
      const unsigned sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned h = (i & 1) ? (sw[i >> 1] & 0xffff0000u) : (sw[i >> 1] << 16);
        const float sv = (m0 + i < r_hi) ? __uint_as_float(h) : 0.f;
        acc[j][0] = fmaf(sv, __uint_as_float(bw[i] << 16), acc[j][0]);
        acc[j][1] = fmaf(sv, __uint_as_float(bw[i] & 0xffff0000u), acc[j][1]);
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < 16; ++i) bw[i] = nx[i];
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    red[wave][j][2 * lane] = acc[j][0];
    red[wave][j][2 * lane + 1] = acc[j][1];
  }
  __syncthreads();
  float* dst = part + (long)blockIdx.y * R * N;
  for (int e = threadIdx.x; e < R * 128; e += 256) {
    const int j = e >> 7, c = e & 127;
    const long n = (long)blockIdx.x * 128 + c;
    if (n < N) dst[(long)j * N + n] = ((red[0][j][c] + red[1][j][c]) + red[2][j][c]) + red[3][j][c];
  }
}

// out = scale * sum over row blocks (index order) of part[rb][j][n]; rows j < j_valid only; [j][n] or transposed [n][j]
template <typename TO>
__global__ void lora_tn_reduce_kernel(const float* part, int nrb, int R, int N, TO* out, long ldo, int transposed, int j_valid,
                                      float scale) {
  const long total = (long)j_valid * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i / N);
    const long n = i - (long)j * N;
    float s = 0.f;
    for (int rb = 0; rb < nrb; ++rb) s += part[((long)rb * R + j) * N + n];
    s *= scale;
    elem<TO>::st(transposed ? out + n * ldo + j : out + (long)j * ldo + n, s);
  }
}

inline hipStream_t HS(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int check_launch() { return hipGetLastError() == hipSuccess ? HAFF_OK : HAFF_ERR_LAUNCH; }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int haff_lora_qkv_rope_fwd(const void* qkv, long ld_qkv, const void* tT, long ldt, const void* Bq, const void* Bv,
                                      int ldb, const float* cos_sin, void* q_out, void* k_out, void* v_out, long ldo, long M,
                                      int H, int d, int T, float scale, void* stream) {
  if (M <= 0 || T <= 0 || H <= 0 || !qkv || !tT || !Bq || !Bv || !cos_sin || !q_out || !k_out || !v_out) return HAFF_ERR_BAD_ARG;
  if (d != HD || H % HD || ldb != 8) return HAFF_ERR_UNSUPPORTED;
  if (ld_qkv < 3L * H || ldo < H || ldt < M || (ld_qkv & 7) || (ldo & 7)) return HAFF_ERR_BAD_ARG;
  if (!al16(qkv) || !al16(Bq) || !al16(Bv) || !al16(cos_sin) || !al16(q_out) || !al16(k_out) || !al16(v_out)) return HAFF_ERR_BAD_ARG;
  LoraFwdArgs p{(const bf16_t*)qkv, ld_qkv, (const bf16_t*)tT, ldt, (const bf16_t*)Bq, (const bf16_t*)Bv, ldb, cos_sin,
                (bf16_t*)q_out, (bf16_t*)k_out, (bf16_t*)v_out, ldo, M, H, T, scale};
  const long n_rt = (M + 15) / 16;
  long gy = (n_rt + 3) / 4;
  const int nh = H / HD;
  const long cap = (2048 + nh - 1) / nh;   // ~2048 workgroups: 8 per CU
  if (gy > cap) gy = cap;
  hipLaunchKernelGGL(lora_qkv_rope_fwd_kernel, dim3(nh, (unsigned)gy), dim3(256), 0, HS(stream), p);
  return check_launch();
}

extern "C" int haff_lora_qkv_rope_bwd(const void* dq, const void* dk, const void* dv, long ld_in, const float* cos_sin, void* dqkv,
                                      long ld_out, long M, int H, int d, int T, void* stream) {
  if (M <= 0 || T <= 0 || H <= 0 || !dq || !dk || !dv || !cos_sin || !dqkv) return HAFF_ERR_BAD_ARG;
  if (d != HD || H % HD) return HAFF_ERR_UNSUPPORTED;
  if (ld_in < H || ld_out < 3L * H || (ld_in & 7) || (ld_out & 7)) return HAFF_ERR_BAD_ARG;
  if (!al16(dq) || !al16(dk) || !al16(dv) || !al16(cos_sin) || !al16(dqkv)) return HAFF_ERR_BAD_ARG;
  const long total = M * (H / HD) * 8;
  long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(lora_qkv_rope_bwd_kernel, dim3((unsigned)g), dim3(256), 0, HS(stream), (const bf16_t*)dq, (const bf16_t*)dk,
                     (const bf16_t*)dv, ld_in, cos_sin, (bf16_t*)dqkv, ld_out, M, H, T);
  return check_launch();
}

static int lora_dx_launch(const void* dtT, long ldt, const void* A2, long lda, const void* keep, const void* keep_v, long ldk, void* dx,
                          long ldx, int accumulate, long M, int K, float scale, void* stream) {
  if (M <= 0 || K <= 0 || !dtT || !A2 || !dx || (keep_v && !keep)) return HAFF_ERR_BAD_ARG;
  if (K % 128) return HAFF_ERR_UNSUPPORTED;
  if (ldt < M || lda < K || ldx < K || (ldx & 7) || (keep && (ldk < K || (ldk & 7)))) return HAFF_ERR_BAD_ARG;
  if (!al16(dx) || (keep && !al16(keep)) || (keep_v && !al16(keep_v))) return HAFF_ERR_BAD_ARG;
  LoraDxArgs p{(const bf16_t*)dtT, ldt, (const bf16_t*)A2, lda, (const bf16_t*)keep, ldk, (bf16_t*)dx, ldx, M, K, accumulate, scale,
               (const bf16_t*)keep_v};
  const long n_rt = (M + 15) / 16;
  long gy = (n_rt + 3) / 4;
  const int gx = K / 128;
  const long cap = (2048 + gx - 1) / gx;
  if (gy > cap) gy = cap;
  hipLaunchKernelGGL(lora_dx_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, HS(stream), p);
  return check_launch();
}

extern "C" int haff_lora_dx(const void* dtT, long ldt, const void* A2, long lda, const void* keep, long ldk, void* dx, long ldx,
                            int accumulate, long M, int K, float scale, void* stream) {
  return lora_dx_launch(dtT, ldt, A2, lda, keep, nullptr, ldk, dx, ldx, accumulate, M, K, scale, stream);
}

// haff_lora_dx with TWO dropout masks (peft: one lora_dropout module per adapted Linear, 2Haff/train_ds.py:218-230):
//   dx (+)= scale * ( keep_q o (dt[0:8]^T . A2[0:8]) + keep_v o (dt[8:16]^T . A2[8:16]) ),   both masks bf16 [M][ldk] of values.
extern "C" int haff_lora_dx2(const void* dtT, long ldt, const void* A2, long lda, const void* keep_q, const void* keep_v, long ldk,
                             void* dx, long ldx, int accumulate, long M, int K, float scale, void* stream) {
  if (!keep_q || !keep_v) return HAFF_ERR_BAD_ARG;
  return lora_dx_launch(dtT, ldt, A2, lda, keep_q, keep_v, ldk, dx, ldx, accumulate, M, K, scale, stream);
}

// rows per block of haff_lora_tn (multiple of 64): ~16 row blocks
static long lora_tn_rows_per_block(long M) {
  long rpb = ((M + 15) / 16 + 63) / 64 * 64;
  return rpb < 64 ? 64 : rpb;
}
static long lora_tn_ws(long M, int R, int N) {
  const long rpb = lora_tn_rows_per_block(M);
  return ((M + rpb - 1) / rpb) * (long)R * N;
}
extern "C" int haff_lora_tn_workspace_elems(long M, int R, int N) {   // f32 values haff_lora_tn wants; < 0: does not fit an int
  const long n = (M > 0 && R > 0 && N > 0) ? lora_tn_ws(M, R, N) : -1;
  return n > 0x7fffffffL ? HAFF_ERR_UNSUPPORTED : (int)n;
}
extern "C" int haff_lora_tn(const void* sT, long lds, int R, const void* big, long ldb, long M, int N, float* workspace,
                            long workspace_elems, void* out, long ldo, int out_f32, int transposed, int j_valid, float scale,
                            void* stream) {
  if (M <= 0 || N <= 0 || !sT || !big || !workspace || !out || j_valid <= 0 || j_valid > R) return HAFF_ERR_BAD_ARG;
  if (R != 8 && R != 16) return HAFF_ERR_UNSUPPORTED;
  if ((lds & 7) || lds < (M + 15) / 16 * 16 || !al16(sT) || (N & 1) || (ldb & 1) || ldb < N || (reinterpret_cast<uintptr_t>(big) & 3))
    return HAFF_ERR_BAD_ARG;
  if (workspace_elems < lora_tn_ws(M, R, N)) return HAFF_ERR_BAD_ARG;
  if (ldo < (transposed ? j_valid : N)) return HAFF_ERR_BAD_ARG;
  const long rpb = lora_tn_rows_per_block(M);
  const int nrb = (int)((M + rpb - 1) / rpb);
  const dim3 g((N + 127) / 128, nrb), b(256);
  if (R == 8)
    hipLaunchKernelGGL((lora_tn_kernel<8>), g, b, 0, HS(stream), (const bf16_t*)sT, lds, (const bf16_t*)big, ldb, M, N, (int)rpb, workspace);
  else
    hipLaunchKernelGGL((lora_tn_kernel<16>), g, b, 0, HS(stream), (const bf16_t*)sT, lds, (const bf16_t*)big, ldb, M, N, (int)rpb, workspace);
  const long total = (long)j_valid * N;
  long gr = (total + 255) / 256;
  if (gr > 4096) gr = 4096;
  if (out_f32)
    hipLaunchKernelGGL((lora_tn_reduce_kernel<float>), dim3((unsigned)gr), b, 0, HS(stream), workspace, nrb, R, N, (float*)out, ldo, transposed, j_valid, scale);
  else
    hipLaunchKernelGGL((lora_tn_reduce_kernel<bf16_t>), dim3((unsigned)gr), b, 0, HS(stream), workspace, nrb, R, N, (bf16_t*)out, ldo, transposed, j_valid, scale);
  return check_launch();
}
