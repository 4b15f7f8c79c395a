// Backward / loss / optimiser kernels of the LoRA fine-tune path (reference: 2Haff/train_ds.py:489-622 driving
// LISAForCausalLM.model_forward, 2Haff/model/LISA.py:175-430). Contractions reuse the forward GEMM kernels
// (dX = dY.W via a transposed weight copy, dW = dY^T.X via transposed operands, attention-shaped products via the
// batched entry points); everything here is the HBM-bound remainder: transposes, activation / norm / softmax /
// RoPE adjoints, fused cross-entropy, mask losses, bilinear adjoint, embedding scatter-add, AdamW.
// All kernels are templated on the storage type (bf16 / f32) and compute in fp32.
#include "haff_common.h"

namespace {

inline int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// ---- batched 2-D transpose: in [nb][R][ld_in] (C valid columns) -> out [nb][Cp][Rp], zero padded ---------------
template <typename T>
__global__ void transpose_kernel(const T* in, long ld_in, long s_in_o, long s_in_i, T* out, int R, int C, int Rp, int Cp,
                                 int nb_inner) {
  __shared__ float tile[32][33];
  const int z = blockIdx.z;
  const int zo = z / nb_inner, zi = z - zo * nb_inner;
  const T* src = in + zo * s_in_o + zi * s_in_i;
  T* dst = out + (long)z * Cp * Rp;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < C) ? elem<T>::ld(src + (long)r * ld_in + c) : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < Cp && r < Rp) elem<T>::st(dst + (long)c * Rp + r, tile[threadIdx.x][i]);
  }
}

// ---- activations -----------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case HAFF_ACT_GELU: {
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case HAFF_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case HAFF_ACT_SILU: { const float s = 1.0f / (1.0f + expf(-x)); return s * (1.0f + x * (1.0f - s)); }
    case HAFF_ACT_QUICK_GELU: { const float s = 1.0f / (1.0f + expf(-1.702f * x)); return s * (1.0f + 1.702f * x * (1.0f - s)); }
    default: return 1.f;
  }
}
template <typename T>
__global__ void act_fwd_kernel(const T* x, T* y, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    elem<T>::st(y + i, apply_act(elem<T>::ld(x + i), act));
}
template <typename T>
__global__ void act_bwd_kernel(const T* x, const T* dy, T* dx, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    elem<T>::st(dx + i, elem<T>::ld(dy + i) * act_grad(elem<T>::ld(x + i), act));
}
// gu [M][2F] interleaved in 16-column groups [gate x16 | up x16] -> y [M][F] = silu(g) * u ; and its adjoint
template <typename T>
__global__ void swiglu_fwd_kernel(const T* gu, T* y, long M, int F) {
  const long n = M * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long m = i / F;
    const int j = (int)(i - m * F);
    const long base = m * 2 * F + (j >> 4) * 32 + (j & 15);
    const float g = elem<T>::ld(gu + base), u = elem<T>::ld(gu + base + 16);
    elem<T>::st(y + i, g / (1.0f + expf(-g)) * u);
  }
}
template <typename T>
__global__ void swiglu_bwd_kernel(const T* gu, const T* dy, T* dgu, long M, int F) {
  const long n = M * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long m = i / F;
    const int j = (int)(i - m * F);
    const long base = m * 2 * F + (j >> 4) * 32 + (j & 15);
    const float g = elem<T>::ld(gu + base), u = elem<T>::ld(gu + base + 16), d = elem<T>::ld(dy + i);
    const float s = 1.0f / (1.0f + expf(-g));
    elem<T>::st(dgu + base, d * u * s * (1.0f + g * (1.0f - s)));
    elem<T>::st(dgu + base + 16, d * g * s);
  }
}
template <typename T>
__global__ void axpby_kernel(const T* a, const T* b, T* out, long n, float alpha, float beta) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    elem<T>::st(out + i, alpha * elem<T>::ld(a + i) + (b ? beta * elem<T>::ld(b + i) : 0.f));
}

// out[r][c] = a[r][c] * alpha[r * alpha_stride]  (alpha: DEVICE fp32; alpha_stride 0 = one scalar for the whole tensor)
template <typename T>
__global__ void scale_dev_kernel(const T* a, T* out, long rows, long cols, const float* alpha, long alpha_stride) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    elem<T>::st(out + i, elem<T>::ld(a + i) * alpha[(i / cols) * alpha_stride]);
}

template <typename T>
__global__ void mul_kernel(const T* a, const T* b, T* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    elem<T>::st(out + i, elem<T>::ld(a + i) * elem<T>::ld(b + i));
}

// bf16 forms with 16-byte accesses: a thread owns 8 consecutive outputs (half a 16-column group)
__global__ void swiglu_fwd_vec_kernel(const bf16_t* gu, bf16_t* y, long M, int F) {
  const long n8 = M * F / 8;
  const int f8 = F >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long m = i / f8;
    const int j = (int)(i - m * f8) * 8;
    const long base = m * 2 * F + (j >> 4) * 32 + (j & 15);
    float g[8], u[8], o[8];
    load8(gu + base, g);
    load8(gu + base + 16, u);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = g[e] / (1.0f + expf(-g[e])) * u[e];
    store8(y + m * F + j, o);
  }
}
__global__ void swiglu_bwd_vec_kernel(const bf16_t* gu, const bf16_t* dy, bf16_t* dgu, long M, int F) {
  const long n8 = M * F / 8;
  const int f8 = F >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long m = i / f8;
    const int j = (int)(i - m * f8) * 8;
    const long base = m * 2 * F + (j >> 4) * 32 + (j & 15);
    float g[8], u[8], d[8], dg[8], du[8];
    load8(gu + base, g);
    load8(gu + base + 16, u);
    load8(dy + m * F + j, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sg = 1.0f / (1.0f + expf(-g[e]));
      dg[e] = d[e] * u[e] * sg * (1.0f + g[e] * (1.0f - sg));
      du[e] = d[e] * g[e] * sg;
    }
    store8(dgu + base, dg);
    store8(dgu + base + 16, du);
  }
}

// ---- norm adjoints: one wave per row ---------------------------------------------------------------------------
// LayerNorm: xhat = (x-mean)*rstd, g = dy*w ; dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) ; dyx = dy*xhat (f32, for dw)
// RMSNorm  : xhat = x*rstd ;            dx = rstd*(g - xhat*mean(g*xhat))
template <typename T, bool RMS>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const T* x, const T* dy, const float* w, T* dx, float* dyx,
                                                      int rows, int C, float eps, const T* add) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (long)row * C;
  const T* dr = dy + (long)row * C;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = elem<T>::ld(xr + c); s1 += v; s2 += v * v; }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  float mean = 0.f, rstd;
  if (RMS) rstd = 1.0f / sqrtf(s2 / C + eps);
  else {
    mean = s1 / C;
    float sq = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = elem<T>::ld(xr + c) - mean; sq += d * d; }
    sq = wave_sum(sq);
    rstd = 1.0f / sqrtf(sq / C + eps);
  }
  float a = 0.f, b = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float xh = (elem<T>::ld(xr + c) - mean) * rstd;
    const float g = elem<T>::ld(dr + c) * w[c];
    a += g; b += g * xh;
  }
  a = wave_sum(a) / C; b = wave_sum(b) / C;
  for (int c = lane; c < C; c += 64) {
    const float xh = (elem<T>::ld(xr + c) - mean) * rstd;
    const float d = elem<T>::ld(dr + c);
    const float g = d * w[c];
    // add: the gradient that reaches the SAME tensor along the residual branch (haff_norm_bwd_add): summed here in fp32
    elem<T>::st(dx + (long)row * C + c, rstd * (g - (RMS ? 0.f : a) - xh * b) + (add ? elem<T>::ld(add + (long)row * C + c) : 0.f));
    if (dyx) dyx[(long)row * C + c] = d * xh;
  }
}

// the same for bf16 rows of C = 512 * NV elements (Llama hidden sizes): the row and its gradient live in registers as 16-byte
// loads — ONE pass over HBM; the generic kernel above walks the row four times with 2-byte loads (74 us for 2808 x 4096
// against 15 us of traffic at the HBM rate: 2.6 % of the fine-tune step)
template <bool RMS, int NV>
__global__ __launch_bounds__(256) void norm_bwd_vec_kernel(const bf16_t* x, const bf16_t* dy, const float* w, bf16_t* dx, float* dyx,
                                                          int rows, float eps, const bf16_t* add) {
  constexpr int C = 512 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (long)row * C;
  const bf16_t* dr = dy + (long)row * C;
  uint4 xv[NV], dv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    xv[i] = *reinterpret_cast<const uint4*>(xr + (i * 64 + lane) * 8);
    dv[i] = *reinterpret_cast<const uint4*>(dr + (i * 64 + lane) * 8);
  }
  auto unpack = [](const uint4& r, float (&v)[8]) {
    const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(u[e] << 16); v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
  };
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float v[8];
    unpack(xv[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1 += v[e]; s2 += v[e] * v[e]; }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  float mean = 0.f, rstd;
  if (RMS) rstd = 1.0f / sqrtf(s2 / C + eps);
  else {
    mean = s1 / C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float v[8];
      unpack(xv[i], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; sq += d * d; }
    }
    sq = wave_sum(sq);
    rstd = 1.0f / sqrtf(sq / C + eps);
  }
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float v[8], d[8], wv[8];
    unpack(xv[i], v);
    unpack(dv[i], d);
    load8(w + (i * 64 + lane) * 8, wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float g = d[e] * wv[e]; a += g; b += g * ((v[e] - mean) * rstd); }
  }
  a = wave_sum(a) / C; b = wave_sum(b) / C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float v[8], d[8], wv[8], o[8];
    unpack(xv[i], v);
    unpack(dv[i], d);
    load8(w + (i * 64 + lane) * 8, wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - mean) * rstd;
      o[e] = rstd * (d[e] * wv[e] - (RMS ? 0.f : a) - xh * b);
      d[e] *= xh;
    }
    if (add) {   // (wave-uniform) the residual branch's gradient of the same tensor
      float ad[8];
      load8(add + (long)row * C + (i * 64 + lane) * 8, ad);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += ad[e];
    }
    store8(dx + (long)row * C + (i * 64 + lane) * 8, o);
    if (dyx) store8(dyx + (long)row * C + (i * 64 + lane) * 8, d);
  }
}

// column sums of x [R][C] -> out f32 [C] (bias / LN weight gradients); out must be zeroed by the caller
template <typename T>
__global__ void colsum_kernel(const T* x, float* out, long R, int C, long rows_per_block) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(r0 + rows_per_block, R);
  float acc = 0.f;
  for (long r = r0; r < r1; ++r) acc += elem<T>::ld(x + r * C + c);
  atomicAdd(out + c, acc);
}

// bf16 rows of C = 8 * cpr columns with cpr a power of two <= 256 (decoder widths 128 / 256, hidden sizes up to 2048): a thread
// owns 8 columns (16-byte loads) of every (1024 / cpr)-th row of its row block, eight rows in flight; the row groups meet in
// LDS in index order and group 0 adds the block's sums to out. At most 64 blocks of 16 waves: with one block per 256 rows
// the 65 536 x 256 image-token bias gradients of the mask decoders spent their time in 256 atomics per output address
// (38 us, 0.9 TB/s); the 2-byte one-column-per-thread walk above ran them at 0.7 TB/s
__global__ __launch_bounds__(1024) void colsum_vec_kernel(const bf16_t* x, float* out, long R, int C, long rows_per_block) {
  __shared__ float red[1024][9];
  const int cpr = C >> 3, rpp = 1024 / cpr;
  const int ch = threadIdx.x % cpr, rg = threadIdx.x / cpr;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(r0 + rows_per_block, R);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  long r = r0 + rg;
  for (; r + 7L * rpp < r1; r += 8L * rpp) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4*>(x + (r + (long)u * rpp) * C + 8 * ch);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(w[e] << 16); acc[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
    }
  }
  for (; r < r1; r += rpp) {
    float v[8];
    load8(x + r * C + 8 * ch, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if (rg == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = acc[e];
      for (int g = 1; g < rpp; ++g) t += red[g * cpr + ch][e];
      atomicAdd(out + 8 * ch + e, t);
    }
  }
}

// ---- attention pieces ------------------------------------------------------------------------------------------
// scores f32 [rows][ld] -> P (T) [rows][ldp]: softmax(scale*s + causal mask) over the first Nk columns, zeros beyond.
// row r belongs to query (r % Nq); causal: key j visible iff j <= q + q_pos0.
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* s, long ld, T* p, long ldp, long rows, int Nq, int Nk,
                                                         float scale, int causal, int q_pos0) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int q = (int)(row % Nq);
  const int lim = causal ? min(Nk, q + q_pos0 + 1) : Nk;
  const float* sr = s + row * ld;
  float m = -INFINITY;
  for (int j = lane; j < lim; j += 64) m = fmaxf(m, sr[j] * scale);
  m = wave_max(m);
  float sum = 0.f;
  for (int j = lane; j < lim; j += 64) sum += expf(sr[j] * scale - m);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int j = lane; j < ldp; j += 64) elem<T>::st(p + row * ldp + j, j < lim ? expf(sr[j] * scale - m) * inv : 0.f);
}
// dS = scale * P o (dP - rowsum(dP o P)) ; dP f32 [rows][ld], P (T) [rows][ldp] -> dS (T) [rows][ldp]
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* p, long ldp, const float* dp, long ld, T* ds, long rows,
                                                         int Nk, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int j = lane; j < Nk; j += 64) dot += elem<T>::ld(p + row * ldp + j) * dp[row * ld + j];
  dot = wave_sum(dot);
  for (int j = lane; j < ldp; j += 64) {
    const float pv = j < Nk ? elem<T>::ld(p + row * ldp + j) : 0.f;
    elem<T>::st(ds + row * ldp + j, j < Nk ? scale * pv * (dp[row * ld + j] - dot) : 0.f);
  }
}
// rotate-half RoPE on x [rows][H][d] (row stride ld), position = pos0 + (row % T); sign = +1 forward, -1 adjoint
template <typename T>
__global__ void rope_kernel(const T* x, long ldx, T* y, long ldy, const float* cs, long rows, int Tlen, int H, int d, int pos0,
                            float sign) {
  const int half = d / 2;
  const long total = rows * H * half;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % half);
    const int h = (int)((i / half) % H);
    const long r = i / ((long)half * H);
    const int pos = pos0 + (int)(r % Tlen);
    const float co = cs[(long)pos * d + c], si = sign * cs[(long)pos * d + half + c];
    const float x1 = elem<T>::ld(x + r * ldx + h * d + c), x2 = elem<T>::ld(x + r * ldx + h * d + half + c);
    elem<T>::st(y + r * ldy + h * d + c, x1 * co - x2 * si);
    elem<T>::st(y + r * ldy + h * d + half + c, x2 * co + x1 * si);
  }
}

// ---- losses ----------------------------------------------------------------------------------------------------
// shift-by-one CE (llava_llama.py:108-118): row r of logits predicts labels[r]; ignore_index -100; mean over valid rows.
// Writes per-row loss (f32) and, if dlogits, (softmax - onehot) * gscale for valid rows (zeros otherwise).
template <typename T>
__global__ __launch_bounds__(256) void cross_entropy_kernel(const T* logits, long ld, const long* labels, float* row_loss,
                                                           T* dlogits, int V, float gscale) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const T* lr = logits + row * ld;
  const long lab = labels[row];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < V; j += 256) m = fmaxf(m, elem<T>::ld(lr + j));
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = threadIdx.x; j < V; j += 256) sum += expf(elem<T>::ld(lr + j) - m);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  const float lse = m + logf(sum);
  const bool valid = lab >= 0;
  if (threadIdx.x == 0) row_loss[row] = valid ? lse - elem<T>::ld(lr + lab) : 0.f;
  if (dlogits) {
    T* dr = dlogits + row * ld;
    for (int j = threadIdx.x; j < V; j += 256) {
      float g = 0.f;
      if (valid) g = (expf(elem<T>::ld(lr + j) - lse) - (j == lab ? 1.f : 0.f)) * gscale;
      elem<T>::st(dr + j, g);
    }
  }
}

// per-sample mask losses (LISA.py:16-59) on logits x (f32) scaled by wgt, targets t (f32), n pixels:
//   bce = mean(BCEWithLogits(w*x, t)) ; dice = 1 - (2*sum(p*t)/1000 + eps)/(sum(p)/1000 + sum(t)/1000 + eps), p = sigmoid(w*x)
// pass 1 accumulates [bce_sum, sum_pt, sum_p, sum_t] per sample (f32[4], zeroed by caller);
// pass 2 writes dL/dx = w * (c_bce * (p - t)/n + c_dice * d dice/dz).
__global__ void mask_loss_stats_kernel(const float* x, const float* t, float* stats, long n, float wgt) {
  const long s = blockIdx.y;
  const float* xs = x + s * n;
  const float* ts = t + s * n;
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float z = wgt * xs[i], tt = ts[i];
    const float p = 1.0f / (1.0f + expf(-z));
    a += fmaxf(z, 0.f) - z * tt + log1pf(expf(-fabsf(z)));
    b += p * tt; c += p; d += tt;
  }
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(stats + s * 4 + 0, a); atomicAdd(stats + s * 4 + 1, b);
    atomicAdd(stats + s * 4 + 2, c); atomicAdd(stats + s * 4 + 3, d);
  }
}
// coef (device, f32 [n_samples][2], may be null): the upstream gradients of {bce, dice}; they multiply c_bce / c_dice, so the
// caller never has to read them back to the host (a read-back at the head of backward drains the launch queue)
__global__ void mask_loss_grad_kernel(const float* x, const float* t, const float* stats, float* dx, long n, float wgt,
                                      float c_bce, float c_dice, const float* coef) {
  const long s = blockIdx.y;
  if (coef) { c_bce *= coef[2 * s]; c_dice *= coef[2 * s + 1]; }
  const float sum_pt = stats[s * 4 + 1], sum_p = stats[s * 4 + 2], sum_t = stats[s * 4 + 3];
  const float num = 2.f * sum_pt / 1000.f + 1e-6f, den = sum_p / 1000.f + sum_t / 1000.f + 1e-6f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float z = wgt * x[s * n + i], tt = t[s * n + i];
    const float p = 1.0f / (1.0f + expf(-z));
    // d(1 - num/den)/dp = -(2*t/1000)/den + num/(den*den)/1000
    const float ddice_dp = -(2.f * tt / 1000.f) / den + num / (den * den) / 1000.f;
    dx[s * n + i] = wgt * (c_bce * (p - tt) / (float)n + c_dice * ddice_dp * p * (1.f - p));
  }
}

// adjoint of haff_resize_bilinear: din [N][Hs][Ws] (zeroed by the caller; only the Hc x Wc crop receives gradient) from dout
// [N][Ho][Wo]. A GATHER: one thread per source pixel walks the output rows / columns whose two taps can touch it, recomputing
// the forward's tap indices and weights for each (so border clamping is the forward's, bit for bit), and sums in a fixed order —
// no atomics (the scatter form issued 4 per output pixel, 64 queueing on every source address at 4x upsampling: 52 us for a
// 1024^2 mask), repeatable to the bit.
__device__ __forceinline__ void bilinear_taps(int o, float scale, int Lc, int& i0, int& i1, float& w0, float& w1) {
  float f = scale * ((float)o + 0.5f) - 0.5f;
  f = f < 0.f ? 0.f : f;
  i0 = (int)f;
  i0 = i0 > Lc - 1 ? Lc - 1 : i0;
  i1 = i0 + (i0 < Lc - 1 ? 1 : 0);
  w1 = f - (float)i0;
  w0 = 1.f - w1;
}
__global__ void resize_bilinear_bwd_kernel(const float* dout, float* din, int N, int Hs, int Ws, int Hc, int Wc, int Ho, int Wo) {
  const float sh = (float)Hc / (float)Ho, sw = (float)Wc / (float)Wo;
  const long total = (long)N * Hc * Wc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % Wc), iy = (int)((i / Wc) % Hc);
    const long n = i / ((long)Wc * Hc);
    // output rows whose first tap is iy - 1 or iy (plus one row of slack each side for rounding, filtered exactly below)
    int oy_lo = (int)floorf(((float)iy - 0.5f) / sh - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / sh - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / sw - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / sw - 0.5f) + 1;
    if (iy == 0) oy_lo = 0;            // clamped taps: every row above the first sample maps to 0
    if (ix == 0) ox_lo = 0;
    if (iy == Hc - 1) oy_hi = Ho - 1;  // and every row below the last one to Hc - 1
    if (ix == Wc - 1) ox_hi = Wo - 1;
    oy_lo = oy_lo < 0 ? 0 : oy_lo; oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi;
    ox_lo = ox_lo < 0 ? 0 : ox_lo; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
    const float* g = dout + n * (long)Ho * Wo;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float hy, ly;
      bilinear_taps(oy, sh, Hc, y0, y1, hy, ly);
      const float wy = (y0 == iy ? hy : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f && y0 != iy && y1 != iy) continue;
      float row = 0.f;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float hx, lx;
        bilinear_taps(ox, sw, Wc, x0, x1, hx, lx);
        const float wx = (x0 == ix ? hx : 0.f) + (x1 == ix ? lx : 0.f);
        row += wx * g[(long)oy * Wo + ox];
      }
      acc += wy * row;
    }
    din[n * (long)Hs * Ws + (long)iy * Ws + ix] = acc;
  }
}

// embedding gradient: dE[ids[r]] += dx[r] (f32 accumulator, zeroed by caller); rows with ids < 0 are skipped
template <typename T>
__global__ void scatter_add_rows_kernel(const long* ids, const T* dx, float* dE, long rows, int C) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const long id = ids[r];
    if (id >= 0) atomicAdd(dE + id * C + (i - r * C), elem<T>::ld(dx + i));
  }
}

// taxonomy loss (LISA.py:151,414-417; mask_decoder.py:177): CrossEntropyLoss applied to the ALREADY soft-maxed
// class probabilities p = softmax(z) with a soft target t: loss = -sum_c t_c * log_softmax(p)_c.
// One thread per row of C <= 8 classes; writes loss and dL/dz (for upstream gradient 1).
__global__ void taxonomy_ce_kernel(const float* z, const float* t, float* probs, float* loss, float* dz, int rows, int C) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float p[8], q[8];
  float m = -INFINITY;
  for (int c = 0; c < C; ++c) m = fmaxf(m, z[r * C + c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) { p[c] = expf(z[r * C + c] - m); s += p[c]; }
  for (int c = 0; c < C; ++c) { p[c] /= s; if (probs) probs[r * C + c] = p[c]; }
  float m2 = -INFINITY;
  for (int c = 0; c < C; ++c) m2 = fmaxf(m2, p[c]);
  float s2 = 0.f;
  for (int c = 0; c < C; ++c) { q[c] = expf(p[c] - m2); s2 += q[c]; }
  const float lse2 = m2 + logf(s2);
  float l = 0.f, tsum = 0.f;
  for (int c = 0; c < C; ++c) { l -= t[r * C + c] * (p[c] - lse2); tsum += t[r * C + c]; }
  if (loss) loss[r] = l;
  if (dz) {
    float gp[8], dot = 0.f;
    for (int c = 0; c < C; ++c) { gp[c] = -t[r * C + c] + tsum * q[c] / s2; dot += gp[c] * p[c]; }
    for (int c = 0; c < C; ++c) dz[r * C + c] = p[c] * (gp[c] - dot);
  }
}

// ---- ordered reductions (round 4) ----------------------------------------------------------------------------------
// The fp32 atomicAdd forms above (column sums, loss statistics, the embedding scatter, the gradient norm) add in whatever order
// the blocks finish: a repeated fine-tune step differed by up to 1.4e-2 of a gradient tensor's scale. The forms below have no
// atomics: every block writes its partial result (threads stride in a fixed pattern, wave_sum and the cross-wave sum are fixed
// trees), and reduce_partials_kernel adds a block's partials in index order — bitwise repeatable for a given launch geometry.
__device__ __forceinline__ float block_sum_256(float a, float* red) {   // every thread of a 256-thread block calls; result in thread 0
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  const float s = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return s;
}
template <typename T>
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const T* g, float* partials, long n) {
  __shared__ float red[4];
  float a = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = elem<T>::ld(g + i);
    a += v * v;
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = a;
}
__global__ __launch_bounds__(256) void mask_loss_stats_partials_kernel(const float* x, const float* t, float* partials, long n, float wgt) {
  __shared__ float red[4];
  const long s = blockIdx.y;
  const float* xs = x + s * n;
  const float* ts = t + s * n;
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float z = wgt * xs[i], tt = ts[i];
    const float p = 1.0f / (1.0f + expf(-z));
    a += fmaxf(z, 0.f) - z * tt + log1pf(expf(-fabsf(z)));
    b += p * tt; c += p; d += tt;
  }
  a = block_sum_256(a, red); b = block_sum_256(b, red); c = block_sum_256(c, red); d = block_sum_256(d, red);
  if (threadIdx.x == 0) {
    float* o = partials + (s * gridDim.x + blockIdx.x) * 4;
    o[0] = a; o[1] = b; o[2] = c; o[3] = d;
  }
}
// column sums of a row block into partials[blockIdx.y][C]; a thread owns one column (coalesced rows), rows in index order
template <typename T>
__global__ void colsum_partials_kernel(const T* x, float* partials, long R, int C, long rows_per_block) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(r0 + rows_per_block, R);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four independent chains (fixed assignment: row index mod 4)
  long r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += elem<T>::ld(x + r * C + c); a1 += elem<T>::ld(x + (r + 1) * C + c);
    a2 += elem<T>::ld(x + (r + 2) * C + c); a3 += elem<T>::ld(x + (r + 3) * C + c);
  }
  for (; r < r1; ++r) a0 += elem<T>::ld(x + r * C + c);
  partials[(long)blockIdx.y * C + c] = (a0 + a1) + (a2 + a3);
}
// out[j] (+)= sum over p < n_parts, in index order, of partials[(j / out_inner) * group_stride + p * part_stride + j % out_inner].
// One wave per output: lane l adds parts l, l + 64, ... then a fixed wave_sum tree.
__global__ __launch_bounds__(64) void reduce_partials_kernel(const float* partials, float* out, int n_out, int n_parts, int out_inner,
                                                             long part_stride, long group_stride, int accumulate) {
  const int j = blockIdx.x;
  if (j >= n_out) return;
  const float* base = partials + (long)(j / out_inner) * group_stride + (j % out_inner);
  float a = 0.f;
  for (int p = threadIdx.x; p < n_parts; p += 64) a += base[(long)p * part_stride];
  a = wave_sum(a);
  if (threadIdx.x == 0) out[j] = accumulate ? out[j] + a : a;
}
// embedding gradient without atomics: rows visited in the order of a STABLE sort of their ids (order[k] = row, sorted_ids[k] =
// its id); the block at the head of a run of equal ids adds the run's rows in that order into dE[id]. Negative ids are skipped.
template <typename T>
__global__ __launch_bounds__(256) void scatter_add_rows_sorted_kernel(const long* sorted_ids, const long* order, const T* dx, float* dE,
                                                                      long rows, int C) {
  const long k = blockIdx.x;
  const long id = sorted_ids[k];
  if (id < 0 || (k > 0 && sorted_ids[k - 1] == id)) return;
  long end = k + 1;
  while (end < rows && sorted_ids[end] == id) ++end;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = dE[id * C + c];
    for (long q = k; q < end; ++q) a += elem<T>::ld(dx + order[q] * C + c);
    dE[id * C + c] = a;
  }
}

// ---- optimiser ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void sumsq_kernel(const T* g, float* out, long n) {
  float a = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = elem<T>::ld(g + i);
    a += v * v;
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, a);
}
// AdamW (torch semantics; train_ds.py:352-360: lr 3e-4 default, betas (0.9,0.95), wd 0): fp32 master + moments,
// gradient scaled by gscale (clipping / accumulation average), optional low-precision copy of the parameter.
template <typename TG, typename TP>
__global__ void adamw_kernel(float* master, float* m, float* v, const TG* g, TP* param_lp, long n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2, float gscale, const float* gscale_dev) {
  if (gscale_dev) gscale *= *gscale_dev;   // the clip coefficient straight from the device (no host read of the gradient norm)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gr = elem<TG>::ld(g + i) * gscale;
    float w = master[i];
    w -= lr * wd * w;
    const float mm = b1 * m[i] + (1.f - b1) * gr;
    const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mm; v[i] = vv;
    w -= lr * (mm / bc1) / (sqrtf(vv / bc2) + eps);
    master[i] = w;
    if (param_lp) elem<TP>::st(param_lp + i, w);
  }
}

}  // namespace

#define HS(s) reinterpret_cast<hipStream_t>(s)
#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
  do { if ((dtype) == 0) { CALL_BF16; } else { CALL_F32; } } while (0)

extern "C" int haff_transpose(const void* in, long ld_in, long s_in_o, long s_in_i, void* out, int R, int C, int Rp, int Cp,
                              int nb_outer, int nb_inner, int dtype, void* stream) {
  if (R <= 0 || C <= 0 || Rp < R || Cp < C || nb_outer <= 0 || nb_inner <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g((Cp + 31) / 32, (Rp + 31) / 32, nb_outer * nb_inner), b(32, 8);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((transpose_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)in, ld_in, s_in_o, s_in_i, (bf16_t*)out, R, C, Rp, Cp, nb_inner),
             hipLaunchKernelGGL((transpose_kernel<float>), g, b, 0, HS(stream), (const float*)in, ld_in, s_in_o, s_in_i, (float*)out, R, C, Rp, Cp, nb_inner));
  return haff_check_launch();
}

extern "C" int haff_act_fwd(const void* x, void* y, long n, int act, int dtype, void* stream) {
  if (n <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(n, 256)), b(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((act_fwd_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)x, (bf16_t*)y, n, act),
             hipLaunchKernelGGL((act_fwd_kernel<float>), g, b, 0, HS(stream), (const float*)x, (float*)y, n, act));
  return haff_check_launch();
}
extern "C" int haff_act_bwd(const void* x, const void* dy, void* dx, long n, int act, int dtype, void* stream) {
  if (n <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(n, 256)), b(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((act_bwd_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n, act),
             hipLaunchKernelGGL((act_bwd_kernel<float>), g, b, 0, HS(stream), (const float*)x, (const float*)dy, (float*)dx, n, act));
  return haff_check_launch();
}
extern "C" int haff_swiglu_fwd(const void* gu, void* y, long M, int F, int dtype, void* stream) {
  if (M <= 0 || F <= 0 || (F & 15)) return HAFF_ERR_BAD_ARG;
  if (dtype == 0 && ((reinterpret_cast<uintptr_t>(gu) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    hipLaunchKernelGGL(swiglu_fwd_vec_kernel, dim3(grid_for(M * F / 8, 256)), dim3(256), 0, HS(stream), (const bf16_t*)gu, (bf16_t*)y, M, F);
    return haff_check_launch();
  }
  dim3 g(grid_for(M * F, 256)), b(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((swiglu_fwd_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)gu, (bf16_t*)y, M, F),
             hipLaunchKernelGGL((swiglu_fwd_kernel<float>), g, b, 0, HS(stream), (const float*)gu, (float*)y, M, F));
  return haff_check_launch();
}
extern "C" int haff_swiglu_bwd(const void* gu, const void* dy, void* dgu, long M, int F, int dtype, void* stream) {
  if (M <= 0 || F <= 0 || (F & 15)) return HAFF_ERR_BAD_ARG;
  if (dtype == 0 && ((reinterpret_cast<uintptr_t>(gu) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dgu)) & 15) == 0) {
    hipLaunchKernelGGL(swiglu_bwd_vec_kernel, dim3(grid_for(M * F / 8, 256)), dim3(256), 0, HS(stream), (const bf16_t*)gu, (const bf16_t*)dy, (bf16_t*)dgu, M, F);
    return haff_check_launch();
  }
  dim3 g(grid_for(M * F, 256)), b(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((swiglu_bwd_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)gu, (const bf16_t*)dy, (bf16_t*)dgu, M, F),
             hipLaunchKernelGGL((swiglu_bwd_kernel<float>), g, b, 0, HS(stream), (const float*)gu, (const float*)dy, (float*)dgu, M, F));
  return haff_check_launch();
}
// out = alpha*a + beta*b (b may be null)
extern "C" int haff_axpby(const void* a, const void* b, void* out, long n, float alpha, float beta, int dtype, void* stream) {
  if (n <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(n, 256)), blk(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((axpby_kernel<bf16_t>), g, blk, 0, HS(stream), (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n, alpha, beta),
             hipLaunchKernelGGL((axpby_kernel<float>), g, blk, 0, HS(stream), (const float*)a, (const float*)b, (float*)out, n, alpha, beta));
  return haff_check_launch();
}
// out = a * alpha with alpha an fp32 scalar (alpha_stride = 0) or one fp32 value per row (alpha_stride = 1) in DEVICE memory: the
// upstream gradient of a loss node applied without rounding it to the tensor's dtype and without a host read (a bf16 upstream
// scalar skews the CE gradient by up to 0.4 % against the mask gradients once a loss weight is not a power of two: ADVICE r3;
// LISA.py:417-422 weights ce / bce / dice by 1.0 / 2.0 / 0.5 by default, train_ds.py:92-94 makes them flags)
extern "C" int haff_scale_dev(const void* a, void* out, long rows, long cols, const float* alpha, long alpha_stride, int dtype,
                              void* stream) {
  if (rows <= 0 || cols <= 0 || !alpha || (alpha_stride != 0 && alpha_stride != 1)) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(rows * cols, 256)), blk(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((scale_dev_kernel<bf16_t>), g, blk, 0, HS(stream), (const bf16_t*)a, (bf16_t*)out, rows, cols, alpha, alpha_stride),
             hipLaunchKernelGGL((scale_dev_kernel<float>), g, blk, 0, HS(stream), (const float*)a, (float*)out, rows, cols, alpha, alpha_stride));
  return haff_check_launch();
}
// out = a * b elementwise (LoRA dropout mask, peft lora_dropout, train_ds.py:224)
extern "C" int haff_mul(const void* a, const void* b, void* out, long n, int dtype, void* stream) {
  if (n <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(n, 256)), blk(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((mul_kernel<bf16_t>), g, blk, 0, HS(stream), (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n),
             hipLaunchKernelGGL((mul_kernel<float>), g, blk, 0, HS(stream), (const float*)a, (const float*)b, (float*)out, n));
  return haff_check_launch();
}
// rms != 0: RMSNorm adjoint (dx only). dyx (f32 [rows][C], may be null) receives dy*xhat for the weight gradient.
static int norm_bwd_launch(const void* x, const void* dy, const float* w, const void* add, void* dx, float* dyx, int rows, int C,
                           float eps, int rms, int dtype, void* stream) {
  if (rows <= 0 || C <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g((rows + 3) / 4), b(256);
  const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                    reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dyx) | reinterpret_cast<uintptr_t>(add)) & 15) == 0;
  if (dtype == 0 && al && (C == 4096 || C == 5120)) {   // Llama 7B / 13B rows: one pass, the row in registers
#define HAFF_NBV(R_, NV_) hipLaunchKernelGGL((norm_bwd_vec_kernel<R_, NV_>), g, b, 0, HS(stream), (const bf16_t*)x, (const bf16_t*)dy, w, (bf16_t*)dx, dyx, rows, eps, (const bf16_t*)add)
    if (C == 4096) { if (rms) HAFF_NBV(true, 8); else HAFF_NBV(false, 8); }
    else { if (rms) HAFF_NBV(true, 10); else HAFF_NBV(false, 10); }
#undef HAFF_NBV
    return haff_check_launch();
  }
  if (dtype == 0) {
    if (rms) hipLaunchKernelGGL((norm_bwd_kernel<bf16_t, true>), g, b, 0, HS(stream), (const bf16_t*)x, (const bf16_t*)dy, w, (bf16_t*)dx, dyx, rows, C, eps, (const bf16_t*)add);
    else hipLaunchKernelGGL((norm_bwd_kernel<bf16_t, false>), g, b, 0, HS(stream), (const bf16_t*)x, (const bf16_t*)dy, w, (bf16_t*)dx, dyx, rows, C, eps, (const bf16_t*)add);
  } else {
    if (rms) hipLaunchKernelGGL((norm_bwd_kernel<float, true>), g, b, 0, HS(stream), (const float*)x, (const float*)dy, w, (float*)dx, dyx, rows, C, eps, (const float*)add);
    else hipLaunchKernelGGL((norm_bwd_kernel<float, false>), g, b, 0, HS(stream), (const float*)x, (const float*)dy, w, (float*)dx, dyx, rows, C, eps, (const float*)add);
  }
  return haff_check_launch();
}

extern "C" int haff_norm_bwd(const void* x, const void* dy, const float* w, void* dx, float* dyx, int rows, int C, float eps,
                             int rms, int dtype, void* stream) {
  return norm_bwd_launch(x, dy, w, nullptr, dx, dyx, rows, C, eps, rms, dtype, stream);
}

// haff_norm_bwd with the gradient of the residual branch folded in: the tensor x of a pre-norm block feeds the norm AND the
// residual add (x_new = x + f(norm(x)): transformers' LlamaDecoderLayer, image_encoder.py:186-193), so its gradient is
// dx = norm adjoint(dy) + add, add = the gradient that arrives along the residual branch (same shape / dtype as x; dx may alias
// it). One pass instead of the adjoint + autograd's separate accumulation add.
extern "C" int haff_norm_bwd_add(const void* x, const void* dy, const float* w, const void* add, void* dx, float* dyx, int rows, int C,
                                 float eps, int rms, int dtype, void* stream) {
  if (!add) return HAFF_ERR_BAD_ARG;
  return norm_bwd_launch(x, dy, w, add, dx, dyx, rows, C, eps, rms, dtype, stream);
}
extern "C" int haff_colsum(const void* x, float* out, long R, int C, int dtype, void* stream) {
  if (R <= 0 || C <= 0) return HAFF_ERR_BAD_ARG;
  if (dtype == 0 && (C & 7) == 0 && C >= 8 && C <= 2048 && ((C >> 3) & ((C >> 3) - 1)) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      R >= 1024) {
    const long rpp = 1024 / (C >> 3);
    long rpb_v = ((R + 63) / 64 + 8 * rpp - 1) / (8 * rpp) * (8 * rpp);   // <= 64 blocks, whole 8-row steps
    hipLaunchKernelGGL(colsum_vec_kernel, dim3((unsigned)((R + rpb_v - 1) / rpb_v)), dim3(1024), 0, HS(stream), (const bf16_t*)x, out, R, C, rpb_v);
    return haff_check_launch();
  }
  const long rpb = 256;
  dim3 g((C + 63) / 64, (unsigned)((R + rpb - 1) / rpb)), b(64);
  DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)x, out, R, C, rpb),
             hipLaunchKernelGGL((colsum_kernel<float>), g, b, 0, HS(stream), (const float*)x, out, R, C, rpb));
  return haff_check_launch();
}
extern "C" int haff_softmax_fwd(const float* s, long ld, void* p, long ldp, long rows, int Nq, int Nk, float scale, int causal,
                                int q_pos0, int dtype, void* stream) {
  if (rows <= 0 || Nk <= 0 || ldp < Nk || ld < Nk) return HAFF_ERR_BAD_ARG;
  dim3 g((unsigned)((rows + 3) / 4)), b(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_fwd_kernel<bf16_t>), g, b, 0, HS(stream), s, ld, (bf16_t*)p, ldp, rows, Nq, Nk, scale, causal, q_pos0),
             hipLaunchKernelGGL((softmax_fwd_kernel<float>), g, b, 0, HS(stream), s, ld, (float*)p, ldp, rows, Nq, Nk, scale, causal, q_pos0));
  return haff_check_launch();
}
extern "C" int haff_softmax_bwd(const void* p, long ldp, const float* dp, long ld, void* ds, long rows, int Nk, float scale,
                                int dtype, void* stream) {
  if (rows <= 0 || Nk <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g((unsigned)((rows + 3) / 4)), b(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)p, ldp, dp, ld, (bf16_t*)ds, rows, Nk, scale),
             hipLaunchKernelGGL((softmax_bwd_kernel<float>), g, b, 0, HS(stream), (const float*)p, ldp, dp, ld, (float*)ds, rows, Nk, scale));
  return haff_check_launch();
}
extern "C" int haff_rope(const void* x, long ldx, void* y, long ldy, const float* cos_sin, long rows, int Tlen, int H, int d,
                         int pos0, int adjoint, int dtype, void* stream) {
  if (rows <= 0 || (d & 1)) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(rows * H * (d / 2), 256)), b(256);
  const float sign = adjoint ? -1.f : 1.f;
  DISPATCH_T(dtype, hipLaunchKernelGGL((rope_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)x, ldx, (bf16_t*)y, ldy, cos_sin, rows, Tlen, H, d, pos0, sign),
             hipLaunchKernelGGL((rope_kernel<float>), g, b, 0, HS(stream), (const float*)x, ldx, (float*)y, ldy, cos_sin, rows, Tlen, H, d, pos0, sign));
  return haff_check_launch();
}
extern "C" int haff_cross_entropy(const void* logits, long ld, const long* labels, float* row_loss, void* dlogits, long rows,
                                  int V, float gscale, int dtype, void* stream) {
  if (rows <= 0 || V <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g((unsigned)rows), b(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((cross_entropy_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)logits, ld, labels, row_loss, (bf16_t*)dlogits, V, gscale),
             hipLaunchKernelGGL((cross_entropy_kernel<float>), g, b, 0, HS(stream), (const float*)logits, ld, labels, row_loss, (float*)dlogits, V, gscale));
  return haff_check_launch();
}
extern "C" int haff_mask_loss_stats(const float* x, const float* t, float* stats, int n_samples, long n, float wgt, void* stream) {
  if (n_samples <= 0 || n <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(n, 256) > 256 ? 256 : grid_for(n, 256), n_samples), b(256);
  hipLaunchKernelGGL(mask_loss_stats_kernel, g, b, 0, HS(stream), x, t, stats, n, wgt);
  return haff_check_launch();
}
extern "C" int haff_mask_loss_grad(const float* x, const float* t, const float* stats, float* dx, int n_samples, long n, float wgt,
                                   float c_bce, float c_dice, void* stream) {
  if (n_samples <= 0 || n <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(n, 256) > 256 ? 256 : grid_for(n, 256), n_samples), b(256);
  hipLaunchKernelGGL(mask_loss_grad_kernel, g, b, 0, HS(stream), x, t, stats, dx, n, wgt, c_bce, c_dice, (const float*)nullptr);
  return haff_check_launch();
}
extern "C" int haff_mask_loss_grad_dev(const float* x, const float* t, const float* stats, float* dx, int n_samples, long n, float wgt,
                                       const float* coef, void* stream) {
  if (n_samples <= 0 || n <= 0 || !coef) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(n, 256) > 256 ? 256 : grid_for(n, 256), n_samples), b(256);
  hipLaunchKernelGGL(mask_loss_grad_kernel, g, b, 0, HS(stream), x, t, stats, dx, n, wgt, 1.f, 1.f, coef);
  return haff_check_launch();
}
extern "C" int haff_resize_bilinear_bwd(const float* dout, float* din, int N, int Hs, int Ws, int Hc, int Wc, int Ho, int Wo,
                                        void* stream) {
  if (N <= 0 || Hc <= 0 || Wc <= 0 || Hc > Hs || Wc > Ws) return HAFF_ERR_BAD_ARG;
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(grid_for((long)N * Hc * Wc, 256)), dim3(256), 0, HS(stream), dout, din, N, Hs, Ws, Hc, Wc, Ho, Wo);
  return haff_check_launch();
}
extern "C" int haff_scatter_add_rows(const long* ids, const void* dx, float* dE, long rows, int C, int dtype, void* stream) {
  if (rows <= 0 || C <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(rows * C, 256)), b(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((scatter_add_rows_kernel<bf16_t>), g, b, 0, HS(stream), ids, (const bf16_t*)dx, dE, rows, C),
             hipLaunchKernelGGL((scatter_add_rows_kernel<float>), g, b, 0, HS(stream), ids, (const float*)dx, dE, rows, C));
  return haff_check_launch();
}
// ---- ordered (atomic-free, bitwise repeatable) forms; partial buffers are caller-provided DEVICE fp32 ----
extern "C" int haff_reduce_partials(const float* partials, float* out, int n_out, int n_parts, int out_inner, long part_stride,
                                    long group_stride, int accumulate, void* stream) {
  if (!partials || !out || n_out <= 0 || n_parts <= 0 || out_inner <= 0) return HAFF_ERR_BAD_ARG;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(n_out), dim3(64), 0, HS(stream), partials, out, n_out, n_parts, out_inner, part_stride,
                     group_stride, accumulate);
  return haff_check_launch();
}
// partials: >= 1024 floats; *n_parts (host) = how many were written
extern "C" int haff_sumsq_partials(const void* g, float* partials, long n, int dtype, int* n_parts, void* stream) {
  if (n <= 0 || !partials || !n_parts) return HAFF_ERR_BAD_ARG;
  const int nb = grid_for(n, 256) > 1024 ? 1024 : grid_for(n, 256);
  *n_parts = nb;
  DISPATCH_T(dtype, hipLaunchKernelGGL((sumsq_partials_kernel<bf16_t>), dim3(nb), dim3(256), 0, HS(stream), (const bf16_t*)g, partials, n),
             hipLaunchKernelGGL((sumsq_partials_kernel<float>), dim3(nb), dim3(256), 0, HS(stream), (const float*)g, partials, n));
  return haff_check_launch();
}
// partials: >= n_samples * 256 * 4 floats, laid out [sample][part][4]; *n_parts = parts per sample
extern "C" int haff_mask_loss_stats_partials(const float* x, const float* t, float* partials, int n_samples, long n, float wgt,
                                             int* n_parts, void* stream) {
  if (n_samples <= 0 || n <= 0 || !partials || !n_parts) return HAFF_ERR_BAD_ARG;
  const int nb = grid_for(n, 256) > 256 ? 256 : grid_for(n, 256);
  *n_parts = nb;
  hipLaunchKernelGGL(mask_loss_stats_partials_kernel, dim3(nb, n_samples), dim3(256), 0, HS(stream), x, t, partials, n, wgt);
  return haff_check_launch();
}
// row blocks of the column sums: haff_colsum_parts(R) blocks, partials [parts][C]
extern "C" int haff_colsum_parts(long R) {
  if (R <= 0) return 0;
  const long rpb = R <= 4096 ? 64 : (R + 255) / 256;   // <= 256 row blocks for tall inputs, 64-row blocks for short ones
  return (int)((R + rpb - 1) / rpb);
}
extern "C" int haff_colsum_partials(const void* x, float* partials, long R, int C, int dtype, void* stream) {
  if (R <= 0 || C <= 0 || !partials) return HAFF_ERR_BAD_ARG;
  const int parts = haff_colsum_parts(R);
  const long rpb = (R + parts - 1) / parts;
  dim3 g((C + 63) / 64, parts), b(64);
  DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_partials_kernel<bf16_t>), g, b, 0, HS(stream), (const bf16_t*)x, partials, R, C, rpb),
             hipLaunchKernelGGL((colsum_partials_kernel<float>), g, b, 0, HS(stream), (const float*)x, partials, R, C, rpb));
  return haff_check_launch();
}
// sorted_ids / order: a STABLE ascending sort of the rows' ids and the permutation that produced it (DEVICE int64 [rows])
extern "C" int haff_scatter_add_rows_sorted(const long* sorted_ids, const long* order, const void* dx, float* dE, long rows, int C,
                                            int dtype, void* stream) {
  if (rows <= 0 || C <= 0 || !sorted_ids || !order) return HAFF_ERR_BAD_ARG;
  DISPATCH_T(dtype, hipLaunchKernelGGL((scatter_add_rows_sorted_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, HS(stream), sorted_ids, order, (const bf16_t*)dx, dE, rows, C),
             hipLaunchKernelGGL((scatter_add_rows_sorted_kernel<float>), dim3((unsigned)rows), dim3(256), 0, HS(stream), sorted_ids, order, (const float*)dx, dE, rows, C));
  return haff_check_launch();
}
extern "C" int haff_sumsq(const void* g, float* out, long n, int dtype, void* stream) {
  if (n <= 0) return HAFF_ERR_BAD_ARG;
  dim3 gr(grid_for(n, 256) > 1024 ? 1024 : grid_for(n, 256)), b(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL((sumsq_kernel<bf16_t>), gr, b, 0, HS(stream), (const bf16_t*)g, out, n),
             hipLaunchKernelGGL((sumsq_kernel<float>), gr, b, 0, HS(stream), (const float*)g, out, n));
  return haff_check_launch();
}
// g_dtype: gradient storage (0 bf16 / 1 f32); lp_dtype: -1 none, 0 bf16 copy of the updated parameter
static int adamw_launch(float* master, float* m, float* v, const void* g, void* param_lp, long n, float lr, float beta1,
                               float beta2, float eps, float wd, int step, float gscale, const float* gscale_dev, int g_dtype, int lp_dtype, void* stream) {
  if (n <= 0 || step <= 0) return HAFF_ERR_BAD_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  dim3 gr(grid_for(n, 256)), b(256);
  if (g_dtype == 0)
    hipLaunchKernelGGL((adamw_kernel<bf16_t, bf16_t>), gr, b, 0, HS(stream), master, m, v, (const bf16_t*)g, lp_dtype == 0 ? (bf16_t*)param_lp : nullptr, n, lr, beta1, beta2, eps, wd, bc1, bc2, gscale, gscale_dev);
  else
    hipLaunchKernelGGL((adamw_kernel<float, bf16_t>), gr, b, 0, HS(stream), master, m, v, (const float*)g, lp_dtype == 0 ? (bf16_t*)param_lp : nullptr, n, lr, beta1, beta2, eps, wd, bc1, bc2, gscale, gscale_dev);
  return haff_check_launch();
}

extern "C" int haff_adamw_step(float* master, float* m, float* v, const void* g, void* param_lp, long n, float lr, float beta1,
                               float beta2, float eps, float wd, int step, float gscale, int g_dtype, int lp_dtype, void* stream) {
  return adamw_launch(master, m, v, g, param_lp, n, lr, beta1, beta2, eps, wd, step, gscale, nullptr, g_dtype, lp_dtype, stream);
}
// the same with the gradient scale multiplied by a device scalar (the clip coefficient min(1, 1 / (norm + 1e-6)) computed on the
// device: the optimizer launches are then queued behind backward without the host waiting for the norm)
extern "C" int haff_adamw_step_dev(float* master, float* m, float* v, const void* g, void* param_lp, long n, float lr, float beta1,
                                   float beta2, float eps, float wd, int step, float gscale, const float* gscale_dev, int g_dtype,
                                   int lp_dtype, void* stream) {
  if (!gscale_dev) return HAFF_ERR_BAD_ARG;
  return adamw_launch(master, m, v, g, param_lp, n, lr, beta1, beta2, eps, wd, step, gscale, gscale_dev, g_dtype, lp_dtype, stream);
}

// z, t f32 [rows][C<=8]; probs (may be null) = softmax(z); loss f32[rows]; dz (may be null) = d loss / d z
extern "C" int haff_taxonomy_ce(const float* z, const float* t, float* probs, float* loss, float* dz, int rows, int C, void* stream) {
  if (rows <= 0 || C <= 0 || C > 8) return HAFF_ERR_BAD_ARG;
  hipLaunchKernelGGL(taxonomy_ce_kernel, dim3((rows + 63) / 64), dim3(64), 0, HS(stream), z, t, probs, loss, dz, rows, C);
  return haff_check_launch();
}
