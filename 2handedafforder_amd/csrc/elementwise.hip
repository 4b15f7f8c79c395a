// Small HBM-bound data-movement kernels of the 2Haff hot path (gfx950). All are pure byte/gather work:
// 16-byte vector accesses, grid sized to fill 256 CUs, no LDS needed.
//
//   haff_patchify_nchw   : conv(k=s=P) as GEMM rows — SAM PatchEmbed.proj (image_encoder.py:418-426) and the
//                          CLIP patch_embedding conv (transformers CLIPVisionEmbeddings)
//   haff_patchify_u8     : same rows straight from uint8 NHWC frames with the SAM mean/std normalisation and
//                          zero pad of inference.preprocess fused in (inference.py:91-105)
//   haff_im2col3x3       : neck 3x3 conv, pad 1, channels-last (image_encoder.py:100-106)
//   haff_embed_splice    : embed_tokens gather + image-feature splice (llava_arch.py:185-208,252-256)
//   haff_rope_cache      : rotate-half RoPE on q,k in place + KV-cache append (transformers apply_rotary_pos_emb)
//   haff_argmax_rows     : greedy next token (LISA.py:443-450 -> generate(num_beams=1))
//   haff_add_bcast       : out = a + b[row % mod]   (PE adds of transformer.py:166-178, src+dense mask_decoder.py:141)
//   haff_softmax_rows    : taxonomy softmax (mask_decoder.py:177)
#include "haff_common.h"

namespace {

template <typename TI, typename TO>
__global__ void patchify_nchw_kernel(const TI* x, TO* out, int B, int Cin, int Hin, int Win, int P, int gh, int gw,
                                     int Kp) {
  const long total = (long)B * gh * gw * Kp;
  const int K = Cin * P * P;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % Kp);
    const long row = i / Kp;
    float v = 0.f;
    if (col < K) {
      const int kx = col % P, ky = (col / P) % P, c = col / (P * P);
      const int px = (int)(row % gw), py = (int)((row / gw) % gh), b = (int)(row / ((long)gw * gh));
      v = elem<TI>::ld(x + (((long)b * Cin + c) * Hin + (py * P + ky)) * Win + (px * P + kx));
    }
    elem<TO>::st(out + i, v);
  }
}

template <typename TO>
__global__ void patchify_u8_kernel(const unsigned char* x, TO* out, int B, int Hf, int Wf, int P, int gh, int gw,
                                   int Kp, float m0, float m1, float m2, float is0, float is1, float is2) {
  const long total = (long)B * gh * gw * Kp;
  const int K = 3 * P * P;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % Kp);
    const long row = i / Kp;
    float v = 0.f;
    if (col < K) {
      const int kx = col % P, ky = (col / P) % P, c = col / (P * P);
      const int px = (int)(row % gw), py = (int)((row / gw) % gh), b = (int)(row / ((long)gw * gh));
      const int yy = py * P + ky, xx = px * P + kx;
      if (yy < Hf && xx < Wf) {
        const float raw = (float)x[(((long)b * Hf + yy) * Wf + xx) * 3 + c];
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const float istd = c == 0 ? is0 : (c == 1 ? is1 : is2);
        v = (raw - mean) / istd;
      }
    }
    elem<TO>::st(out + i, v);
  }
}

// x [B][H][W][C] channels-last -> rows [B*H*W][9*C], column = (ky*3+kx)*C + c, zero padded borders
template <typename T>
__global__ void im2col3x3_kernel(const T* x, T* out, int B, int H, int W, int C) {
  const int c8 = C / 8;
  const long total = (long)B * H * W * 9 * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int tap = (int)((i / c8) % 9);
    const long pix = i / ((long)c8 * 9);
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) load8(x + (((long)b * H + sy) * W + sx) * C + cc * 8, v);
    store8(out + pix * 9 * C + (long)tap * C + cc * 8, v);
  }
}

// ids [B][L] with one sentinel (< 0) per row at index img_pos[b]; out [B][L+n_img-1][Hd]
template <typename T>
__global__ void embed_splice_kernel(const long* ids, const int* img_pos, const T* embed, const T* img, T* out, int L,
                                    int n_img, int Hd) {
  const int T_out = L + n_img - 1;
  const int b = blockIdx.x / T_out, t = blockIdx.x % T_out;
  const int p = img_pos[b];
  const T* src;
  if (t < p) src = embed + ids[(long)b * L + t] * Hd;
  else if (t < p + n_img) src = img + ((long)b * n_img + (t - p)) * Hd;
  else src = embed + ids[(long)b * L + (t - n_img + 1)] * Hd;
  T* dst = out + ((long)b * T_out + t) * Hd;
  for (int c = threadIdx.x * 8; c < Hd; c += blockDim.x * 8) {
    float v[8];
    load8(src + c, v);
    store8(dst + c, v);
  }
}

// qkv [B*Tq][ld] (q at col 0, k at col Hq*d, v at col (Hq+Hkv)*d). Rotates q and k in place, appends k,v to
// the caches [B][Tmax][Hkv*d] at positions pos0+t. cs: fp32 [Tmax][d] = cos(0..d/2) | sin(0..d/2).
template <typename T>
__global__ void rope_cache_kernel(T* qkv, long ld, T* kcache, T* vcache, const float* cs, int B, int Tq, int Hq,
                                  int Hkv, int d, int pos0, int Tmax, const int* pos0_rows) {
  const int half = d / 2, hc = half / 8;
  const int per_row = (Hq + 2 * Hkv) * hc;
  const long total = (long)B * Tq * per_row;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % per_row);
    const long row = i / per_row;
    const int t = (int)(row % Tq), b = (int)(row / Tq);
    const int head = w / hc, ch = w % hc;
    const int pos = (pos0_rows ? pos0_rows[b] : pos0) + t;
    T* base = qkv + row * ld + (long)head * d + ch * 8;
    float x1[8], x2[8];
    load8(base, x1);
    load8(base + half, x2);
    if (head < Hq + Hkv) {
      float c[8], s[8], o1[8], o2[8];
      load8(cs + (long)pos * d + ch * 8, c);
      load8(cs + (long)pos * d + half + ch * 8, s);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o1[j] = x1[j] * c[j] - x2[j] * s[j];
        o2[j] = x2[j] * c[j] + x1[j] * s[j];
      }
      store8(base, o1);
      store8(base + half, o2);
      if (head >= Hq) {
        T* kc = kcache + ((long)b * Tmax + pos) * Hkv * d + (long)(head - Hq) * d + ch * 8;
        store8(kc, o1);
        store8(kc + half, o2);
      }
    } else {
      T* vc = vcache + ((long)b * Tmax + pos) * Hkv * d + (long)(head - Hq - Hkv) * d + ch * 8;
      store8(vc, x1);
      store8(vc + half, x2);
    }
  }
}

// First maximum of each row (torch.argmax tie rule). One 1024-thread workgroup per row, 8 independent loads per thread
// per trip: a 32 003-entry logit row is 4 trips (the 256-thread, one-load-per-trip form spent 38 us per decode step on
// 125 dependent L2 round trips).
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const float* x, long ld, long* out, int V) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const float* r = x + (long)blockIdx.x * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i0 = threadIdx.x; i0 < V; i0 += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (i0 + j * 1024 < V) ? r[i0 + j * 1024] : -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j * 1024;
      if (i < V && (v[j] > best || (v[j] == best && i < bi))) { best = v[j]; bi = i; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    out[blockIdx.x] = bi;
  }
}

// Greedy-decode bookkeeping of one generated token per row, entirely on the device so that it can sit inside the decode
// hipGraph (LISA.py:443-450's generate loop: the host only looks at `finished` between steps). Block b, step s = steps[b]:
//   token = forced ? forced[b][s] : newest argmax; a finished row emits pad; out_ids[b][lens[b] + s] = token;
//   finished[b] |= token == eos; tok[b] = token (next embedding lookup); pos[b] = t_rows[b] + s, nk[b] = pos + 1 (the
//   position the token will be decoded at); for s >= 1 the hidden state of the previous token (h1 row b, row_bytes
//   bytes) is filed at hidden[b][t_rows[b] + s - 1]; steps[b] = s + 1.
struct DecodeBookArgs {
  const long* nxt_raw; const long* forced; long forced_ld; const int* use_forced;
  int* steps; unsigned char* finished; long* out_ids; long out_ld; const long* lens; const int* t_rows;
  long* tok; int* pos; int* nk; const char* h1; char* hidden; long hid_sb, row_bytes; long pad, eos;
};
__global__ __launch_bounds__(256) void decode_book_kernel(DecodeBookArgs p) {
  const int b = blockIdx.x;
  const int s = p.steps[b];
  if (s >= 1 && p.h1) {
    const uint4* src = reinterpret_cast<const uint4*>(p.h1 + (long)b * p.row_bytes);
    uint4* dst = reinterpret_cast<uint4*>(p.hidden + (long)b * p.hid_sb + (long)(p.t_rows[b] + s - 1) * p.row_bytes);
    for (int i = threadIdx.x; i < (int)(p.row_bytes >> 4); i += 256) dst[i] = src[i];
  }
  __syncthreads();   // every thread has read steps[b] before thread 0 advances it
  if (threadIdx.x == 0) {
    long t = (*p.use_forced) ? p.forced[(long)b * p.forced_ld + s] : p.nxt_raw[b];
    const bool fin = p.finished[b] != 0;
    if (fin) t = p.pad;
    p.out_ids[(long)b * p.out_ld + p.lens[b] + s] = t;
    p.finished[b] = (fin || t == p.eos) ? 1 : 0;
    p.tok[b] = t;
    p.pos[b] = p.t_rows[b] + s;
    p.nk[b] = p.t_rows[b] + s + 1;
    p.steps[b] = s + 1;
  }
}

template <typename T>
__global__ void add_bcast_kernel(const T* a, const T* b, T* out, long rows, int C, int mod) {
  const int c8 = C / 8;
  const long total = rows * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / c8;
    const int c = (int)(i % c8) * 8;
    float x[8], y[8];
    load8(a + row * C + c, x);
    load8(b + (row % mod) * C + c, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] += y[j];
    store8(out + row * C + c, x);
  }
}

template <typename T>
__global__ void softmax_rows_kernel(const T* x, float* out, int rows, int C) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float m = -INFINITY;
  for (int c = 0; c < C; ++c) m = fmaxf(m, elem<T>::ld(x + (long)r * C + c));
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(elem<T>::ld(x + (long)r * C + c) - m);
  for (int c = 0; c < C; ++c) out[(long)r * C + c] = expf(elem<T>::ld(x + (long)r * C + c) - m) / s;
}

inline int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

#define HAFF_STREAM(s) reinterpret_cast<hipStream_t>(s)

// in_dtype/out_dtype: 0 = bf16, 1 = f32
extern "C" int haff_patchify_nchw(const void* x, void* out, int B, int Cin, int Hin, int Win, int P, int gh, int gw,
                                  int Kp, int in_dtype, int out_dtype, void* stream) {
  if (B <= 0 || P <= 0 || gh * P > Hin || gw * P > Win || Kp < Cin * P * P) return HAFF_ERR_BAD_ARG;
  const long total = (long)B * gh * gw * Kp;
  dim3 g(grid_for(total, 256)), b(256);
  hipStream_t s = HAFF_STREAM(stream);
  if (in_dtype == 0 && out_dtype == 0)
    hipLaunchKernelGGL((patchify_nchw_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)x, (bf16_t*)out, B, Cin, Hin, Win, P, gh, gw, Kp);
  else if (in_dtype == 1 && out_dtype == 0)
    hipLaunchKernelGGL((patchify_nchw_kernel<float, bf16_t>), g, b, 0, s, (const float*)x, (bf16_t*)out, B, Cin, Hin, Win, P, gh, gw, Kp);
  else if (in_dtype == 1 && out_dtype == 1)
    hipLaunchKernelGGL((patchify_nchw_kernel<float, float>), g, b, 0, s, (const float*)x, (float*)out, B, Cin, Hin, Win, P, gh, gw, Kp);
  else if (in_dtype == 0 && out_dtype == 1)
    hipLaunchKernelGGL((patchify_nchw_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)x, (float*)out, B, Cin, Hin, Win, P, gh, gw, Kp);
  else return HAFF_ERR_BAD_ARG;
  return haff_check_launch();
}

extern "C" int haff_patchify_u8(const void* frames, void* out, int B, int Hf, int Wf, int P, int gh, int gw, int Kp,
                                const float* mean3, const float* std3, int out_dtype, void* stream) {
  if (B <= 0 || P <= 0 || Kp < 3 * P * P || !mean3 || !std3) return HAFF_ERR_BAD_ARG;
  const long total = (long)B * gh * gw * Kp;
  dim3 g(grid_for(total, 256)), b(256);
  hipStream_t s = HAFF_STREAM(stream);
  if (out_dtype == 0)
    hipLaunchKernelGGL((patchify_u8_kernel<bf16_t>), g, b, 0, s, (const unsigned char*)frames, (bf16_t*)out, B, Hf, Wf, P, gh, gw, Kp,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  else
    hipLaunchKernelGGL((patchify_u8_kernel<float>), g, b, 0, s, (const unsigned char*)frames, (float*)out, B, Hf, Wf, P, gh, gw, Kp,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return haff_check_launch();
}

extern "C" int haff_im2col3x3(const void* x, void* out, int B, int H, int W, int C, int dtype, void* stream) {
  if (B <= 0 || (C & 7)) return HAFF_ERR_BAD_ARG;
  const long total = (long)B * H * W * 9 * (C / 8);
  dim3 g(grid_for(total, 256)), b(256);
  if (dtype == 0) hipLaunchKernelGGL((im2col3x3_kernel<bf16_t>), g, b, 0, HAFF_STREAM(stream), (const bf16_t*)x, (bf16_t*)out, B, H, W, C);
  else hipLaunchKernelGGL((im2col3x3_kernel<float>), g, b, 0, HAFF_STREAM(stream), (const float*)x, (float*)out, B, H, W, C);
  return haff_check_launch();
}

extern "C" int haff_embed_splice(const long* ids, const int* img_pos, const void* embed, const void* img, void* out,
                                 int B, int L, int n_img, int Hd, int dtype, void* stream) {
  if (B <= 0 || L <= 0 || n_img <= 0 || (Hd & 7)) return HAFF_ERR_BAD_ARG;
  dim3 g(B * (L + n_img - 1)), b(128);
  if (dtype == 0) hipLaunchKernelGGL((embed_splice_kernel<bf16_t>), g, b, 0, HAFF_STREAM(stream), ids, img_pos, (const bf16_t*)embed, (const bf16_t*)img, (bf16_t*)out, L, n_img, Hd);
  else hipLaunchKernelGGL((embed_splice_kernel<float>), g, b, 0, HAFF_STREAM(stream), ids, img_pos, (const float*)embed, (const float*)img, (float*)out, L, n_img, Hd);
  return haff_check_launch();
}

extern "C" int haff_rope_cache(void* qkv, long ld, void* kcache, void* vcache, const float* cos_sin, int B, int Tq,
                               int Hq, int Hkv, int d, int pos0, int Tmax, int dtype, void* stream) {
  if (B <= 0 || Tq <= 0 || (d & 15) || (ld & 7) || pos0 < 0 || pos0 + Tq > Tmax) return HAFF_ERR_BAD_ARG;
  const long total = (long)B * Tq * (Hq + 2 * Hkv) * (d / 16);
  dim3 g(grid_for(total, 256)), b(256);
  if (dtype == 0) hipLaunchKernelGGL((rope_cache_kernel<bf16_t>), g, b, 0, HAFF_STREAM(stream), (bf16_t*)qkv, ld, (bf16_t*)kcache, (bf16_t*)vcache, cos_sin, B, Tq, Hq, Hkv, d, pos0, Tmax, nullptr);
  else hipLaunchKernelGGL((rope_cache_kernel<float>), g, b, 0, HAFF_STREAM(stream), (float*)qkv, ld, (float*)kcache, (float*)vcache, cos_sin, B, Tq, Hq, Hkv, d, pos0, Tmax, nullptr);
  return haff_check_launch();
}

// Ragged batches: row b's Tq new positions start at pos0_rows[b] (device int32 [B], each with pos0_rows[b] + Tq <= Tmax —
// the caller's contract: the kernel cannot report a device-side violation) instead of one shared pos0.
extern "C" int haff_rope_cache_rows(void* qkv, long ld, void* kcache, void* vcache, const float* cos_sin, int B, int Tq,
                                    int Hq, int Hkv, int d, const int* pos0_rows, int Tmax, int dtype, void* stream) {
  if (B <= 0 || Tq <= 0 || (d & 15) || (ld & 7) || !pos0_rows || Tq > Tmax) return HAFF_ERR_BAD_ARG;
  const long total = (long)B * Tq * (Hq + 2 * Hkv) * (d / 16);
  dim3 g(grid_for(total, 256)), b(256);
  if (dtype == 0) hipLaunchKernelGGL((rope_cache_kernel<bf16_t>), g, b, 0, HAFF_STREAM(stream), (bf16_t*)qkv, ld, (bf16_t*)kcache, (bf16_t*)vcache, cos_sin, B, Tq, Hq, Hkv, d, 0, Tmax, pos0_rows);
  else hipLaunchKernelGGL((rope_cache_kernel<float>), g, b, 0, HAFF_STREAM(stream), (float*)qkv, ld, (float*)kcache, (float*)vcache, cos_sin, B, Tq, Hq, Hkv, d, 0, Tmax, pos0_rows);
  return haff_check_launch();
}

extern "C" int haff_argmax_rows(const float* x, long ld, long* out, int rows, int V, void* stream) {
  if (rows <= 0 || V <= 0) return HAFF_ERR_BAD_ARG;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(1024), 0, HAFF_STREAM(stream), x, ld, out, V);
  return haff_check_launch();
}

// One generated token of the greedy decode loop (see decode_book_kernel). All pointers are device memory: nxt_raw int64 [B]
// (newest argmax), forced int64 [B][forced_ld] used when *use_forced != 0, steps int32 [B], finished uint8 [B], out_ids int64
// [B][out_ld], lens int64 [B], t_rows int32 [B], tok int64 [B], pos / nk int32 [B], h1 [B][row_bytes] (may be null),
// hidden [B][hid_sb bytes per row] with row_bytes per position (row_bytes % 16 == 0, 16-B aligned bases).
extern "C" int haff_decode_book(const long* nxt_raw, const long* forced, long forced_ld, const int* use_forced, int* steps,
                                unsigned char* finished, long* out_ids, long out_ld, const long* lens, const int* t_rows,
                                long* tok, int* pos, int* nk, const void* h1, void* hidden, long hid_sb, long row_bytes,
                                long pad, long eos, int B, void* stream) {
  if (B <= 0 || !nxt_raw || !use_forced || !steps || !finished || !out_ids || !lens || !t_rows || !tok || !pos || !nk)
    return HAFF_ERR_BAD_ARG;
  if (h1 && ((row_bytes & 15) || (hid_sb & 15) || (reinterpret_cast<uintptr_t>(h1) & 15) || (reinterpret_cast<uintptr_t>(hidden) & 15)))
    return HAFF_ERR_BAD_ARG;
  DecodeBookArgs p{nxt_raw, forced, forced_ld, use_forced, steps, finished, out_ids, out_ld, lens, t_rows, tok, pos, nk,
                   reinterpret_cast<const char*>(h1), reinterpret_cast<char*>(hidden), hid_sb, row_bytes, pad, eos};
  hipLaunchKernelGGL(decode_book_kernel, dim3(B), dim3(256), 0, HAFF_STREAM(stream), p);
  return haff_check_launch();
}

extern "C" int haff_add_bcast(const void* a, const void* b, void* out, long rows, int C, int mod, int dtype,
                              void* stream) {
  if (rows <= 0 || (C & 7) || mod <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g(grid_for(rows * (C / 8), 256)), blk(256);
  if (dtype == 0) hipLaunchKernelGGL((add_bcast_kernel<bf16_t>), g, blk, 0, HAFF_STREAM(stream), (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, rows, C, mod);
  else hipLaunchKernelGGL((add_bcast_kernel<float>), g, blk, 0, HAFF_STREAM(stream), (const float*)a, (const float*)b, (float*)out, rows, C, mod);
  return haff_check_launch();
}

extern "C" int haff_softmax_rows(const void* x, float* out, int rows, int C, int dtype, void* stream) {
  if (rows <= 0 || C <= 0) return HAFF_ERR_BAD_ARG;
  dim3 g((rows + 63) / 64), blk(64);
  if (dtype == 0) hipLaunchKernelGGL((softmax_rows_kernel<bf16_t>), g, blk, 0, HAFF_STREAM(stream), (const bf16_t*)x, out, rows, C);
  else hipLaunchKernelGGL((softmax_rows_kernel<float>), g, blk, 0, HAFF_STREAM(stream), (const float*)x, out, rows, C);
  return haff_check_launch();
}
