// fp32 attention — the PARITY-MODE twin of haff_attention_bf16 (same arguments, same semantics, all fp32).
// Plain two-pass softmax, 4 queries per 256-thread workgroup, scores held in LDS; not a performance path.
//   scores = scale * (q . k) + relh[q][k / S] + relw[q][k % S]   [+ causal mask]  ;  out = softmax(scores) @ v
// Reference semantics: image_encoder.py:235-260,354-392 (SAM), transformer.py:185-242 (SAM decoder),
// transformers CLIPAttention / LlamaAttention (third-party, reached from clip_encoder.py:53-56, llava_llama.py:93-102).
#include "haff_common.h"

namespace {

constexpr int QPB = 4;

struct AttnF32Args {
  const float *q, *k, *v;
  float* o;
  long q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st;
  int B, H, Nq, Nk, d;
  float scale;
  int causal, q_pos0;
  const float *relh, *relw;
  int S;
  const int* nk_rows;   // optional device int32 [B]: batch b sees only its first nk_rows[b] keys
};

__global__ __launch_bounds__(256) void attn_f32_kernel(AttnF32Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sS = reinterpret_cast<float*>(smem_raw);           // [QPB][Nk]
  float* sQ = sS + (long)QPB * p.Nk;                         // [QPB][d]
  float* sRed = sQ + QPB * p.d;                              // [256][4] reduction scratch (reused)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QPB;
  const int Nk = p.nk_rows ? min(p.nk_rows[b], p.Nk) : p.Nk;
  const float* qb = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
  const float* kb = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
  const float* vb = p.v + (long)b * p.v_sb + (long)h * p.v_sh;
  const long bh = (long)b * p.H + h;

  for (int i = tid; i < QPB * p.d; i += 256) {
    const int qi = min(q0 + i / p.d, p.Nq - 1);
    sQ[i] = qb[(long)qi * p.q_st + (i % p.d)];
  }
  __syncthreads();

  // pass 1: scores
  for (int j = tid; j < Nk; j += 256) {
    float acc[QPB] = {0.f, 0.f, 0.f, 0.f};
    const float* kr = kb + (long)j * p.k_st;
    for (int c = 0; c < p.d; c += 4) {
      const float4 kv = *reinterpret_cast<const float4*>(kr + c);
#pragma unroll
      for (int qi = 0; qi < QPB; ++qi) {
        const float* qq = sQ + qi * p.d + c;
        acc[qi] = fmaf(qq[0], kv.x, acc[qi]);
        acc[qi] = fmaf(qq[1], kv.y, acc[qi]);
        acc[qi] = fmaf(qq[2], kv.z, acc[qi]);
        acc[qi] = fmaf(qq[3], kv.w, acc[qi]);
      }
    }
#pragma unroll
    for (int qi = 0; qi < QPB; ++qi) {
      const int qrow = min(q0 + qi, p.Nq - 1);
      float s = acc[qi] * p.scale;
      if (p.relh) {
        const int kh = j / p.S, kw = j - kh * p.S;
        s += p.relh[(bh * p.Nq + qrow) * p.S + kh] + p.relw[(bh * p.Nq + qrow) * p.S + kw];
      }
      if (p.causal && j > qrow + p.q_pos0) s = -INFINITY;
      sS[(long)qi * Nk + j] = s;
    }
  }
  __syncthreads();

  // softmax: wave w owns query w
  {
    float* row = sS + (long)wave * Nk;
    float m = -INFINITY;
    for (int j = lane; j < Nk; j += 64) m = fmaxf(m, row[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < Nk; j += 64) {
      const float e = expf(row[j] - m);
      row[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < Nk; j += 64) row[j] *= inv;
  }
  __syncthreads();

  // pass 2: out = P @ V ; thread -> (float4 column chunk c4, key group g)
  const int nc4 = p.d / 4;
  const int G = 256 / nc4;
  const int c4 = tid % nc4, g = tid / nc4;
  float4 acc[QPB];
#pragma unroll
  for (int qi = 0; qi < QPB; ++qi) acc[qi] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < G) {
    for (int j = g; j < Nk; j += G) {
      const float4 vv = *reinterpret_cast<const float4*>(vb + (long)j * p.v_st + c4 * 4);
#pragma unroll
      for (int qi = 0; qi < QPB; ++qi) {
        const float pj = sS[(long)qi * Nk + j];
        acc[qi].x = fmaf(pj, vv.x, acc[qi].x);
        acc[qi].y = fmaf(pj, vv.y, acc[qi].y);
        acc[qi].z = fmaf(pj, vv.z, acc[qi].z);
        acc[qi].w = fmaf(pj, vv.w, acc[qi].w);
      }
    }
  }
  float* ob = p.o + (long)b * p.o_sb + (long)h * p.o_sh;
  for (int qi = 0; qi < QPB; ++qi) {
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(sRed);
    red[tid] = (g < G) ? acc[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tid < nc4 && q0 + qi < p.Nq) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int gg = 0; gg < G; ++gg) {
        const float4 r = red[gg * nc4 + tid];
        t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
      }
      *reinterpret_cast<float4*>(ob + (long)(q0 + qi) * p.o_st + tid * 4) = t;
    }
  }
}

// Few keys (Nk <= 16, no bias, no mask), many queries: the image -> token cross attention of the SAM two-way decoder
// (transformer.py:232-240: 4096 queries x 6 tokens, d = 16). One THREAD per (batch, head, query): its query row lives in
// registers, the head's K and V rows (Nk x d floats each) in LDS; scores, softmax and the weighted sum are straight-line
// per thread. HBM-bound: q read once, out written once (the generic kernel above launched one 256-thread workgroup per 4
// queries for 6 keys: 1 ms per call at 64 prompts against ~0.05 ms of memory time).
template <int D>
__global__ __launch_bounds__(256) void attn_f32_fewkeys_kernel(AttnF32Args p) {
  constexpr int MAXK = 16;
  __shared__ float sK[MAXK * D], sV[MAXK * D];
  const int b = blockIdx.z, h = blockIdx.y;
  const int Nk = p.nk_rows ? min(p.nk_rows[b], p.Nk) : p.Nk;
  const float* kb = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
  const float* vb = p.v + (long)b * p.v_sb + (long)h * p.v_sh;
  for (int i = threadIdx.x; i < Nk * D; i += 256) {
    const int j = i / D, c = i - j * D;
    sK[i] = kb[(long)j * p.k_st + c];
    sV[i] = vb[(long)j * p.v_st + c];
  }
  __syncthreads();
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (qi >= p.Nq) return;
  const float* qp = p.q + (long)b * p.q_sb + (long)h * p.q_sh + (long)qi * p.q_st;
  float q[D];
#pragma unroll
  for (int c = 0; c < D; c += 4) {
    const float4 t = *reinterpret_cast<const float4*>(qp + c);
    q[c] = t.x; q[c + 1] = t.y; q[c + 2] = t.z; q[c + 3] = t.w;
  }
  float sc[MAXK];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    float a = 0.f;
    if (j < Nk) {
#pragma unroll
      for (int c = 0; c < D; ++c) a = fmaf(q[c], sK[j * D + c], a);   // k-ordered, as the generic kernel
      a *= p.scale;
      m = fmaxf(m, a);
    }
    sc[j] = a;
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    sc[j] = j < Nk ? expf(sc[j] - m) : 0.f;
    sum += sc[j];
  }
  const float inv = 1.0f / sum;
  float o[D];
#pragma unroll
  for (int c = 0; c < D; ++c) o[c] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    if (j < Nk) {
      const float pj = sc[j] * inv;
#pragma unroll
      for (int c = 0; c < D; ++c) o[c] = fmaf(pj, sV[j * D + c], o[c]);
    }
  }
  float* op = p.o + (long)b * p.o_sb + (long)h * p.o_sh + (long)qi * p.o_st;
#pragma unroll
  for (int c = 0; c < D; c += 4) *reinterpret_cast<float4*>(op + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
}

}  // namespace

static int attention_f32_impl(const float* q, long q_sb, long q_sh, long q_st,
                              const float* k, long k_sb, long k_sh, long k_st,
                              const float* v, long v_sb, long v_sh, long v_st,
                              float* o, long o_sb, long o_sh, long o_st,
                              int B, int H, int Nq, int Nk, int d, float scale, int causal, int q_pos0,
                              const float* relh, const float* relw, int S, const int* nk_rows, void* stream) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || d <= 0 || d > 256 || (d & 3)) return HAFF_ERR_BAD_ARG;
  if ((q_st & 3) || (k_st & 3) || (v_st & 3) || (o_st & 3) || (q_sh & 3) || (k_sh & 3) || (v_sh & 3) || (o_sh & 3) ||
      (q_sb & 3) || (k_sb & 3) || (v_sb & 3) || (o_sb & 3))
    return HAFF_ERR_BAD_ARG;
  const bool rel = relh != nullptr && relw != nullptr;
  if (rel && (S <= 0 || (Nk % S) != 0)) return HAFF_ERR_BAD_ARG;
  if (!rel && !causal && Nk <= 16 && Nq >= 256 && (d == 16 || d == 32)) {
    AttnF32Args pf{q, k, v, o, q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st,
                   B, H, Nq, Nk, d, scale, 0, 0, nullptr, nullptr, 0, nk_rows};
    dim3 g((Nq + 255) / 256, H, B);
    if (d == 16) hipLaunchKernelGGL(attn_f32_fewkeys_kernel<16>, g, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pf);
    else hipLaunchKernelGGL(attn_f32_fewkeys_kernel<32>, g, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pf);
    return haff_check_launch();
  }
  size_t lds = ((size_t)QPB * Nk + (size_t)QPB * d + 256 * 4) * sizeof(float);
  if (lds > 150 * 1024) return HAFF_ERR_UNSUPPORTED;
  AttnF32Args p{q, k, v, o, q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st,
                B, H, Nq, Nk, d, scale, causal, q_pos0, rel ? relh : nullptr, rel ? relw : nullptr, S, nk_rows};
  dim3 grid((Nq + QPB - 1) / QPB, H, B), block(256);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)e;
  hipLaunchKernelGGL(attn_f32_kernel, grid, block, lds, reinterpret_cast<hipStream_t>(stream), p);
  return haff_check_launch();
}

extern "C" int haff_attention_f32(const float* q, long q_sb, long q_sh, long q_st,
                                  const float* k, long k_sb, long k_sh, long k_st,
                                  const float* v, long v_sb, long v_sh, long v_st,
                                  float* o, long o_sb, long o_sh, long o_st,
                                  int B, int H, int Nq, int Nk, int d, float scale, int causal, int q_pos0,
                                  const float* relh, const float* relw, int S, void* stream) {
  return attention_f32_impl(q, q_sb, q_sh, q_st, k, k_sb, k_sh, k_st, v, v_sb, v_sh, v_st, o, o_sb, o_sh, o_st, B, H, Nq, Nk, d,
                            scale, causal, q_pos0, relh, relw, S, nullptr, stream);
}

// fp32 twin of haff_attention_decode_rows_bf16
extern "C" int haff_attention_decode_rows_f32(const float* q, long q_sb, long q_sh,
                                              const float* k, long k_sb, long k_sh, long k_st,
                                              const float* v, long v_sb, long v_sh, long v_st,
                                              float* o, long o_sb, long o_sh,
                                              int B, int H, int Nk, int d, float scale, const int* nk_rows, void* stream) {
  if (!nk_rows) return HAFF_ERR_BAD_ARG;
  return attention_f32_impl(q, q_sb, q_sh, (long)H * d, k, k_sb, k_sh, k_st, v, v_sb, v_sh, v_st, o, o_sb, o_sh, (long)H * d, B, H, 1, Nk,
                            d, scale, 0, 0, nullptr, nullptr, 0, nk_rows, stream);
}
