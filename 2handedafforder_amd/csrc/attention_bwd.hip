// Flash-style attention BACKWARD for the fine-tune path (row a16: transformers LlamaAttention under autograd, reached from
// 2Haff/model/LISA.py:175-430 through llava_llama.py:93-102): dQ, dK, dV of  out = softmax(scale * q k^T [+ causal]) v  without
// the probabilities ever existing in HBM. Replaces the materialised form (two batched products + a softmax kernel forward, four
// batched products + softmax_bwd + six transposes backward, P and dS [B,H,T,T] in HBM).
//
// The forward is haff_attention_lse_bf16 (attention.hip): it leaves the per-row log-sum-exp of the log2-domain scores.
// Here one workgroup (4 waves) owns one (batch, head). For each 64-key block j it keeps dK^T and dV^T of that block in
// registers (wave w: keys 16w..16w+15, all 128 head-dim rows) and walks the 64-query blocks i that see it:
//   S[q][key]  = Qs . K^T      (Qs = bf16(q * scale * log2 e): the operand the forward used, so P is reproduced bit for bit)
//   dP[q][key] = dO . V^T
//   P = 2^(S - lse[q]) (masked: 0),  dS = P * (dP - delta[q]) * scale,   delta[q] = sum_c dO[q][c] O[q][c]
//   dV^T[d][key] += dO^T[d][q] . P[q][key]         dK^T[d][key] += Q^T[d][q] . dS[q][key]
//   dQ^T[d][q]   += K^T[d][key] . dS^T[key][q]     (dS crosses LDS once, transposed, for this product only)
// The MFMA orientation puts the KEY on the lane (column) for S and dP, so their accumulator tiles are directly the B operands
// of the dV^T / dK^T products (k = the tile's 4 query rows per lane, two tiles per 32-deep k-step — the same permuted k order
// the transposed LDS reads of dO / Q deliver). dQ is summed over the key blocks of its (batch, head) by the one workgroup that
// owns them: read-modify-write of an fp32 scratch by the same lanes in program order — no atomics, bitwise repeatable.
// d == 128 only (Llama heads). The small decoder attentions of the fine-tune path keep the materialised form.
#include "haff_common.h"

namespace {

struct BwdArgs {
  const bf16_t *q, *k, *v, *dout;
  const float *lse, *delta;
  bf16_t *dq, *dk, *dv;
  float* dq_acc;    // [B*H][Nqp][128], Nqp = Nq rounded up to 64
  long ld;          // elements between consecutive tokens of q / k / v / dout / dq / dk / dv (= H * 128)
  int B, H, Nq, Nk;
  float scale;
  int causal, q_pos0;
};

constexpr int BD = 128;            // head dim
constexpr int BB = 64;             // queries / keys per block
constexpr int BSTR = BD + 8;       // LDS row stride (elements) of the [row][d] images: 272 B
constexpr int SSTR = BB + 8;       // LDS row stride of the dS^T [key][q] image: 144 B
constexpr float LOG2E_B = 1.4426950408889634f;
constexpr int BWD_LDS = (5 * BB * BSTR + BB * SSTR) * 2 + 2 * BB * 4;

typedef __attribute__((address_space(3))) bf16x4* lds4_ptr;

// A (or B) fragment of a 16x16x32 MFMA whose k index runs over the ROWS of an LDS image [k][m]: m = m0 + fr, the lane's 8 k slots
// = rows k0 + 4fh + (0..3) and k0 + 16 + 4fh + (0..3) (the order a pair of accumulator tiles delivers them in)
__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* img, int stride, int k0, int m0, int fr, int fh) {
  const bf16_t* a0 = img + (k0 + 4 * fh + (fr >> 2)) * stride + m0 + 4 * (fr & 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_ptr)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_ptr)(a0 + 16 * stride));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  bf16_t* sK = reinterpret_cast<bf16_t*>(bsm);
  bf16_t* sV = sK + BB * BSTR;
  bf16_t* sQ = sV + BB * BSTR;      // raw q (for dK)
  bf16_t* sQs = sQ + BB * BSTR;     // bf16(q * scale * log2 e) (for S, as the forward computed it)
  bf16_t* sdO = sQs + BB * BSTR;
  bf16_t* sdS = sdO + BB * BSTR;    // [key][q]
  float* sL = reinterpret_cast<float*>(sdS + BB * SSTR);
  float* sD = sL + BB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fh = lane >> 4;
  const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
  const long qbase = (long)b * p.Nq * p.ld + (long)h * BD;
  const long kbase = (long)b * p.Nk * p.ld + (long)h * BD;
  const int nqb = (p.Nq + BB - 1) / BB, nkb = (p.Nk + BB - 1) / BB;
  const int nqp = nqb * BB;
  const float sl2 = p.scale * LOG2E_B;
  float* acc_bh = p.dq_acc + (long)bh * nqp * BD;

  // one 64 x 128 bf16 tile = 1024 16-B chunks: 4 per thread
  auto stage_rows = [&](bf16_t* dst, const bf16_t* src, long base, int row0, int nrows) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, r = c >> 4, col = (c & 15) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row0 + r < nrows) v = *reinterpret_cast<const uint4*>(src + base + (long)(row0 + r) * p.ld + col);
      *reinterpret_cast<uint4*>(dst + r * BSTR + col) = v;
    }
  };

  for (int j = 0; j < nkb; ++j) {
    __syncthreads();   // the previous block's last reads of sK / sV (dQ product) are done
    stage_rows(sK, p.k, kbase, j * BB, p.Nk);
    stage_rows(sV, p.v, kbase, j * BB, p.Nk);
    f32x4 dKt[8], dVt[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) dKt[dt] = dVt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int i0 = 0;
    if (p.causal) i0 = max(0, j * BB - p.q_pos0) / BB;   // first query block with a query that sees a key of this block
    // the query block of iteration i + 1 is fetched into registers while iteration i computes (228 -> 202 us per layer at
    // 8 x 32 x 351 tokens; the rest is one wave per SIMD waiting on its own LDS reads: 3 % of the fine-tune step)
    uint4 nq[4], ndo[4];
    float nl = 0.f, nd = 0.f;
    auto fetch_block = [&](int i) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int c = tid + ii * 256, r = c >> 4, col = (c & 15) * 8;
        nq[ii] = ndo[ii] = make_uint4(0, 0, 0, 0);
        if (i < nqb && i * BB + r < p.Nq) {
          const long off = qbase + (long)(i * BB + r) * p.ld + col;
          nq[ii] = *reinterpret_cast<const uint4*>(p.q + off);
          ndo[ii] = *reinterpret_cast<const uint4*>(p.dout + off);
        }
      }
      nl = nd = 0.f;
      if (tid < BB && i < nqb && i * BB + tid < p.Nq) {
        nl = p.lse[(long)bh * p.Nq + i * BB + tid];
        nd = p.delta[(long)bh * p.Nq + i * BB + tid];
      }
    };
    fetch_block(i0);
    for (int i = i0; i < nqb; ++i) {
      __syncthreads();   // previous iteration's reads of sQ / sQs / sdO / sdS are done
      // ---- the query block (already in registers) -> LDS: q raw + pre-scaled, dO, lse, delta ----
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int c = tid + ii * 256, r = c >> 4, col = (c & 15) * 8;
        const unsigned w4[4] = {nq[ii].x, nq[ii].y, nq[ii].z, nq[ii].w};
        unsigned o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o4[e] = pack_bf16x2(__uint_as_float(w4[e] << 16) * sl2, __uint_as_float(w4[e] & 0xffff0000u) * sl2);
        *reinterpret_cast<uint4*>(sQ + r * BSTR + col) = nq[ii];
        *reinterpret_cast<uint4*>(sQs + r * BSTR + col) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        *reinterpret_cast<uint4*>(sdO + r * BSTR + col) = ndo[ii];
      }
      if (tid < BB) {
        sL[tid] = nl;
        sD[tid] = nd;
      }
      __syncthreads();
      fetch_block(i + 1);   // in flight under this iteration's products

      // ---- S = Qs . K^T and dP = dO . V^T for the wave's 16 keys x 64 queries ----
      f32x4 sS[4], dP[4];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) sS[qt] = dP[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kd = 0; kd < 4; ++kd) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (16 * wave + fr) * BSTR + kd * 32 + fh * 8);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + (16 * wave + fr) * BSTR + kd * 32 + fh * 8);
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sQs + (16 * qt + fr) * BSTR + kd * 32 + fh * 8);
          const bf16x8 of = *reinterpret_cast<const bf16x8*>(sdO + (16 * qt + fr) * BSTR + kd * 32 + fh * 8);
          sS[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, sS[qt], 0, 0, 0);
          dP[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of, vf, dP[qt], 0, 0, 0);
        }
      }
      // ---- P and dS: lane = key column 16 wave + fr, rows = queries 16 qt + 4 fh + r ----
      const int kk = j * BB + 16 * wave + fr;
      float pv[4][4], dsv[4][4];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        float L4[4], D4[4];
        load4(sL + 16 * qt + 4 * fh, L4);
        load4(sD + 16 * qt + 4 * fh, D4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = i * BB + 16 * qt + 4 * fh + r;
          bool ok = qq < p.Nq && kk < p.Nk;
          if (p.causal) ok = ok && (kk <= qq + p.q_pos0);
          const float pr = ok ? __builtin_amdgcn_exp2f(sS[qt][r] - L4[r]) : 0.f;
          pv[qt][r] = pr;
          dsv[qt][r] = pr * (dP[qt][r] - D4[r]) * p.scale;
        }
      }
      // ---- dV^T += dO^T . P,  dK^T += Q^T . dS  (k = queries: tiles 2ks, 2ks+1 per 32-deep step) ----
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 up, ud;
        up.x = pack_bf16x2(pv[2 * ks][0], pv[2 * ks][1]);         up.y = pack_bf16x2(pv[2 * ks][2], pv[2 * ks][3]);
        up.z = pack_bf16x2(pv[2 * ks + 1][0], pv[2 * ks + 1][1]); up.w = pack_bf16x2(pv[2 * ks + 1][2], pv[2 * ks + 1][3]);
        ud.x = pack_bf16x2(dsv[2 * ks][0], dsv[2 * ks][1]);         ud.y = pack_bf16x2(dsv[2 * ks][2], dsv[2 * ks][3]);
        ud.z = pack_bf16x2(dsv[2 * ks + 1][0], dsv[2 * ks + 1][1]); ud.w = pack_bf16x2(dsv[2 * ks + 1][2], dsv[2 * ks + 1][3]);
        const bf16x8 pB = __builtin_bit_cast(bf16x8, up), dB = __builtin_bit_cast(bf16x8, ud);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const bf16x8 a_do = frag_tr(sdO, BSTR, 32 * ks, 16 * dt, fr, fh);
          const bf16x8 a_q = frag_tr(sQ, BSTR, 32 * ks, 16 * dt, fr, fh);
          dVt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_do, pB, dVt[dt], 0, 0, 0);
          dKt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_q, dB, dKt[dt], 0, 0, 0);
        }
      }
      // ---- dS^T -> LDS [key][q] (the lane's 4 consecutive queries of each tile: one 8-byte store) ----
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) store4(sdS + (16 * wave + fr) * SSTR + 16 * qt + 4 * fh, dsv[qt]);
      __syncthreads();
      // ---- dQ^T[d][q] (query tile = wave) = K^T[d][key] . dS^T[key][q], summed over this key block ----
      f32x4 dQt[8];
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) dQt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 bS = frag_tr(sdS, SSTR, 32 * ks, 16 * wave, fr, fh);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const bf16x8 a_k = frag_tr(sK, BSTR, 32 * ks, 16 * dt, fr, fh);
          dQt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_k, bS, dQt[dt], 0, 0, 0);
        }
      }
      {
        const int qq = i * BB + 16 * wave + fr;   // lane: column q, rows d = 16 dt + 4 fh + r
        if (qq < p.Nq) {
          float* row = acc_bh + (long)qq * BD + 4 * fh;
#pragma unroll
          for (int dt = 0; dt < 8; ++dt) {
            float v[4] = {dQt[dt][0], dQt[dt][1], dQt[dt][2], dQt[dt][3]};
            if (j > 0) {   // key block 0 sees every query block first: it writes, later blocks add (same lane, program order)
              float old[4];
              load4(row + 16 * dt, old);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += old[r];
            }
            store4(row + 16 * dt, v);
          }
        }
      }
    }
    // ---- dK, dV of this key block: lane = key column, rows d = 16 dt + 4 fh + r ----
    {
      const int kk = j * BB + 16 * wave + fr;
      if (kk < p.Nk) {
        bf16_t* rk = p.dk + kbase + (long)kk * p.ld + 4 * fh;
        bf16_t* rv = p.dv + kbase + (long)kk * p.ld + 4 * fh;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          float a[4] = {dKt[dt][0], dKt[dt][1], dKt[dt][2], dKt[dt][3]};
          float c[4] = {dVt[dt][0], dVt[dt][1], dVt[dt][2], dVt[dt][3]};
          store4(rk + 16 * dt, a);
          store4(rv + 16 * dt, c);
        }
      }
    }
  }
  // ---- dQ: fp32 sums -> bf16 (written by other lanes of this workgroup: fence + barrier) ----
  __threadfence();
  __syncthreads();
  for (int c = tid; c < p.Nq * (BD / 8); c += 256) {
    const int qq = c >> 4, col = (c & 15) * 8;
    float v[8];
    load8(acc_bh + (long)qq * BD + col, v);
    store8(p.dq + qbase + (long)qq * p.ld + col, v);
  }
}

// delta[b][h][q] = sum_c dO[b][q][h][c] * O[b][q][h][c]: one 16-lane group per (b, q, h) row
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const bf16_t* o, const bf16_t* dout, float* delta, long ld, int B,
                                                             int H, int Nq) {
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int c = threadIdx.x & 15;
  const long total = (long)B * Nq * H;
  float s = 0.f;
  long b = 0, q = 0, h = 0;
  if (row < total) {
    h = row % H;
    const long bq = row / H;
    q = bq % Nq;
    b = bq / Nq;
    const long off = (b * Nq + q) * ld + h * BD + c * 8;
    float a[8], d[8];
    load8(o + off, a);
    load8(dout + off, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += a[e] * d[e];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 8, 64);
  if (row < total && c == 0) delta[(b * H + h) * Nq + q] = s;
}

}  // namespace

// dq, dk, dv of out = softmax(scale * q k^T [causal: key j visible to query i iff j <= i + q_pos0]) v, bf16, d == 128.
// q / dout / o / dq: [B][Nq][ld], k / v / dk / dv: [B][Nk][ld] token-major with head h at columns h*128 (ld = H*128 for the
// fused projections); lse f32 [B][H][Nq] from haff_attention_lse_bf16; workspace f32, >= B*H*(Nq + 128*roundup(Nq, 64)) values.
extern "C" int haff_attention_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                       const float* lse, void* dq, void* dk, void* dv, float* workspace, long workspace_elems,
                                       long ld, int B, int H, int Nq, int Nk, int d, float scale, int causal, int q_pos0,
                                       void* stream) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || !q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || !workspace)
    return HAFF_ERR_BAD_ARG;
  if (d != BD) return HAFF_ERR_UNSUPPORTED;
  if ((ld & 7) || ld < (long)H * BD) return HAFF_ERR_BAD_ARG;
  const void* ptrs[8] = {q, k, v, o, dout, dq, dk, dv};
  for (const void* x : ptrs)
    if (reinterpret_cast<uintptr_t>(x) & 15) return HAFF_ERR_BAD_ARG;
  const long nqp = (long)((Nq + BB - 1) / BB) * BB;
  const long need = (long)B * H * (Nq + nqp * BD);
  if (workspace_elems < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return HAFF_ERR_BAD_ARG;
  float* delta = workspace;
  float* dq_acc = workspace + (((long)B * H * Nq + 3) / 4) * 4;
  if (dq_acc + (long)B * H * nqp * BD > workspace + workspace_elems) return HAFF_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long rows = (long)B * Nq * H;
  hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const bf16_t*>(o), reinterpret_cast<const bf16_t*>(dout), delta, ld, B, H, Nq);
  BwdArgs p{reinterpret_cast<const bf16_t*>(q), reinterpret_cast<const bf16_t*>(k), reinterpret_cast<const bf16_t*>(v),
            reinterpret_cast<const bf16_t*>(dout), lse, delta, reinterpret_cast<bf16_t*>(dq), reinterpret_cast<bf16_t*>(dk),
            reinterpret_cast<bf16_t*>(dv), dq_acc, ld, B, H, Nq, Nk, scale, causal, q_pos0};
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS) !=
      hipSuccess)
    return HAFF_ERR_LAUNCH;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * H), dim3(256), BWD_LDS, s, p);
  return haff_check_launch();
}
