// Flash-style fused attention for the 2Haff hot path on MI355X (gfx950), bf16 MFMA, fp32 softmax.
//
// One kernel family covers every attention on the reference path:
//   * SAM ViT-H windowed (14x14=196 tokens) and global (64x64=4096 tokens) attention with decomposed
//     relative-position bias            (2Haff/model/segment_anything/modeling/image_encoder.py:235-260,354-392)
//   * CLIP ViT-L/14 self-attention, S=257, d=64   (transformers CLIPAttention, reached from clip_encoder.py:53-56)
//   * Llama causal self-attention, d=128, prefill and KV-cached decode (transformers LlamaAttention,
//     reached from llava_llama.py:93-102)
//   * SAM two-way decoder attentions (d=16/32)      (segment_anything/modeling/transformer.py:185-242)
//
//   scores = scale * (q . k) + bias(q, k)   [+ causal mask]  ;  out = softmax(scores) @ v
//
// Layout/tiling (CDNA4): workgroup = 4 waves = 128 query rows (32 per wave), KV tile = 64 keys.
// The score MFMA is issued swapped (S^T = K . Q^T with v_mfma_f32_16x16x32_bf16): each lane then owns ONE
// query column and 4 keys per 16x16 tile, so the online softmax is lane-local except for a 2-step
// cross-lane max, the exponentiated tile is already the B operand of the P.V MFMA (k-order permuted
// consistently on both operands) and V^T fragments come straight from the row-major LDS image through
// ds_read_b64_tr_b16 (no transposed copy of V anywhere). K/V tiles are register-staged (issue the next
// tile's global loads before the MFMAs, write them to LDS after the barrier) with padded rows
// (K: 2D+16 B, V: 2D+32 B) so ds_read_b128 / tr reads are bank-conflict free.
// Rel-pos bias comes from per-query tables relh[q][kh], relw[q][kw] (haff_relpos_tables): BIAS=2 keeps
// relw in registers when a KV tile is exactly one key-grid row (global attention, S=64); BIAS=1 looks
// both terms up in an LDS copy (windows, S<=32).
#include "haff_common.h"

#include <type_traits>

namespace {

struct AttnArgs {
  const bf16_t *q, *k, *v;
  bf16_t* o;
  long q_sb, q_sh, q_st;
  long k_sb, k_sh, k_st;
  long v_sb, v_sh, v_st;
  long o_sb, o_sh, o_st;
  int B, H, Nq, Nk, d;
  float scale;
  int q_pos0;  // causal: key j visible to query i iff j <= i + q_pos0
  const float *relh, *relw;  // [B*H][Nq][S]
  int S;
  const int* nk_rows;  // optional, device int32 [B]: batch b sees only its first nk_rows[b] (<= Nk) keys (ragged KV caches)
  // decode kernel with RoPE + cache append fused in (attn_decode_kernel<.., true>): q/k/v above are the RAW q row and the
  // K/V caches; the newest position (nk_rows[b] - 1) comes from knew/vnew (raw, same (batch, head) strides as q), is
  // rotated here and appended to the caches by this kernel
  const bf16_t *knew, *vnew;
  const float* cos_sin;   // [Tmax][d] = cos(0..d/2) | sin(0..d/2)
  // attn_global_pp_kernel<true>: the decomposed rel-pos TABLES (bf16 [2S-1][d]); rel_h / rel_w are computed in the prologue
  const bf16_t *tab_h, *tab_w;
  // attn_fwd_kernel: optional per-row log-sum-exp of the scores in the LOG2 domain (scale * log2(e) * q.k, masked), f32
  // [B][H][Nq] — what the flash backward (attention_bwd.hip) recomputes the probabilities from
  float* lse;
};

constexpr int QB = 128;   // queries per workgroup
constexpr int KT = 64;    // keys per tile
constexpr float LOG2E = 1.4426950408889634f;

typedef __attribute__((address_space(3))) bf16x4* lds_v4_ptr;


// NDT: output d-tiles actually computed (d <= 16*NDT <= DP): SAM's d = 80 lives in DP = 96 for the QK k-steps but
// needs only 5 of the 6 P.V output tiles.
// LSUM (needs d % 16 == 0 and d < DP): column d of the staged V tile is a constant 1, so the softmax denominator falls
// out of the P.V MFMA as output row d (one extra d-tile of MFMAs instead of 16 VALU adds per q-tile per KV tile, and it
// sums exactly the bf16-rounded probabilities the numerator uses).
// FULL: Nq % 128 == 0 and Nk % 64 == 0 (and no causal mask, no grid remap): every tile is whole, so the clamped /
// remapped load path, the key masks and the row guards are compiled out (the SAM global blocks: 4096 x 4096).
template <int DP, int BIAS, bool CAUSAL, int NDT = DP / 16, bool LSUM = false, bool FULL = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
  static_assert(!FULL || (!CAUSAL && BIAS != 3), "FULL tiles only");
  // K rows: the fragment read (lane = (key fr, d-chunk fh): 16 B at row fr, column 16 fh) is conflict-free exactly for row strides
  // of 32 modulo 64 bytes (tools/probes/lds_b128_pattern.hip; the DP * 2 + 16 of rounds 1-3 ran every K read at half rate)
  constexpr int KSTRIDE = (DP * 2) % 64 == 32 ? DP * 2 : DP * 2 + 32;  // bytes
  constexpr int VSTRIDE = DP * 2 + 32;
  constexpr int NCH = DP / 32;          // 16-B chunks per thread per operand per tile
  constexpr int CPR = DP / 8;           // chunks per row
  constexpr int ND = NDT;               // output d-tiles
  constexpr int NKD = DP / 32;          // k-steps over head dim
  constexpr int SMAX = 32;
  constexpr int RSTRIDE = 2 * SMAX + 1;  // odd stride: the 16 query lanes of a tile hit 16 different banks

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int STAGE_BYTES = KT * KSTRIDE + KT * VSTRIDE;  // one K tile + one V tile
  unsigned char* sKV = smem_raw;                             // two stages (double buffer: one barrier per tile)
  float* sRel = reinterpret_cast<float*>(smem_raw + 2 * STAGE_BYTES);  // [QB][2*SMAX+1] when BIAS==1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int fr = lane & 15, fh = lane >> 4;
  // XCD-aware decode of the 1-D grid (speed only): workgroups are dealt round-robin over the 8 XCDs, so ids that
  // are equal mod 8 share an L2. All query blocks of one (batch, head) are given ids equal mod 8 and adjacent in
  // that XCD's order: its K/V (N*d*4 B, 1.3 MB for the SAM global layers) is then fetched from HBM once per XCD
  // and re-read from that XCD's 4 MiB L2 by the other query blocks.
  const int nqb = (p.Nq + QB - 1) / QB;
  const int nbh = p.B * p.H;
  int bh_id, qblk;
  {
    const int id = blockIdx.x;
    if ((nbh & 7) == 0) {
      bh_id = (id & 7) + 8 * (id / (8 * nqb));
      qblk = (id >> 3) % nqb;
    } else {
      bh_id = id / nqb;
      qblk = id - bh_id * nqb;
    }
  }
  const int b = bh_id / p.H, h = bh_id - b * p.H;
  const int q0 = qblk * QB;
  const int Nk = p.nk_rows ? min(p.nk_rows[b], p.Nk) : p.Nk;   // keys visible to this batch entry

  const bf16_t* qb = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
  const bf16_t* kb = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
  const bf16_t* vb = p.v + (long)b * p.v_sb + (long)h * p.v_sh;

  const float sl2 = p.scale * LOG2E;            // scores are kept in the log2 domain
  // ---- Q fragments (B operand: lane = (query fr, d-chunk fh)) ----
  bf16x8 qf[2][NKD];
  int qrow[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = q0 + wave * 32 + qt * 16 + fr;
    qrow[qt] = qi;
    const int qc = min(qi, p.Nq - 1);
#pragma unroll
    for (int kd = 0; kd < NKD; ++kd) {
      const int col = kd * 32 + fh * 8;
      uint4 r = make_uint4(0, 0, 0, 0);
      if (col < p.d) {
        // q is pre-multiplied by scale*log2(e) (rounded to bf16 once per workgroup, like the reference's bf16
        // `q * self.scale`): scores then leave the MFMA already in the log2 domain with the bias folded into
        // the accumulator's initial value — no per-element fma/zero-fill in the tile loop.
        float qv[8];
        load8(qb + (long)qc * p.q_st + col, qv);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] *= sl2;
        r.x = pack_bf16x2(qv[0], qv[1]); r.y = pack_bf16x2(qv[2], qv[3]);
        r.z = pack_bf16x2(qv[4], qv[5]); r.w = pack_bf16x2(qv[6], qv[7]);
      }
      qf[qt][kd] = __builtin_bit_cast(bf16x8, r);
    }
  }

  const float inv_S = p.S > 0 ? 1.0f / (float)p.S : 0.f;
  // ---- bias setup ----
  float relw_r[2][4][4];
  const float* relh_row[2];
  if (BIAS == 2) {
    const long bh = (long)b * p.H + h;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qc = min(qrow[qt], p.Nq - 1);
      const float* rw = p.relw + (bh * p.Nq + qc) * p.S;
      relh_row[qt] = p.relh + (bh * p.Nq + qc) * p.S;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 f = *reinterpret_cast<const float4*>(rw + 16 * t + 4 * fh);
        relw_r[qt][t][0] = f.x * LOG2E; relw_r[qt][t][1] = f.y * LOG2E;
        relw_r[qt][t][2] = f.z * LOG2E; relw_r[qt][t][3] = f.w * LOG2E;
      }
    }
  }
  // BIAS == 3 (S <= 16, SAM windows): keys are visited as a grid with rows padded to 16 (virtual key v -> kh = v>>4,
  // kw = v&15, real key kh*S+kw, kw >= S masked). A 64-key tile is then 4 whole grid rows, so the lane's 4 relw terms
  // are tile-invariant and the 4 relh terms of a tile are picked from 16 registers: no LDS lookups, no index math.
  float relh3[2][16], relw3[2][4];
  bool kw_ok[4];
  if (BIAS == 3) {
    const long bh = (long)b * p.H + h;
#pragma unroll
    for (int r = 0; r < 4; ++r) kw_ok[r] = (4 * fh + r) < p.S;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qc = min(qrow[qt], p.Nq - 1);
      const float* rh = p.relh + (bh * p.Nq + qc) * p.S;
      const float* rw = p.relw + (bh * p.Nq + qc) * p.S;
#pragma unroll
      for (int j = 0; j < 16; ++j) relh3[qt][j] = rh[min(j, p.S - 1)] * LOG2E;
#pragma unroll
      for (int r = 0; r < 4; ++r) relw3[qt][r] = rw[min(4 * fh + r, p.S - 1)] * LOG2E;
    }
  }
  if (BIAS == 1) {
    const long bh = (long)b * p.H + h;
    for (int i = tid; i < QB * 2 * p.S; i += 256) {
      const int ql = i / (2 * p.S);
      const int j = i - ql * 2 * p.S;
      const int qc = min(q0 + ql, p.Nq - 1);
      const float val = (j < p.S) ? p.relh[(bh * p.Nq + qc) * p.S + j] : p.relw[(bh * p.Nq + qc) * p.S + (j - p.S)];
      sRel[ql * RSTRIDE + j] = val * LOG2E;
    }
  }

  float relh_next[2] = {0.f, 0.f};   // BIAS 2: natural-log domain (scaled at use)
  if (BIAS == 2) {
    relh_next[0] = relh_row[0][0];
    relh_next[1] = relh_row[1][0];
  }

  // ---- K/V staging coordinates ----
  int st_row[NCH], st_c[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = tid + i * 256;
    st_row[i] = id / CPR;
    st_c[i] = id - st_row[i] * CPR;
  }
  uint4 kreg[NCH], vreg[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) kreg[i] = vreg[i] = make_uint4(0, 0, 0, 0);
  const bf16_t* kptr[NCH];
  const bf16_t* vptr[NCH];
  bool col_ok[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    col_ok[i] = st_c[i] * 8 < p.d;
    kptr[i] = kb + (long)st_row[i] * p.k_st + st_c[i] * 8;
    vptr[i] = vb + (long)st_row[i] * p.v_st + st_c[i] * 8;
  }
  const long k_step = (long)KT * p.k_st, v_step = (long)KT * p.v_st;
  auto load_tile = [&](int kt) {
    const bool plain = FULL || ((BIAS != 3) && (kt * KT + KT <= Nk));  // wave-uniform: whole tile in range, no remap
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (col_ok[i]) {   // chunks past d are never loaded nor written: their LDS slots are set once (below)
        if (plain) {
          kreg[i] = *reinterpret_cast<const uint4*>(kptr[i] + kt * k_step);
          vreg[i] = *reinterpret_cast<const uint4*>(vptr[i] + kt * v_step);
        } else {
          int key = kt * KT + st_row[i];
          if (BIAS == 3) key = min(key >> 4, p.S - 1) * p.S + min(key & 15, p.S - 1);
          key = min(key, Nk - 1);
          kreg[i] = *reinterpret_cast<const uint4*>(kb + (long)key * p.k_st + st_c[i] * 8);
          vreg[i] = *reinterpret_cast<const uint4*>(vb + (long)key * p.v_st + st_c[i] * 8);
        }
      }
    }
  };
  auto write_tile = [&](int buf) {
    unsigned char* wK = sKV + buf * STAGE_BYTES;
    unsigned char* wV = wK + KT * KSTRIDE;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (col_ok[i]) {
        *reinterpret_cast<uint4*>(wK + st_row[i] * KSTRIDE + st_c[i] * 16) = kreg[i];
        *reinterpret_cast<uint4*>(wV + st_row[i] * VSTRIDE + st_c[i] * 16) = vreg[i];
      }
    }
  };
  // head-dim padding (d < DP), both stages, once: K columns zero (Q is zero there too), V columns zero except the
  // ones-column at d when the row sums are taken from the MFMA
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (!col_ok[i]) {
      uint4 vpad = make_uint4(0, 0, 0, 0);
      if (LSUM && st_c[i] * 8 == p.d) vpad.x = 0x3f80u;   // bf16 1.0 in column d
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        unsigned char* wK = sKV + b2 * STAGE_BYTES;
        unsigned char* wV = wK + KT * KSTRIDE;
        *reinterpret_cast<uint4*>(wK + st_row[i] * KSTRIDE + st_c[i] * 16) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(wV + st_row[i] * VSTRIDE + st_c[i] * 16) = vpad;
      }
    }
  }

  f32x4 oacc[ND][2];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt) {
    oacc[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    oacc[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float m_run[2] = {0.f, 0.f};   // reference max the accumulators are expressed against (set by the first tile)
  float l_run[2] = {0.f, 0.f};

  int nkt = (Nk + KT - 1) / KT;
  if (BIAS == 3) nkt = (16 * p.S + KT - 1) / KT;
  if (CAUSAL) {
    const int last_q = min(q0 + QB - 1, p.Nq - 1);
    const int last_key = min(last_q + p.q_pos0, Nk - 1);
    nkt = min(nkt, last_key / KT + 1);
  }

  load_tile(0);
  write_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const unsigned char* sK = sKV + (kt & 1) * STAGE_BYTES;
    const unsigned char* sV = sK + KT * KSTRIDE;
    // BIAS 2: this tile's rel_h scalars were loaded one iteration ago. Materialise them BEFORE the next tile's K/V loads
    // are issued: hipcc otherwise places their wait at the first use, behind those loads, as a full vmcnt(0) — every
    // iteration then stalled for the whole K/V round trip before its first MFMA (51 % of the wave cycles parked).
    float rh_cur[2] = {relh_next[0], relh_next[1]};
    if (BIAS == 2) asm volatile("" : "+v"(rh_cur[0]), "+v"(rh_cur[1]));
    if (kt + 1 < nkt) load_tile(kt + 1);  // HBM -> registers, in flight during the MFMAs below

    // ---- S^T = K . Q^T, accumulators start at the (log2-domain) bias ----
    f32x4 sacc[4][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float rh3[4] = {0.f, 0.f, 0.f, 0.f};
      if (BIAS == 3) {
        // (select chain, not a switch on kt: hipcc 7.2 miscompiled the switch's default arm in this loop — wrong
        // rel_h terms for the fourth KV tile; tests/test_ops_gpu.py::test_attention case S=14 catches it)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          rh3[j] = kt == 0 ? relh3[qt][j] : (kt == 1 ? relh3[qt][4 + j] : (kt == 2 ? relh3[qt][8 + j] : relh3[qt][12 + j]));
      }
      // The running max is folded into the accumulator's initial value too: the MFMA then delivers s - m_run and
      // the exponentials need no subtraction unless this tile raises the max (first tile: no max yet, plain s).
      const float msub = m_run[qt];
      const float rhv = rh_cur[qt] * LOG2E - msub;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (BIAS == 2) sacc[t][qt] = f32x4{relw_r[qt][t][0] + rhv, relw_r[qt][t][1] + rhv, relw_r[qt][t][2] + rhv, relw_r[qt][t][3] + rhv};
        else if (BIAS == 3) {
          const float rb = rh3[t] - msub;
          sacc[t][qt] = f32x4{relw3[qt][0] + rb, relw3[qt][1] + rb, relw3[qt][2] + rb, relw3[qt][3] + rb};
        }
        else sacc[t][qt] = f32x4{-msub, -msub, -msub, -msub};
      }
      if (BIAS == 2 && kt + 1 < nkt) relh_next[qt] = relh_row[qt][kt + 1];  // one tile ahead: latency hidden
    }
#pragma unroll
    for (int kd = 0; kd < NKD; ++kd) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (16 * t + fr) * KSTRIDE + (kd * 4 + fh) * 16);
        sacc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][kd], sacc[t][0], 0, 0, 0);
        sacc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][kd], sacc[t][1], 0, 0, 0);
      }
    }

    // ---- scale, bias, mask, online softmax in the log2 domain (lane owns query column fr of each q-tile) ----
    // wave-uniform: does any element of this tile need masking?
    bool need_mask = !FULL && (kt * KT + KT > Nk);
    if (CAUSAL) need_mask = need_mask || (kt * KT + KT - 1 > q0 + wave * 32 + p.q_pos0);
    int off_h[4][4], off_w[4][4];
    if (BIAS == 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kc = min(kt * KT + 16 * t + 4 * fh + r, Nk - 1);
          const int kh = (int)(((float)kc + 0.5f) * inv_S);  // exact for kc < 2^20
          off_h[t][r] = kh;
          off_w[t][r] = p.S + kc - kh * p.S;
        }
    }
    bf16x8 pf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const float* rel_q = sRel + (wave * 32 + qt * 16 + fr) * RSTRIDE;
      float mx = -1e30f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (BIAS == 1) sacc[t][qt][r] += rel_q[off_h[t][r]] + rel_q[off_w[t][r]];
        }
      }
      if (BIAS == 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool row_ok = (4 * kt + t) < p.S;
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[t][qt][r] = (row_ok && kw_ok[r]) ? sacc[t][qt][r] : -INFINITY;
        }
      } else if (need_mask) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * KT + 16 * t + 4 * fh + r;
            bool ok = key < Nk;
            if (CAUSAL) ok = ok && (key <= qrow[qt] + p.q_pos0);
            sacc[t][qt][r] = ok ? sacc[t][qt][r] : -INFINITY;
          }
      }
      // (this file is compiled with -fno-honor-nans: fmaxf() on MFMA results otherwise gets a canonicalising
      // v_max_f32 x, x in front of every maximum — 32 extra VALU instructions per KV tile; the chain below then folds to
      // v_max3_f32. An inline-asm v_max3 is not an option: hipcc pads no MFMA -> VALU wait states around asm operands.)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        mx = fmaxf(fmaxf(mx, sacc[t][qt][0]), sacc[t][qt][1]);
        mx = fmaxf(fmaxf(mx, sacc[t][qt][2]), sacc[t][qt][3]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // s holds scores relative to m_run (or absolute before the first max is known). Exact lazy rescale: only a tile
      // that raises the max pays for a subtraction and for rescaling the accumulators.
      // (first tile: m_run is 0 and the tile max is taken unconditionally; a row with no visible key in the first
      // tile keeps a finite -1e30 reference, which cannot happen for the masks this path uses: key 0 is visible to
      // every causal query with q_pos0 >= 0)
      if (kt == 0 || __any(mx > 0.f)) {
        const float delta = kt == 0 ? fmaxf(mx, -1e30f) : fmaxf(mx, 0.f);
        const float alpha = kt == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);   // first tile: accumulators are still zero
        m_run[qt] += delta;
        const float shift = delta;
        l_run[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
          oacc[dt][qt][0] *= alpha; oacc[dt][qt][1] *= alpha;
          oacc[dt][qt][2] *= alpha; oacc[dt][qt][3] *= alpha;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[t][qt][r] -= shift;
      }
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(sacc[t][qt][r]);  // raw v_exp_f32: arguments are <= 0, tiny results may flush
          sacc[t][qt][r] = e;
          if (!LSUM) psum += e;
        }
      if (!LSUM) l_run[qt] += psum;
      // P^T fragment for k-step ks: slots j<4 <- tile 2ks (keys 4fh+j), j>=4 <- tile 2ks+1
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 u;
        u.x = pack_bf16x2(sacc[2 * ks][qt][0], sacc[2 * ks][qt][1]);
        u.y = pack_bf16x2(sacc[2 * ks][qt][2], sacc[2 * ks][qt][3]);
        u.z = pack_bf16x2(sacc[2 * ks + 1][qt][0], sacc[2 * ks + 1][qt][1]);
        u.w = pack_bf16x2(sacc[2 * ks + 1][qt][2], sacc[2 * ks + 1][qt][3]);
        pf[qt][ks] = __builtin_bit_cast(bf16x8, u);
      }
    }

    // ---- O^T += V^T . P^T  (V^T fragments via transposed LDS reads, same permuted k order) ----
    const int tr_q = fr >> 2, tr_p = fr & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        const unsigned char* a0 = sV + (16 * (2 * ks) + 4 * fh + tr_q) * VSTRIDE + (16 * dt + 4 * tr_p) * 2;
        const unsigned char* a1 = a0 + 16 * VSTRIDE;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_ptr)(a0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_ptr)(a1));
        const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        oacc[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][ks], oacc[dt][0], 0, 0, 0);
        oacc[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][ks], oacc[dt][1], 0, 0, 0);
      }
    }
    // registers -> the OTHER stage (last read one iteration ago, every wave is past that barrier), then ONE barrier:
    // it publishes the new tile and fences this tile's reads before the next iteration overwrites its stage
    if (kt + 1 < nkt) write_tile((kt + 1) & 1);
    __syncthreads();
  }

  // ---- finalize: out[q][16dt + 4fh + r] = O^T / l ----
  bf16_t* ob = p.o + (long)b * p.o_sb + (long)h * p.o_sh;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l;
    if (LSUM) {
      l = __shfl(oacc[ND - 1][qt][0], fr, 64);   // O^T row d lives in tile d/16, register 0 of the fh == 0 lanes
    } else {
      l = l_run[qt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
    if (p.lse && fh == 0 && qrow[qt] < p.Nq) p.lse[((long)b * p.H + h) * p.Nq + qrow[qt]] = m_run[qt] + __builtin_amdgcn_logf(l);
    if (qrow[qt] < p.Nq) {
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        const int col = 16 * dt + 4 * fh;
        if (col < p.d) {
          float v[4] = {oacc[dt][qt][0] * inv, oacc[dt][qt][1] * inv, oacc[dt][qt][2] * inv, oacc[dt][qt][3] * inv};
          store4(ob + (long)qrow[qt] * p.o_st + col, v);
        }
      }
    }
  }
}

template <int DP, int BIAS, bool CAUSAL, int NDT = DP / 16, bool LSUM = false, bool FULL = false>
int launch_attn(const AttnArgs& p, hipStream_t s) {
  constexpr int KSTRIDE = (DP * 2) % 64 == 32 ? DP * 2 : DP * 2 + 32, VSTRIDE = DP * 2 + 32;   // (the kernel's)
  size_t lds = 2 * ((size_t)KT * KSTRIDE + (size_t)KT * VSTRIDE);
  if (BIAS == 1) lds += (size_t)QB * (2 * 32 + 1) * sizeof(float);
  dim3 grid(((p.Nq + QB - 1) / QB) * p.H * p.B), block(256);
  hipLaunchKernelGGL((attn_fwd_kernel<DP, BIAS, CAUSAL, NDT, LSUM, FULL>), grid, block, lds, s, p);
  return haff_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// SAM global attention (image_encoder.py:235-260 at 64x64 tokens, d = 80, decomposed rel-pos), 8-wave ping-pong form.
//
// Why a second kernel: in attn_fwd_kernel every wave runs  QK MFMAs -> softmax VALU -> PV MFMAs  back to back and the two
// waves a SIMD holds mostly march in step (one barrier per KV tile), so the matrix pipe idles through the softmax and the
// VALU through the MFMAs: 36 % MFMA-busy + 44 % VALU-busy, serialised (profiles/r2_pmc_attn_global.csv). Here a workgroup
// is 8 waves = two groups of 4 (group g = wave >> 2; waves w and w+4 share a SIMD), 256 queries, and the groups run ONE
// SLOT apart, two barriers per KV tile:
//
//     slot        group 0                       group 1
//     2t+1        PV(t-1) + QK(t)   [MFMA]      softmax(t-1)      [VALU]
//     2t+2        softmax(t)        [VALU]      PV(t-1) + QK(t)   [MFMA]
//
// so each SIMD always has one wave on the matrix pipe and one on the VALU. K/V tiles come in by LDS-DMA
// (global_load_lds_dwordx4, slot-linear image: K and V rows of 10 x 16 B, no padding — see PP_KSTR), four stages each; the
// request / wait schedule is written out at `tile` below.
// rel_h of the workgroup's 256 queries (256 x 64 fp32, contiguous in the table) is copied to LDS once: the tile loop then
// holds no global load except the DMA, so the counted waits are exactly the ones written here.
// The softmax denominator comes out of one extra MFMA per (k-step, q-tile) against a register fragment of ones (the same
// bf16-rounded probabilities the numerator sums; attn_fwd_kernel's LSUM did this through a ones-column staged in LDS).
//
// Measured (32 frames x 16 heads, tools/attn_variant.py, profiles/r3_attn_pp_ablate.txt): 3.41 ms against 4.18 ms for
// attn_fwd_kernel on the same box (0.81x; 806 vs 657 TFLOP/s of useful work). What bounds it now is the SIMD's single vector
// issue port, which the two co-resident waves share: the MFMA slot's 48 MFMAs take ~1150 cycles, not 768, because the
// partner's exponentials / conversions / DMA requests are issued between them (an 8-cycle v_exp_f32 in flight delays the next
// MFMA; s_setprio cannot preempt it); without the DMA requests the launch takes 2.90 ms, without the exponentials 2.87 ms.
// Ten forms were timed on the way (header of profiles/r3_attn_pp_ablate.txt): reading next slot's fragments in the VALU slot
// made THAT slot the long one (24 LDS reads against the other group's: 3.9-4.1 ms); a per-tile cross-lane maximum with an
// eager rescale cost 14 % (rescales fired on 60 % of the tiles of random data); joined rescale arms make hipcc copy all 72
// accumulator / score registers on the common path.
constexpr int PPQ = 256;                         // queries per workgroup
constexpr int PP_D = 80, PP_CPR = PP_D / 8;      // head dim, 16-B chunks per row
// Round 4: K rows UNPADDED (160 B). The 176-byte rows of round 3 (one pad chunk, borrowed from the window kernel) made every K
// fragment read — lane (key fr, d-chunk fh): 16 B at row fr, column 16 fh — a 2-way bank conflict: PMC counted 3.4 conflict cycles
// per ds_read_b128 of K and none on the V^T reads (profiles/r4_pmc_attn_global_lds.txt), and tools/probes/lds_b128_pattern.hip
// shows why: that access pattern runs at the LDS's full rate exactly when the row stride is 32 modulo 64 bytes (160, 224, 288:
// 18 clocks per read and wave with 4 waves reading) and at half of it for 144 / 176 / 192 / 208 / 240 / 272 (32 clocks).
constexpr int PP_KSTR = PP_CPR * 16, PP_VSTR = PP_CPR * 16;
static_assert(PP_KSTR % 64 == 32, "K fragment reads are conflict-free for row strides of 32 mod 64 bytes");
constexpr int PP_KBYTES = KT * PP_KSTR, PP_VBYTES = KT * PP_VSTR;
constexpr int PP_KINS = PP_KBYTES / 1024, PP_VINS = PP_VBYTES / 1024;   // DMA instructions (64 lanes x 16 B) per tile
constexpr int PP_NST = 4;
constexpr int PP_RSTR = KT + 1;                  // rel_h row stride in floats (odd: 16 query lanes -> 16 banks)
constexpr int PP_LDS = PP_NST * (PP_KBYTES + PP_VBYTES) + PPQ * PP_RSTR * 4;   // stages, rel_h
constexpr float PP_LAZY = 40.f;                  // log2 units a score may exceed the softmax reference before the reference is moved
static_assert(PP_KBYTES % 1024 == 0 && PP_VBYTES % 1024 == 0, "whole DMA instructions per tile");
static_assert(PP_LDS <= 160 * 1024, "LDS");

#if defined(HAFF_TUNING) && defined(HAFF_PP_TRACE)
// cycle stamps of waves 0 and 4 of the first 256 workgroups, KV tiles 16..19: [wg][group][tile][8]
__device__ unsigned long long haff_pp_trace_buf[256 * 2 * 4 * 8];
#define PP_STAMP(i) do { if ((tid & 255) == 0 && blockIdx.x < 256 && kt >= 16 && kt < 20) \
    haff_pp_trace_buf[((blockIdx.x * 2 + grp) * 4 + (kt - 16)) * 8 + (i)] = clock64(); } while (0)
#else
#define PP_STAMP(i) do {} while (0)
#endif

// FUSED_REL: rel_h / rel_w do not arrive as fp32 [B*H][N][64] tables (1.07 GB written by haff_relpos_tables and read back
// per 32-frame launch): the prologue computes them for the workgroup's 256 queries from the raw q rows and the two bf16
// [127][80] parameter tables (add_decomposed_rel_pos, image_encoder.py:354-392: rel_h[q][kh] = q . Rh[qh - kh + 63],
// rel_w[q][kw] = q . Rw[qw - kw + 63], UNSCALED q) with 54 MFMAs per wave — table rows on the MFMA's row side, the lane's
// query on the column side, like the score MFMA. A wave's 32 queries share qh, so rel_h lands in its LDS rows by a plain
// index flip; rel_w's row index depends on the query's own qw, so the products pass through a wave-private LDS scratch
// (in the K/V stage area, before the first K/V request) indexed [query][kw] and come back as the 16 values a lane keeps.
template <bool FUSED_REL>
__global__ __launch_bounds__(512, 1) void attn_global_pp_kernel(AttnArgs p) {
  constexpr int NKD = 3, ND = 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
  float* sRh = reinterpret_cast<float*>(smem_raw + PP_NST * (PP_KBYTES + PP_VBYTES));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int fr = lane & 15, fh = lane >> 4;
  const int nqb = p.Nq / PPQ;
  const int nbh = p.B * p.H;
  int bh_id, qblk;
  {
    const int id = blockIdx.x;   // XCD-aware decode: ids equal mod 8 share an L2; all query blocks of a (batch, head) on one XCD
    if ((nbh & 15) == 0 && (p.H & 1) == 0) {
      // ... and the two heads an XCD runs side by side (32 CUs = 2 x 16 query blocks) are NEIGHBOURS in the fused q|k|v row:
      // a head's 160-B K (V) slice of a token row straddles 128-B lines it shares with the next head, which another XCD
      // would fetch again (PMC: 1.62 GB fetched per 32-frame launch with heads x, x+8 paired, against 0.60 GB of q, k, v)
      const int x = id & 7, seq = id / (8 * nqb);
      bh_id = 16 * (seq >> 1) + 2 * x + (seq & 1);
      qblk = (id >> 3) % nqb;
    } else if ((nbh & 7) == 0) {
      bh_id = (id & 7) + 8 * (id / (8 * nqb));
      qblk = (id >> 3) % nqb;
    } else {
      bh_id = id / nqb;
      qblk = id - bh_id * nqb;
    }
  }
  const int b = bh_id / p.H, h = bh_id - b * p.H;
  const int q0 = qblk * PPQ;
  const bf16_t* qb = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
  const bf16_t* kb = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
  const long bh = (long)b * p.H + h;
  const int nkt = p.Nk / KT;

  auto fence_barrier_early = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- DMA plan: combined instruction index n = wave + 8 i (n < 10: K, else V), lane -> slot -> source byte offset ----
  constexpr int NINS = PP_KINS + PP_VINS;   // 20
  unsigned d_off[3];
  const unsigned v_minus_k = (unsigned)((p.v - p.k) * 2);   // bytes; host checked: same for every (batch, head), >= 0
  int n_kins = 0, n_vins = 0;                                // this wave's instructions per K tile / per V tile
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int n = wave + 8 * i;
    unsigned off = 0;
    if (n < PP_KINS) {
      const int g = n * 64 + lane, row = g / PP_CPR, c = g - row * PP_CPR;
      off = (unsigned)(row * p.k_st * 2) + c * 16;
      ++n_kins;
    } else if (n < NINS) {
      const int g = (n - PP_KINS) * 64 + lane, row = g / PP_CPR, c = g - row * PP_CPR;
      off = v_minus_k + (unsigned)(row * p.v_st * 2) + c * 16;
      ++n_vins;
    }
    d_off[i] = off;
  }
  const unsigned k_step = (unsigned)(KT * p.k_st * 2), v_step = (unsigned)(KT * p.v_st * 2);
  const unsigned ldsK0 = (unsigned)(uintptr_t)(lds_ptr)smem_raw;
  const unsigned ldsV0 = ldsK0 + PP_NST * PP_KBYTES;
  // this wave's share of K tile kt_k and V tile kt_v (each skipped when past the end); returns the instructions issued
  auto issue = [&](int kt_k, int kt_v) -> int {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int n = wave + 8 * i;   // wave-uniform
      if (n < PP_KINS) {
        if (kt_k < nkt) {
          const unsigned m0v = ldsK0 + (kt_k % PP_NST) * PP_KBYTES + n * 1024;
          const unsigned off = d_off[i] + kt_k * k_step;
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(kb), "s"(m0v) : "memory");
        }
      } else if (n < NINS) {
        if (kt_v < nkt) {
          const unsigned m0v = ldsV0 + (kt_v % PP_NST) * PP_VBYTES + (n - PP_KINS) * 1024;
          const unsigned off = d_off[i] + kt_v * v_step;
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(kb), "s"(m0v) : "memory");
        }
      }
    }
    return (kt_k < nkt ? n_kins : 0) + (kt_v < nkt ? n_vins : 0);
  };
  // ---- Q fragments (pre-scaled, as attn_fwd_kernel) ----
  const float sl2 = p.scale * LOG2E;
  bf16x8 qf[2][NKD];
  float relw_r[2][4][4];
  int qloc[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    qloc[qt] = wave * 32 + qt * 16 + fr;
    const int qi = q0 + qloc[qt];
#pragma unroll
    for (int kd = 0; kd < NKD; ++kd) {
      const int col = kd * 32 + fh * 8;
      uint4 r = make_uint4(0, 0, 0, 0);
      if (col < PP_D) {
        float qv[8];
        load8(qb + (long)qi * p.q_st + col, qv);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] *= sl2;
        r.x = pack_bf16x2(qv[0], qv[1]); r.y = pack_bf16x2(qv[2], qv[3]);
        r.z = pack_bf16x2(qv[4], qv[5]); r.w = pack_bf16x2(qv[6], qv[7]);
      }
      qf[qt][kd] = __builtin_bit_cast(bf16x8, r);
    }
  }
  if constexpr (FUSED_REL) {
    constexpr int SCR = KT + 4;                       // scratch row stride in floats (16-B aligned rows)
    static_assert(8 * 32 * SCR * 4 <= PP_NST * (PP_KBYTES + PP_VBYTES), "rel_w scratch fits the stage area");
    float* scr = reinterpret_cast<float*>(smem_raw) + wave * 32 * SCR;
    const int tok0 = q0 + wave * 32;                  // the wave's first query: 32 consecutive tokens of ONE grid row
    const int qh = tok0 / KT, qwb = tok0 - qh * KT;   // qwb = 0 or 32
    const int nrow = 2 * KT - 2;                      // last table row
    bf16x8 qr[2][NKD];                                // raw q fragments (the reference multiplies the UNSCALED q)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int kd = 0; kd < NKD; ++kd) {
        const int col = kd * 32 + fh * 8;
        uint4 r = make_uint4(0, 0, 0, 0);
        if (col < PP_D) r = *reinterpret_cast<const uint4*>(qb + (long)(q0 + qloc[qt]) * p.q_st + col);
        qr[qt][kd] = __builtin_bit_cast(bf16x8, r);
      }
    auto table_frag = [&](const bf16_t* tab, int row, int kd) {   // A operand: lane = (table row, d-chunk fh)
      const int col = kd * 32 + fh * 8;
      uint4 r = make_uint4(0, 0, 0, 0);
      if (col < PP_D) r = *reinterpret_cast<const uint4*>(tab + (long)min(row, nrow) * PP_D + col);
      return __builtin_bit_cast(bf16x8, r);
    };
    // rel_h: table rows qh .. qh + 63 (4 tiles); lane (query fr, fh) register r of tile j holds row qh + 16j + 4fh + r,
    // i.e. kh = 63 - 16j - 4fh - r
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kd = 0; kd < NKD; ++kd) {
        const bf16x8 tf = table_frag(p.tab_h, qh + 16 * j + fr, kd);
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, qr[0][kd], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, qr[1][kd], a1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kh = KT - 1 - 16 * j - 4 * fh - r;
        sRh[qloc[0] * PP_RSTR + kh] = a0[r] * LOG2E;
        sRh[qloc[1] * PP_RSTR + kh] = a1[r] * LOG2E;
      }
    }
    // rel_w: q-tile qt holds qw = qwb + 16 qt + fr; table rows qwb + 16 qt + 16 j + (0..15), j = 0..4; register r of tile j
    // is row qwb + 16 qt + 16 j + 4 fh + r, i.e. kw = 63 + fr - 16 j - 4 fh - r (kept when inside 0..63)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd)
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(table_frag(p.tab_w, qwb + 16 * qt + 16 * j + fr, kd), qr[qt][kd], a, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kw = KT - 1 + fr - 16 * j - 4 * fh - r;
          if (kw >= 0 && kw < KT) scr[(qt * 16 + fr) * SCR + kw] = a[r] * LOG2E;
        }
      }
    }
    // wave-private scratch: my lanes' writes are in LDS before my lanes' reads (explicit: LDS operations of one wave
    // execute in order, but the compiler may not move the reads up, and round 1's fused kernel taught not to lean on
    // un-waited same-wave traffic)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 f = *reinterpret_cast<const float4*>(scr + (qt * 16 + fr) * SCR + 16 * t + 4 * fh);
        relw_r[qt][t][0] = f.x; relw_r[qt][t][1] = f.y; relw_r[qt][t][2] = f.z; relw_r[qt][t][3] = f.w;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fence_barrier_early();   // every wave is done with the scratch: the K/V requests below overwrite it
  } else {
    // ---- rel_h of the 256 queries -> LDS, rel_w -> registers (log2 domain), from the fp32 tables ----
    const float* rh = p.relh + (bh * p.Nq + q0) * KT;   // [256][64] contiguous
#pragma unroll
    for (int i = 0; i < PPQ * KT / 4 / 512; ++i) {
      const int e = (tid + i * 512) * 4;
      const float4 f = *reinterpret_cast<const float4*>(rh + e);
      float* dst = sRh + (e >> 6) * PP_RSTR + (e & 63);
      dst[0] = f.x * LOG2E; dst[1] = f.y * LOG2E; dst[2] = f.z * LOG2E; dst[3] = f.w * LOG2E;
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const float* rw = p.relw + (bh * p.Nq + q0 + qloc[qt]) * KT;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 f = *reinterpret_cast<const float4*>(rw + 16 * t + 4 * fh);
        relw_r[qt][t][0] = f.x * LOG2E; relw_r[qt][t][1] = f.y * LOG2E;
        relw_r[qt][t][2] = f.z * LOG2E; relw_r[qt][t][3] = f.w * LOG2E;
      }
    }
  }
  issue(0, 0);
  issue(1, 1);
  issue(2, nkt);
  if (grp == 1) issue(3, 2);   // group 1 requests one tile further ahead (see the slot schedule at `tile`)

  f32x4 oacc[ND][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) oacc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // row sums: one more MFMA per (k-step, q-tile) against a register fragment of ones — the issue port is what this kernel is
  // bound by, and 32 f32 adds per tile cost it four times what 4 MFMAs do; every row of lacc[qt] holds the same sums
  const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  f32x4 lacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float m_run[2] = {0.f, 0.f};
  bf16x8 pf[2][2];

  auto fence_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // rel_h copy written, the first K / V tiles landed
  fence_barrier();

  // lane-constant LDS offsets: K fragment rows, V transposed-read rows, the lane's rel_h rows. Stage bases are compile-time
  // (the tile loop is unrolled over the 4 stages), so every LDS address below is one VGPR + an immediate.
  const int tr_q = fr >> 2, tr_p = fr & 3;
  // third k-step of the score MFMA covers columns 64..95: lanes fh >= 2 (columns 80..95, where Q is zero) re-read the real
  // chunks 8 / 9 — whatever they multiply has to be FINITE, and the bytes behind a row's 10 chunks are not always
  const unsigned k_lane01 = fr * PP_KSTR + fh * 16;
  const unsigned k_lane2 = fr * PP_KSTR + (fh < 2 ? 8 + fh : 6 + fh) * 16;
  const unsigned v_lane = (4 * fh + tr_q) * PP_VSTR + 8 * tr_p;
  const float* rh_lane0 = sRh + qloc[0] * PP_RSTR;
  const float* rh_lane1 = sRh + qloc[1] * PP_RSTR;

  // LDS -> fragment loads. Stage bases are compile-time, so each is one ds_read with an immediate offset.
  auto load_v = [&](auto stage, auto ks_tag, bf16x8 (&vf)[ND]) {   // V^T fragments of k-step ks (keys 32ks .. 32ks+31)
    constexpr int ST = decltype(stage)::value, ks = decltype(ks_tag)::value;
#if defined(HAFF_TUNING) && defined(HAFF_PP_NOVREAD)   // counter / timing experiment: the V^T fragment reads are left out (wrong results)
    if (ST >= 0) return;
#endif
    const unsigned char* v0 = smem_raw + PP_NST * PP_KBYTES + ST * PP_VBYTES + v_lane + (32 * ks) * PP_VSTR;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_ptr)(v0 + 32 * dt));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_ptr)(v0 + 32 * dt + 16 * PP_VSTR));
      vf[dt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto load_k = [&](auto stage, auto kd_tag, bf16x8 (&kf)[4]) {    // K fragments of head-dim step kd, the 4 key tiles
    constexpr int ST = decltype(stage)::value, kd = decltype(kd_tag)::value;
#if defined(HAFF_TUNING) && defined(HAFF_PP_NOKREAD)   // counter / timing experiment: the K fragment reads are left out (wrong results)
    if (ST >= 0) return;
#endif
    const unsigned char* k0 = smem_raw + ST * PP_KBYTES + (kd < 2 ? k_lane01 + kd * 64 : k_lane2);
#pragma unroll
    for (int t = 0; t < 4; ++t) kf[t] = *reinterpret_cast<const bf16x8*>(k0 + 16 * t * PP_KSTR);
  };
  auto mma_pv = [&](auto ks_tag, const bf16x8 (&vf)[ND]) {   // O^T += V^T . P^T for k-step ks
    constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      oacc[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt], pf[0][ks], oacc[dt][0], 0, 0, 0);
      oacc[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt], pf[1][ks], oacc[dt][1], 0, 0, 0);
    }
    lacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[0][ks], lacc[0], 0, 0, 0);
    lacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[1][ks], lacc[1], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  auto phase = [&]() { __builtin_amdgcn_sched_barrier(0); };

  // One KV tile = the MFMA slot (PV of the previous tile, then this tile's scores), then the VALU slot (requests, softmax).
  // In the MFMA slot only ONE wave per SIMD feeds the matrix pipe (its partner is in the VALU slot), so an LDS read waited
  // for right before its MFMA is latency nobody covers (first form of this kernel: 1430 cycles to issue 48 MFMAs,
  // profiles/r3_attn_pp_trace.txt). Every fragment is therefore read at least one phase before the MFMAs that consume it,
  // and the first 56 registers' worth already in the VALU slot BEFORE, where the wave has issue slots to spare:
  //   VALU slot kt:   requests | softmax(kt) -> P | reads V[kt] (both k-steps), K[kt+1] kd0 | score init for kt+1
  //   MFMA slot kt+1: PV ks0 | reads K kd1, PV ks1 | reads K kd2, QK kd0 | QK kd1 | QK kd2
  // Requests: tile pair X = (K[X+1], V[X]) is first read in group 0's VALU slot X, so it must have landed — every wave's
  // share — by the barrier in front of that slot. Group 0 requests its share in its VALU slot X-2 and waits for it at the end
  // of its MFMA slot X (one newer batch may stay in flight); group 1 — whose slots are the odd ones — requests in its VALU
  // slot X-3 and waits at the end of its VALU slot X-1 (two newer batches in flight). The stages those requests overwrite
  // (K[X-3], V[X-4]) were last read two or more barriers earlier by both groups.
  bf16x8 vf0[ND], vf1[ND], kf0[4];
#if defined(HAFF_TUNING) && (defined(HAFF_PP_NOVREAD) || defined(HAFF_PP_NOKREAD))
  for (int i = 0; i < ND; ++i) { vf0[i] = ones; vf1[i] = ones; }
  for (int i = 0; i < 4; ++i) kf0[i] = ones;
#endif
  f32x4 sacc[4][2];
  auto batch_cnt = [&](int x) { return (x + 1 < nkt ? n_kins : 0) + (x < nkt ? n_vins : 0); };
  auto wait_vm = [&](int n) {   // wave-uniform n: at most n of my requests still in flight
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    }
  };
  // score accumulators start at  rel_w + rel_h - m_run  (log2 domain)
  auto init_scores = [&](float rh0, float rh1) {
    const float rhv0 = rh0 - m_run[0], rhv1 = rh1 - m_run[1];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      sacc[t][0] = f32x4{relw_r[0][t][0] + rhv0, relw_r[0][t][1] + rhv0, relw_r[0][t][2] + rhv0, relw_r[0][t][3] + rhv0};
      sacc[t][1] = f32x4{relw_r[1][t][0] + rhv1, relw_r[1][t][1] + rhv1, relw_r[1][t][2] + rhv1, relw_r[1][t][3] + rhv1};
    }
  };
  auto tile = [&](int kt, auto stage, auto first_tag) {
    constexpr int ST = decltype(stage)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    using PrevStage = std::integral_constant<int, (ST + PP_NST - 1) % PP_NST>;
    PP_STAMP(0);
    // ================= MFMA slot =================
    bf16x8 kf1[4], kf2[4];
#if defined(HAFF_TUNING) && defined(HAFF_PP_NOKREAD)
    for (int i = 0; i < 4; ++i) { kf1[i] = ones; kf2[i] = ones; }
#endif
    __builtin_amdgcn_s_setprio(1);
#if !(defined(HAFF_TUNING) && defined(HAFF_PP_NOMFMA))
    auto qk = [&](auto kd_tag, const bf16x8 (&kf)[4]) {   // S^T += K . Q^T over head-dim step kd
      constexpr int kd = decltype(kd_tag)::value;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sacc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t], qf[0][kd], sacc[t][0], 0, 0, 0);
        sacc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t], qf[1][kd], sacc[t][1], 0, 0, 0);
      }
    };
    // (sched_group_barrier: the phase's reads go out BEFORE its MFMAs — left alone, hipcc reuses the registers of the
    // fragments being consumed for the ones being fetched and so sinks the reads behind the MFMAs. Alternating PV and QK
    // steps to put 20 MFMAs between every read and its use measured no better: the slot's 48 MFMAs take ~1150 cycles
    // instead of 768 because the partner wave's VALU / DMA instructions share the SIMD's issue port, not for LDS latency.)
    if (!FIRST) {
      load_v(PrevStage{}, I1{}, vf1);
      load_k(stage, I0{}, kf0);
      init_scores(rh_lane0[kt], rh_lane1[kt]);   // (adds in the shadow of the reads / the first MFMAs)
      mma_pv(I0{}, vf0);
      phase();
      load_k(stage, I1{}, kf1);
      mma_pv(I1{}, vf1);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      phase();
      load_k(stage, I2{}, kf2);
      qk(I0{}, kf0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      phase();
      qk(I1{}, kf1);
      phase();
      qk(I2{}, kf2);
    } else {
      load_k(stage, I0{}, kf0);
      load_k(stage, I1{}, kf1);
      load_k(stage, I2{}, kf2);
      init_scores(rh_lane0[kt], rh_lane1[kt]);
      qk(I0{}, kf0);
      qk(I1{}, kf1);
      qk(I2{}, kf2);
    }
#endif
    __builtin_amdgcn_s_setprio(0);
    PP_STAMP(1);
    // my LDS reads are done (the stages go back to the DMA). Group 0: batch kt has landed (batch kt+1 may be in flight).
    if (grp == 0) wait_vm(FIRST ? 0 : batch_cnt(kt + 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_STAMP(2);
    fence_barrier();
    PP_STAMP(3);
    // ================= VALU slot =================
    // (the two small LDS reads of this slot go first: LDS returns in order, a value read behind the 24 fragment reads
    // below would be waited for behind all of them)
#if !(defined(HAFF_TUNING) && defined(HAFF_PP_NODMA))
    if (grp == 0) issue(kt + 3, kt + 2);
    else issue(kt + 4, kt + 3);
#endif
    PP_STAMP(4);
    // fragments for the MFMA slot behind the next barrier — V[kt] whole, K[kt+1] kd0 — requested NOW: they land under the
    // softmax arithmetic below (issued behind it, this wave would sit in the issue stage until most of them had returned:
    // 14 KB per wave, four waves at once, against the other group's reads)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      // Exponentials -> P^T fragments (k-step ks: slots j < 4 <- key tile 2ks, j >= 4 <- key tile 2ks+1). The exponentials
      // only need a reference m_run that keeps them in RANGE, not the running maximum: the first tile sets it to that tile's
      // maximum; later p = 2^(s - m_run) may exceed 1 (bf16 and fp32 carry an 8-bit exponent: exact arithmetic is unchanged up
      // to rounding). Only when a score lands more than PP_LAZY above the reference is the reference moved: that pass takes
      // the cross-lane maximum and rescales the accumulators.
      auto exp_pack = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          float e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float sv = sacc[2 * ks + (j >> 2)][qt][j & 3];
#if defined(HAFF_TUNING) && defined(HAFF_PP_NOSOFTMAX)
            e[j] = sv;
#else
            e[j] = __builtin_amdgcn_exp2f(sv);
#endif
          }
          uint4 u;
          u.x = pack_bf16x2(e[0], e[1]); u.y = pack_bf16x2(e[2], e[3]);
          u.z = pack_bf16x2(e[4], e[5]); u.w = pack_bf16x2(e[6], e[7]);
          pf[qt][ks] = __builtin_bit_cast(bf16x8, u);
        }
      };
      auto move_reference = [&]() {
        float mx = fmaxf(fmaxf(sacc[0][qt][0], sacc[0][qt][1]), fmaxf(sacc[0][qt][2], sacc[0][qt][3]));
#pragma unroll
        for (int t = 1; t < 4; ++t) {
          mx = fmaxf(fmaxf(mx, sacc[t][qt][0]), sacc[t][qt][1]);
          mx = fmaxf(fmaxf(mx, sacc[t][qt][2]), sacc[t][qt][3]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float delta = FIRST ? mx : fmaxf(mx, 0.f);
        m_run[qt] += delta;
        if (!FIRST) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          lacc[qt][0] *= alpha; lacc[qt][1] *= alpha; lacc[qt][2] *= alpha; lacc[qt][3] *= alpha;
#pragma unroll
          for (int dt = 0; dt < ND; ++dt) {
            oacc[dt][qt][0] *= alpha; oacc[dt][qt][1] *= alpha;
            oacc[dt][qt][2] *= alpha; oacc[dt][qt][3] *= alpha;
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[t][qt][r] -= delta;
      };
#if defined(HAFF_TUNING) && defined(HAFF_PP_NOSOFTMAX)
      exp_pack();
#else
      if (FIRST) {
        move_reference();
        exp_pack();
      } else {
        // lane-local maximum of the lane's 16 scores (relative to m_run): 8 instructions, no cross-lane step
        float mx = fmaxf(fmaxf(sacc[0][qt][0], sacc[0][qt][1]), fmaxf(sacc[0][qt][2], sacc[0][qt][3]));
#pragma unroll
        for (int t = 1; t < 4; ++t) {
          mx = fmaxf(fmaxf(mx, sacc[t][qt][0]), sacc[t][qt][1]);
          mx = fmaxf(fmaxf(mx, sacc[t][qt][2]), sacc[t][qt][3]);
        }
        if (__any(mx > PP_LAZY)) move_reference();   // rare (never on SAM's logits)
        exp_pack();
      }
#endif
    }
    phase();
#if !(defined(HAFF_TUNING) && defined(HAFF_PP_NOMFMA))
    load_v(stage, I0{}, vf0);   // the first fragments the next MFMA slot consumes; nothing behind them in this slot reads LDS
#endif
    PP_STAMP(5);
    // group 1: batch kt+1 has landed (batches kt+2, kt+3 may be in flight) — group 0 reads it right behind this barrier
    if (grp == 1) wait_vm((kt >= 1 ? batch_cnt(kt + 2) : 0) + batch_cnt(kt + 3));
    fence_barrier();
    PP_STAMP(6);
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  if (grp == 1) fence_barrier();   // group 1 runs one slot behind
  tile(0, S0{}, std::true_type{});
  tile(1, S1{}, std::false_type{});           // (the host admits nkt >= 2 only)
  if (2 < nkt) tile(2, S2{}, std::false_type{});
  if (3 < nkt) tile(3, S3{}, std::false_type{});
  for (int kt = 4; kt < nkt; kt += 4) {
    tile(kt, S0{}, std::false_type{});
    if (kt + 1 < nkt) tile(kt + 1, S1{}, std::false_type{});
    if (kt + 2 < nkt) tile(kt + 2, S2{}, std::false_type{});
    if (kt + 3 < nkt) tile(kt + 3, S3{}, std::false_type{});
  }
  switch ((nkt - 1) & 3) {   // PV of the last tile (its first V fragments were read in the last VALU slot)
    case 0: load_v(S0{}, I1{}, vf1); break;
    case 1: load_v(S1{}, I1{}, vf1); break;
    case 2: load_v(S2{}, I1{}, vf1); break;
    default: load_v(S3{}, I1{}, vf1); break;
  }
  mma_pv(I0{}, vf0);
  mma_pv(I1{}, vf1);
  if (grp == 0) fence_barrier();   // the barrier group 1 took at the top

  // ---- finalize: out[q][16dt + 4fh + r] = O^T / l ----
  bf16_t* ob = p.o + (long)b * p.o_sb + (long)h * p.o_sh;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float inv = 1.0f / lacc[qt][0];
    const long qi = q0 + qloc[qt];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      float v[4] = {oacc[dt][qt][0] * inv, oacc[dt][qt][1] * inv, oacc[dt][qt][2] * inv, oacc[dt][qt][3] * inv};
      store4(ob + qi * p.o_st + 16 * dt + 4 * fh, v);
    }
  }
}

#if defined(HAFF_TUNING) && defined(HAFF_PP_TRACE)
extern "C" int haff_pp_trace_read(void* dst, int n) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(haff_pp_trace_buf), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif

// host-side admission for attn_global_pp_kernel: whole tiles, the fused q|k|v row layout (V a fixed, non-negative
// distance behind K, same strides), 32-bit source offsets; fused_rel: rel-pos from the parameter tables in the prologue
// (a wave's 32 queries must be consecutive tokens of one grid row: Nq == Nk == S*S)
static bool attn_global_pp_ok(const AttnArgs& p, bool fused_rel) {
  if (p.d != PP_D || p.S != KT || (p.Nq % PPQ) || (p.Nk % KT) || p.Nk < 2 * KT || p.nk_rows) return false;
  if (p.k_sb != p.v_sb || p.k_sh != p.v_sh || p.k_st != p.v_st || p.v < p.k) return false;
  const long span = (p.v - p.k) * 2 + (long)p.Nk * p.k_st * 2;
  if (span >= (1L << 31)) return false;
  if ((reinterpret_cast<uintptr_t>(p.k) & 15) || (reinterpret_cast<uintptr_t>(p.v) & 15) || (reinterpret_cast<uintptr_t>(p.q) & 15))
    return false;
  if (fused_rel)
    return p.Nq == KT * KT && p.Nk == KT * KT && p.tab_h && p.tab_w && (reinterpret_cast<uintptr_t>(p.tab_h) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(p.tab_w) & 15) == 0;
  return (reinterpret_cast<uintptr_t>(p.relh) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.relw) & 15) == 0;
}

template <bool FUSED_REL>
static int launch_attn_global_pp(const AttnArgs& p, hipStream_t s) {
  // the attribute is per device and this entry point keeps no state: set it on every call (a host-side table write)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_global_pp_kernel<FUSED_REL>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS) != hipSuccess)
    return HAFF_ERR_LAUNCH;
  dim3 grid((p.Nq / PPQ) * p.H * p.B), block(512);
  hipLaunchKernelGGL(attn_global_pp_kernel<FUSED_REL>, grid, block, PP_LDS, s, p);
  return haff_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// KV-cached decode step (Nq == 1, d == 128, every key visible): HBM-bound, no matrix shape to it — one wave per
// (batch, head) streams the K and V rows of that head once and does the two GEMVs on the VALU.
// Lane = (g = lane>>4, c = lane&15): per iteration the wave reads 4 consecutive keys (one per 16-lane row g), lane
// (g, c) holding the 16-B chunk c of K[key] and of V[key]: a whole 256-B row per 16 lanes, 1 KB per instruction.
// score(key) = sum over the row's 16 lanes (DPP row reduction leaves it in all 16), so the lanes that need p(key) for
// the P.V update already have it: no LDS, no barrier. Online softmax per row group (keys = g mod 4), the four groups
// are merged once at the end. (The generic 128-query flash kernel spent 190 us per layer on this at B=64.)
constexpr int DEC_D = 128;
constexpr int DEC_UNROLL = 4;   // 16-B loads in flight per lane per operand

__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, result in every lane
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// SPLIT: the four waves of a workgroup share ONE (batch, head) and take every fourth batch of 16 keys each (small
// B*H: one wave per head left the step latency-bound — 35 us per layer at batch 1 for 4.7 MB of K/V); their partial
// (max, sum, P.V) meet in LDS.
// rotate-half RoPE of the 8 values a lane holds of a 128-wide head row (lane c: dims 8c .. 8c+7; the partner half sits 8
// lanes away in the same 16-lane row), rounded to bf16 as the stand-alone rope kernel stores it
__device__ __forceinline__ void rope8(float (&x)[8], const float* cs_row, int c) {
  const int ci = (c & 7) * 8;
  float cv[8], sv[8];
  load8(cs_row + ci, cv);
  load8(cs_row + DEC_D / 2 + ci, sv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float other = __shfl_xor(x[j], 8, 64);
    const float r = c < 8 ? x[j] * cv[j] - other * sv[j] : x[j] * cv[j] + other * sv[j];
    x[j] = bf16_to_f32(f32_to_bf16(r));
  }
}

// NW: waves per workgroup. SPLIT with 16 waves (few (batch, head) pairs: batch 1..4): each wave takes ONE trip of 16 keys
// per 256 — a 300-key cache is one or two round trips per wave instead of five (10.7 -> ~6 us per layer at batch 1).
template <bool SPLIT, bool ROPE = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void attn_decode_kernel(AttnArgs p) {
  static_assert(NW == 4 || SPLIT, "the one-wave-per-head form packs 4 heads per workgroup");
  __shared__ float s_ml[NW][2];
  __shared__ float s_o[NW][DEC_D];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int bh = SPLIT ? blockIdx.x : blockIdx.x * 4 + wave;
  if (bh >= p.B * p.H) return;
  const int b = bh / p.H, h = bh - b * p.H;
  const int Nk = p.nk_rows ? min(p.nk_rows[b], p.Nk) : p.Nk;
  const int g = lane >> 4, c = lane & 15;
  const bf16_t* qb = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
  const bf16_t* kb = p.k + (long)b * p.k_sb + (long)h * p.k_sh + c * 8;
  const bf16_t* vb = p.v + (long)b * p.v_sb + (long)h * p.v_sh + c * 8;

  float qv[8];
  load8(qb + c * 8, qv);
  uint4 knew_bits = make_uint4(0, 0, 0, 0), vnew_bits = make_uint4(0, 0, 0, 0);
  if (ROPE) {
    // position of this step = Nk - 1: rotate q and the new k (transformers apply_rotary_pos_emb), append k, v to the caches
    const float* cs_row = p.cos_sin + (long)(Nk - 1) * DEC_D;
    rope8(qv, cs_row, c);
    float kn[8];
    load8(p.knew + (long)b * p.q_sb + (long)h * p.q_sh + c * 8, kn);
    rope8(kn, cs_row, c);
    knew_bits.x = pack_bf16x2(kn[0], kn[1]); knew_bits.y = pack_bf16x2(kn[2], kn[3]);
    knew_bits.z = pack_bf16x2(kn[4], kn[5]); knew_bits.w = pack_bf16x2(kn[6], kn[7]);
    vnew_bits = *reinterpret_cast<const uint4*>(p.vnew + (long)b * p.q_sb + (long)h * p.q_sh + c * 8);
    if (g == 0 && (!SPLIT || wave == 0)) {
      *reinterpret_cast<uint4*>(const_cast<bf16_t*>(kb) + (long)(Nk - 1) * p.k_st) = knew_bits;
      *reinterpret_cast<uint4*>(const_cast<bf16_t*>(vb) + (long)(Nk - 1) * p.v_st) = vnew_bits;
    }
  }
  const float sl2 = p.scale * LOG2E;
#pragma unroll
  for (int j = 0; j < 8; ++j) qv[j] *= sl2;

  float m_run = -1e30f, l_run = 0.f;
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int n_it = (Nk + 3) / 4;
  for (int it0 = SPLIT ? wave * DEC_UNROLL : 0; it0 < n_it; it0 += (SPLIT ? NW : 1) * DEC_UNROLL) {
    uint4 kr[DEC_UNROLL], vr[DEC_UNROLL];
#pragma unroll
    for (int u = 0; u < DEC_UNROLL; ++u) {
      const int key = min((it0 + u) * 4 + g, Nk - 1);
      if (ROPE) {   // the newest row is not in the cache yet for the other waves: take it from registers
        const int kc = min(key, max(Nk - 2, 0));
        kr[u] = *reinterpret_cast<const uint4*>(kb + (long)kc * p.k_st);
        vr[u] = *reinterpret_cast<const uint4*>(vb + (long)kc * p.v_st);
        if (key == Nk - 1) { kr[u] = knew_bits; vr[u] = vnew_bits; }
      } else {
        kr[u] = *reinterpret_cast<const uint4*>(kb + (long)key * p.k_st);
        vr[u] = *reinterpret_cast<const uint4*>(vb + (long)key * p.v_st);
      }
    }
#pragma unroll
    for (int u = 0; u < DEC_UNROLL; ++u) {
      const int key = (it0 + u) * 4 + g;
      const unsigned kw[4] = {kr[u].x, kr[u].y, kr[u].z, kr[u].w};
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s += qv[2 * j] * __builtin_bit_cast(float, kw[j] << 16);
        s += qv[2 * j + 1] * __builtin_bit_cast(float, kw[j] & 0xffff0000u);
      }
      s = row16_sum(s);
      s = key < Nk ? s : -INFINITY;
      const float m_new = fmaxf(m_run, s);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      const float pr = __builtin_amdgcn_exp2f(s - m_new);
      m_run = m_new;
      l_run = l_run * alpha + pr;
      const unsigned vw[4] = {vr[u].x, vr[u].y, vr[u].z, vr[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[2 * j] = o[2 * j] * alpha + pr * __builtin_bit_cast(float, vw[j] << 16);
        o[2 * j + 1] = o[2 * j + 1] * alpha + pr * __builtin_bit_cast(float, vw[j] & 0xffff0000u);
      }
    }
  }
  // merge the four row groups (lanes c, c+16, c+32, c+48 hold partial results for the same 8 output columns)
  float m_all = fmaxf(m_run, __shfl_xor(m_run, 16, 64));
  m_all = fmaxf(m_all, __shfl_xor(m_all, 32, 64));
  const float w = __builtin_amdgcn_exp2f(m_run - m_all);
  float l = l_run * w;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = o[j] * w;
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    o[j] = x;
  }
  if (!SPLIT) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] /= l;
    if (g == 0) store8(p.o + (long)b * p.o_sb + (long)h * p.o_sh + c * 8, o);
    return;
  }
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s_o[wave][c * 8 + j] = o[j];
    if (c == 0) { s_ml[wave][0] = m_all; s_ml[wave][1] = l; }
  }
  __syncthreads();
  if (wave == 0 && g == 0) {
    float M = s_ml[0][0];
#pragma unroll
    for (int w4 = 1; w4 < NW; ++w4) M = fmaxf(M, s_ml[w4][0]);
    float L = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w4 = 0; w4 < NW; ++w4) {
      const float f = __builtin_amdgcn_exp2f(s_ml[w4][0] - M);   // a wave that saw no key carries max -1e30, sum 0
      L += s_ml[w4][1] * f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += s_o[w4][c * 8 + j] * f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] /= L;
    store8(p.o + (long)b * p.o_sb + (long)h * p.o_sh + c * 8, acc);
  }
}

}  // namespace

// q/k/v/o: bf16; strides in elements (batch, head, token). d % 8 == 0, d <= 128.
// causal != 0: key j visible to query i iff j <= i + q_pos0.
// relh/relw (may be null): fp32 [B*H][Nq][S] decomposed rel-pos terms; key index -> (kh, kw) = (j / S, j % S).
static int attention_bf16_impl(const void* q, long q_sb, long q_sh, long q_st,
                               const void* k, long k_sb, long k_sh, long k_st,
                               const void* v, long v_sb, long v_sh, long v_st,
                               void* o, long o_sb, long o_sh, long o_st,
                               int B, int H, int Nq, int Nk, int d, float scale,
                               int causal, int q_pos0,
                               const float* relh, const float* relw, int S, const int* nk_rows, void* stream,
                               float* lse = nullptr) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || d <= 0 || d > 128 || (d & 7)) return HAFF_ERR_BAD_ARG;
  if ((q_st & 7) || (k_st & 7) || (v_st & 7) || (o_st & 3) || (q_sh & 7) || (k_sh & 7) || (v_sh & 7) || (o_sh & 3) ||
      (q_sb & 7) || (k_sb & 7) || (v_sb & 7) || (o_sb & 3))
    return HAFF_ERR_BAD_ARG;
  const bool rel = relh != nullptr && relw != nullptr;
  if (rel && (causal || S <= 0 || (Nk % S) != 0)) return HAFF_ERR_BAD_ARG;
  AttnArgs p{reinterpret_cast<const bf16_t*>(q), reinterpret_cast<const bf16_t*>(k), reinterpret_cast<const bf16_t*>(v),
             reinterpret_cast<bf16_t*>(o), q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st,
             B, H, Nq, Nk, d, scale, q_pos0, relh, relw, S, nk_rows, nullptr, nullptr, nullptr};
  p.lse = lse;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (!lse && !rel && Nq == 1 && d == DEC_D && (!causal || q_pos0 >= Nk - 1) && (o_sh & 7) == 0 && (o_sb & 7) == 0 &&
      (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
    if (B * H <= 128) hipLaunchKernelGGL((attn_decode_kernel<true, false, 16>), dim3(B * H), dim3(1024), 0, s, p);
    else if (B * H <= 1024) hipLaunchKernelGGL(attn_decode_kernel<true>, dim3(B * H), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_decode_kernel<false>, dim3((B * H + 3) / 4), dim3(256), 0, s, p);
    return haff_check_launch();
  }
  const int dp = d <= 64 ? 64 : (d <= 96 ? 96 : 128);
  if (rel) {
    const int mode = (S == 64) ? 2 : (S <= 16 && Nk == S * S ? 3 : 1);
    if (mode == 1 && S > 32) return HAFF_ERR_UNSUPPORTED;
    if (mode == 3) {
      if (dp == 64) return launch_attn<64, 3, false>(p, s);
      if (dp == 96) return launch_attn<96, 3, false>(p, s);
      return launch_attn<128, 3, false>(p, s);
    }
    if (dp == 64) return mode == 2 ? launch_attn<64, 2, false>(p, s) : launch_attn<64, 1, false>(p, s);
    if (dp == 96 && mode == 2 && d == 80) {   // SAM global blocks
#ifdef HAFF_TUNING
      if (getenv("HAFF_ATTN_NO_PP")) return launch_attn<96, 2, false, 6, true, true>(p, s);
#endif
      if (attn_global_pp_ok(p, false)) return launch_attn_global_pp<false>(p, s);
      if ((Nq % 128) == 0 && (Nk % KT) == 0 && !nk_rows) return launch_attn<96, 2, false, 6, true, true>(p, s);
      return launch_attn<96, 2, false, 6, true>(p, s);
    }
    if (dp == 96) return mode == 2 ? launch_attn<96, 2, false>(p, s) : launch_attn<96, 1, false>(p, s);
    return mode == 2 ? launch_attn<128, 2, false>(p, s) : launch_attn<128, 1, false>(p, s);
  }
  if (causal) {
    if (dp == 64) return launch_attn<64, 0, true>(p, s);
    if (dp == 96) return launch_attn<96, 0, true>(p, s);
    return launch_attn<128, 0, true>(p, s);
  }
  if (dp == 64) return launch_attn<64, 0, false>(p, s);
  if (dp == 96) return launch_attn<96, 0, false>(p, s);
  return launch_attn<128, 0, false>(p, s);
}

extern "C" int haff_attention_bf16(const void* q, long q_sb, long q_sh, long q_st,
                                   const void* k, long k_sb, long k_sh, long k_st,
                                   const void* v, long v_sb, long v_sh, long v_st,
                                   void* o, long o_sb, long o_sh, long o_st,
                                   int B, int H, int Nq, int Nk, int d, float scale,
                                   int causal, int q_pos0,
                                   const float* relh, const float* relw, int S, void* stream) {
  return attention_bf16_impl(q, q_sb, q_sh, q_st, k, k_sb, k_sh, k_st, v, v_sb, v_sh, v_st, o, o_sb, o_sh, o_st, B, H, Nq, Nk, d,
                             scale, causal, q_pos0, relh, relw, S, nullptr, stream);
}

// haff_attention_bf16 that also returns the per-row log-sum-exp of the scores, LOG2 domain (log2 sum_k 2^(scale*log2(e)*q.k)), f32
// [B][H][Nq]: the forward half of the flash pair whose backward is haff_attention_bwd_bf16 (no probabilities are kept).
extern "C" int haff_attention_lse_bf16(const void* q, long q_sb, long q_sh, long q_st,
                                       const void* k, long k_sb, long k_sh, long k_st,
                                       const void* v, long v_sb, long v_sh, long v_st,
                                       void* o, long o_sb, long o_sh, long o_st,
                                       int B, int H, int Nq, int Nk, int d, float scale, int causal, int q_pos0,
                                       float* lse, void* stream) {
  if (!lse) return HAFF_ERR_BAD_ARG;
  return attention_bf16_impl(q, q_sb, q_sh, q_st, k, k_sb, k_sh, k_st, v, v_sb, v_sh, v_st, o, o_sb, o_sh, o_st, B, H, Nq, Nk, d,
                             scale, causal, q_pos0, nullptr, nullptr, 0, nullptr, stream, lse);
}

// SAM GLOBAL attention with the decomposed rel-pos bias computed inside the kernel (Attention.forward,
// image_encoder.py:235-260, + add_decomposed_rel_pos :354-392 at q_size == k_size == S x S): replaces haff_relpos_tables_bf16 +
// haff_attention_bf16 — no fp32 [B*H][N][S] tables are written or read. q/k/v/o: bf16 [B][H][S*S][d] views by strides, k and v in
// one fused row layout (same strides, v behind k); tab_*: bf16 [2S-1][d]. Supported geometry: S == 64, d == 80 (ViT-H global
// blocks); otherwise HAFF_ERR_UNSUPPORTED and the caller takes the two-kernel path.
extern "C" int haff_global_attention_bf16(const void* q, long q_sb, long q_sh, long q_st,
                                          const void* k, long k_sb, long k_sh, long k_st,
                                          const void* v, long v_sb, long v_sh, long v_st,
                                          void* o, long o_sb, long o_sh, long o_st,
                                          int B, int H, int S, int d, float scale,
                                          const void* tab_h, const void* tab_w, void* stream) {
  if (B <= 0 || H <= 0 || S <= 0 || d <= 0 || !tab_h || !tab_w) return HAFF_ERR_BAD_ARG;
  if ((q_st & 7) || (k_st & 7) || (v_st & 7) || (o_st & 3) || (q_sh & 7) || (k_sh & 7) || (v_sh & 7) || (o_sh & 3) ||
      (q_sb & 7) || (k_sb & 7) || (v_sb & 7) || (o_sb & 3))
    return HAFF_ERR_BAD_ARG;
  AttnArgs p{reinterpret_cast<const bf16_t*>(q), reinterpret_cast<const bf16_t*>(k), reinterpret_cast<const bf16_t*>(v),
             reinterpret_cast<bf16_t*>(o), q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st,
             B, H, S * S, S * S, d, scale, 0, nullptr, nullptr, S, nullptr, nullptr, nullptr, nullptr,
             reinterpret_cast<const bf16_t*>(tab_h), reinterpret_cast<const bf16_t*>(tab_w)};
  if (!attn_global_pp_ok(p, true)) return HAFF_ERR_UNSUPPORTED;
  return launch_attn_global_pp<true>(p, reinterpret_cast<hipStream_t>(stream));
}

// KV-cached decode over RAGGED caches (batched prompts of different lengths): one query per (batch, head), batch b
// attends its first nk_rows[b] cached keys (device int32 [B], each in 1..Nk; Nk = the cache view's extent). No mask, no
// bias: the query is the newest position, every cached key is visible (transformers LlamaAttention with a KV cache).
extern "C" int haff_attention_decode_rows_bf16(const void* q, long q_sb, long q_sh,
                                               const void* k, long k_sb, long k_sh, long k_st,
                                               const void* v, long v_sb, long v_sh, long v_st,
                                               void* o, long o_sb, long o_sh,
                                               int B, int H, int Nk, int d, float scale, const int* nk_rows, void* stream) {
  if (!nk_rows) return HAFF_ERR_BAD_ARG;
  return attention_bf16_impl(q, q_sb, q_sh, (long)H * d, k, k_sb, k_sh, k_st, v, v_sb, v_sh, v_st, o, o_sb, o_sh, (long)H * d, B, H, 1,
                             Nk, d, scale, 0, 0, nullptr, nullptr, 0, nk_rows, stream);
}

// One KV-cached decode position per (batch, head) with RoPE and the cache append fused in — replaces haff_rope_cache_rows +
// haff_attention_decode_rows_bf16 (two launches per layer per generated token) on the Llama decode path (transformers
// LlamaAttention.forward with a KV cache, reached from llava_llama.py:93-102; rotate-half RoPE, theta from cos_sin).
// qkv: [B][ld] rows holding q | k | v of the new position (H heads x d each, RAW: not rotated); kcache / vcache:
// [B][Tmax][H*d]; cos_sin f32 [Tmax][d] = cos(0..d/2) | sin(0..d/2); nk_rows[b] = position of the new token + 1
// (DEVICE int32 [B]): the new k (rotated) and v are written at cache row nk_rows[b] - 1 and attended with every older row.
// out: [B][H*d]. d must be 128. Bit-identical to the two-kernel path.
extern "C" int haff_decode_attention_rope_rows_bf16(const void* qkv, long ld, void* kcache, void* vcache, const float* cos_sin,
                                                    void* out, int B, int H, int d, int Tmax, float scale, const int* nk_rows,
                                                    void* stream) {
  if (B <= 0 || H <= 0 || d != DEC_D || Tmax <= 0 || !nk_rows || !cos_sin || (ld & 7)) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(kcache) & 15) ||
      (reinterpret_cast<uintptr_t>(vcache) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return HAFF_ERR_BAD_ARG;
  const bf16_t* q = reinterpret_cast<const bf16_t*>(qkv);
  const long hd = (long)H * d;
  AttnArgs p{q, reinterpret_cast<const bf16_t*>(kcache), reinterpret_cast<const bf16_t*>(vcache), reinterpret_cast<bf16_t*>(out),
             ld, d, hd, (long)Tmax * hd, d, hd, (long)Tmax * hd, d, hd, hd, d, hd,
             B, H, 1, Tmax, d, scale, 0, nullptr, nullptr, 0, nk_rows, q + hd, q + 2 * hd, cos_sin};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (B * H <= 128) hipLaunchKernelGGL((attn_decode_kernel<true, true, 16>), dim3(B * H), dim3(1024), 0, s, p);
  else if (B * H <= 1024) hipLaunchKernelGGL((attn_decode_kernel<true, true>), dim3(B * H), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((attn_decode_kernel<false, true>), dim3((B * H + 3) / 4), dim3(256), 0, s, p);
  return haff_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Decomposed rel-pos tables (image_encoder.py:376-384): relh[bh][q][kh] = q_vec . Rh[qh - kh + S - 1],
// relw[bh][q][kw] = q_vec . Rw[qw - kw + S - 1], with the UNSCALED q (image_encoder.py:244-248).
// q: bf16 (dtype 0) or f32 (dtype 1) with (batch, head, token) strides; tables: fp32 [2S-1][d]; outputs fp32 [B*H][N][S], N = S*S.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void relpos_tables_kernel(const T* q, long q_sb, long q_sh, long q_st,
                                                          const float* tab_h, const float* tab_w,
                                                          float* relh, float* relw, int H, int S, int d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sq = reinterpret_cast<float*>(smem_raw);  // [QPB][d+1]
  const int N = S * S;
  const int QPB = 32;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QPB;
  const T* qb = q + (long)b * q_sb + (long)h * q_sh;
  for (int i = threadIdx.x; i < QPB * d; i += 256) {
    const int ql = i / d, c = i - ql * d;
    const int qi = min(q0 + ql, N - 1);
    sq[ql * (d + 1) + c] = elem<T>::ld(qb + (long)qi * q_st + c);
  }
  __syncthreads();
  const long bh = (long)b * H + h;
  for (int i = threadIdx.x; i < QPB * 2 * S; i += 256) {
    const int ql = i / (2 * S);
    const int j = i - ql * 2 * S;
    const int qi = q0 + ql;
    if (qi >= N) continue;
    const int qh = qi / S, qw = qi - qh * S;
    const bool is_h = j < S;
    const int kk = is_h ? j : j - S;
    const float* trow = (is_h ? tab_h + (long)(qh - kk + S - 1) * d : tab_w + (long)(qw - kk + S - 1) * d);
    float acc = 0.f;
    for (int c = 0; c < d; ++c) acc += sq[ql * (d + 1) + c] * trow[c];
    (is_h ? relh : relw)[(bh * N + qi) * S + kk] = acc;
  }
}
}  // namespace

// bf16 MFMA version (throughput mode). Per wave: 16 queries. P_h[q][r] = q . Th[r], P_w[q][r] = q . Tw[r] for ALL
// table rows r < 2S-1 come out of v_mfma_f32_16x16x32_bf16 (A = table rows, B = query rows, head dim zero-padded
// to a multiple of 32), land in LDS, and the diagonal gather relh[q][kh] = P_h[q][qh - kh + S - 1] (same for w)
// writes coalesced rows. ~50 MFMAs per 16 queries at S=64 instead of 10k scalar FMAs per query.
namespace {
constexpr int RELPOS_QT = 4;   // 16-query tiles per wave: a table's fragments are fetched once and reused for all of them

template <int NKD>  // k-steps of 32 over the (padded) head dim
__global__ __launch_bounds__(256) void relpos_tables_mfma_kernel(const bf16_t* q, long q_sb, long q_sh, long q_st,
                                                               const bf16_t* tab_h, const bf16_t* tab_w,
                                                               float* relh, float* relw, int H, int S, int d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = S * S;
  const int L = 2 * S - 1;
  const int RT = (L + 15) / 16;       // 16-row table tiles
  const int PS = RT * 16 + 4;         // LDS row stride (floats)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fh = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qbase = (blockIdx.x * 4 + wave) * (16 * RELPOS_QT);
  if (qbase >= N) return;
  // wave-private [16 queries][PS] image of ONE table's products for ONE query tile at a time (LDS ops of a wave
  // execute in order, so the gather sees the stores without a workgroup barrier): 8.4 KB per wave at S = 64
  float* sP = reinterpret_cast<float*>(smem_raw) + (long)wave * 16 * PS;
  const bf16_t* qb = q + (long)b * q_sb + (long)h * q_sh;
  bf16x8 qf[RELPOS_QT][NKD];
#pragma unroll
  for (int qt = 0; qt < RELPOS_QT; ++qt) {
    const int qi = min(qbase + qt * 16 + fr, N - 1);
#pragma unroll
    for (int kd = 0; kd < NKD; ++kd) {
      const int col = kd * 32 + fh * 8;
      uint4 r = make_uint4(0, 0, 0, 0);
      if (col < d) r = *reinterpret_cast<const uint4*>(qb + (long)qi * q_st + col);
      qf[qt][kd] = __builtin_bit_cast(bf16x8, r);
    }
  }
  const long bh = (long)b * H + h;
  const bool one_chunk = RT <= 8;   // S <= 64: the whole table (8 row tiles x NKD fragments) stays in registers
  uint4 tf[8][NKD];
  auto load_tf = [&](const bf16_t* tab, int rc) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int row = min((rc + g) * 16 + fr, L - 1);
#pragma unroll
      for (int kd = 0; kd < NKD; ++kd) {
        const int col = kd * 32 + fh * 8;
        tf[g][kd] = make_uint4(0, 0, 0, 0);
        if (col < d) tf[g][kd] = *reinterpret_cast<const uint4*>(tab + (long)row * d + col);
      }
    }
  };
  for (int tb = 0; tb < 2; ++tb) {
    const bf16_t* tab = tb == 0 ? tab_h : tab_w;
    float* out = tb == 0 ? relh : relw;
    if (one_chunk) load_tf(tab, 0);
#pragma unroll
    for (int qt = 0; qt < RELPOS_QT; ++qt) {
      const int q0 = qbase + qt * 16;
      if (q0 >= N) break;
      for (int rc = 0; rc < RT; rc += 8) {
        if (!one_chunk) load_tf(tab, rc);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          if (rc + g < RT) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kd = 0; kd < NKD; ++kd)
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, tf[g][kd]), qf[qt][kd], acc, 0, 0, 0);
            // D[i = table row 4fh + reg][j = query fr]
            float v[4] = {acc[0], acc[1], acc[2], acc[3]};
            store4(sP + (long)fr * PS + (rc + g) * 16 + fh * 4, v);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      // diagonal gather: term j of query (qh, qw) is product row (qh or qw) - j + S - 1
      if ((S & 3) == 0) {
        // 16-B stores: a lane gathers 4 consecutive terms (descending LDS addresses) of one query; a 16-lane row of
        // the wave covers one 256-B output row when S == 64 (4-byte stores ran this kernel at 1 TB/s of output)
        const int s4 = S >> 2;
        for (int idx = lane; idx < 16 * s4; idx += 64) {
          const int ql = idx / s4;
          const int j = (idx - ql * s4) << 2;
          const int qi = q0 + ql;
          if (qi >= N) continue;
          const int qh = qi / S, qw = qi - qh * S;
          const float* src = sP + (long)ql * PS + ((tb == 0 ? qh : qw) - j + S - 1);
          float v[4] = {src[0], src[-1], src[-2], src[-3]};
          store4(out + (bh * N + qi) * S + j, v);
        }
      } else {
        for (int idx = lane; idx < 16 * S; idx += 64) {
          const int ql = idx / S;
          const int j = idx - ql * S;
          const int qi = q0 + ql;
          if (qi >= N) continue;
          const int qh = qi / S, qw = qi - qh * S;
          out[(bh * N + qi) * S + j] = sP[(long)ql * PS + ((tb == 0 ? qh : qw) - j + S - 1)];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}
}  // namespace

// tab_h/tab_w: bf16 [2S-1][d]; q bf16; d % 8 == 0, d <= 128.
extern "C" int haff_relpos_tables_bf16(const void* q, long q_sb, long q_sh, long q_st, const void* tab_h,
                                       const void* tab_w, float* relh, float* relw, int B, int H, int S, int d,
                                       void* stream) {
  if (B <= 0 || H <= 0 || S <= 0 || d <= 0 || d > 128 || (d & 7) || (q_st & 7) || (q_sh & 7) || (q_sb & 7)) return HAFF_ERR_BAD_ARG;
  const int N = S * S;
  const int RT = (2 * S - 1 + 15) / 16;
  const size_t lds = (size_t)4 * 16 * (RT * 16 + 4) * sizeof(float);   // one table's products per wave at a time
  if (lds > 150 * 1024) return HAFF_ERR_UNSUPPORTED;
  dim3 grid((N + 64 * RELPOS_QT - 1) / (64 * RELPOS_QT), H, B), block(256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (lds > 64 * 1024) {  // S > 120
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relpos_tables_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relpos_tables_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relpos_tables_mfma_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relpos_tables_mfma_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const bf16_t* qp = reinterpret_cast<const bf16_t*>(q);
  const bf16_t* th = reinterpret_cast<const bf16_t*>(tab_h);
  const bf16_t* tw = reinterpret_cast<const bf16_t*>(tab_w);
  const int nkd = (d + 31) / 32;
  if (nkd == 1) hipLaunchKernelGGL((relpos_tables_mfma_kernel<1>), grid, block, lds, s, qp, q_sb, q_sh, q_st, th, tw, relh, relw, H, S, d);
  else if (nkd == 2) hipLaunchKernelGGL((relpos_tables_mfma_kernel<2>), grid, block, lds, s, qp, q_sb, q_sh, q_st, th, tw, relh, relw, H, S, d);
  else if (nkd == 3) hipLaunchKernelGGL((relpos_tables_mfma_kernel<3>), grid, block, lds, s, qp, q_sb, q_sh, q_st, th, tw, relh, relw, H, S, d);
  else hipLaunchKernelGGL((relpos_tables_mfma_kernel<4>), grid, block, lds, s, qp, q_sb, q_sh, q_st, th, tw, relh, relw, H, S, d);
  return haff_check_launch();
}

extern "C" int haff_relpos_tables(const void* q, long q_sb, long q_sh, long q_st,
                                  const float* tab_h, const float* tab_w, float* relh, float* relw,
                                  int B, int H, int S, int d, int dtype, void* stream) {
  if (B <= 0 || H <= 0 || S <= 0 || d <= 0) return HAFF_ERR_BAD_ARG;
  const int N = S * S;
  dim3 grid((N + 31) / 32, H, B), block(256);
  size_t lds = (size_t)32 * (d + 1) * sizeof(float);
  if (dtype == 0)
    hipLaunchKernelGGL((relpos_tables_kernel<bf16_t>), grid, block, lds, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16_t*>(q), q_sb, q_sh, q_st, tab_h, tab_w, relh, relw, H, S, d);
  else
    hipLaunchKernelGGL((relpos_tables_kernel<float>), grid, block, lds, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float*>(q), q_sb, q_sh, q_st, tab_h, tab_w, relh, relw, H, S, d);
  return haff_check_launch();
}
