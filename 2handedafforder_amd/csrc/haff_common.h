// Shared device helpers for the 2HandedAfforder MI355X (gfx950 / CDNA4) hot path.
// Wave = 64 lanes everywhere; bf16 is carried as raw 16-bit payloads (unsigned short).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HAFF_OK 0
#define HAFF_ERR_BAD_ARG (-1)
#define HAFF_ERR_UNSUPPORTED (-2)
#define HAFF_ERR_LAUNCH (-3)

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) short bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define HAFF_WAVE 64

// activation codes shared by the GEMM epilogues (see include/haff_hip.h)
#define HAFF_ACT_NONE 0
#define HAFF_ACT_GELU 1        // exact erf GELU (SAM encoder MLP, upscaler)
#define HAFF_ACT_QUICK_GELU 2  // x*sigmoid(1.702x) (CLIP)
#define HAFF_ACT_RELU 3        // SAM decoder MLPs
#define HAFF_ACT_SILU 4

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even; a plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
// two floats -> one dword of two bf16 with ONE v_cvt_pk_bf16_f32 (converting them separately and or-ing the halves
// costs 3-4 VALU ops per pair: measured as a third of the attention kernels' VALU work)
typedef float haff_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 haff_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const haff_f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, haff_bf16x2));
}

template <typename T> struct elem;
template <> struct elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 8-element vector load/store (16 B for bf16, 32 B for f32) into fp32 registers
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  uint4 r = *reinterpret_cast<const uint4*>(p);
  unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  uint4 r;
  r.x = pack_bf16x2(v[0], v[1]);
  r.y = pack_bf16x2(v[2], v[3]);
  r.z = pack_bf16x2(v[4], v[5]);
  r.w = pack_bf16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = r;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
  uint2 r;
  r.x = pack_bf16x2(v[0], v[1]);
  r.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = r;
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case HAFF_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    case HAFF_ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
    case HAFF_ACT_RELU: return fmaxf(x, 0.0f);
    case HAFF_ACT_SILU: return x / (1.0f + __expf(-x));
    default: return x;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int haff_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? HAFF_OK : HAFF_ERR_LAUNCH;
}
