// SAM mask-decoder tail and mask post-processing for the 2Haff hot path (gfx950).
//
//   haff_upscale_mask : the part of MaskDecoder.predict_masks after the first transposed conv
//       (mask_decoder.py:54-64,153-165): LayerNorm2d(64) -> GELU -> ConvTranspose2d(64->32,k2,s2) -> GELU ->
//       dot with the hypernetwork vector of mask token 0 (multimask_output=False keeps only mask 0,
//       mask_decoder.py:110-114). k=s=2 transposed convs do not overlap, so each input pixel expands to its
//       own 4x4 block of low-res mask logits; the 32x256x256 upscaled embedding never touches HBM.
//       The first ConvTranspose2d(256->64) is a plain per-pixel GEMM done by haff_gemm_bf16 with columns
//       ordered (dy, dx, co); this kernel consumes its [pixels][4*64] output.
//   haff_resize_bilinear : F.interpolate(mode="bilinear", align_corners=False) with an input crop, the two
//       stages of Sam.postprocess_masks (sam.py:177-188), fp32.
//   haff_threshold_masks : the caller-side gating of inference.py:294-301 / chat.py:226 (sigmoid(m) > th,
//       i.e. m > logit(th)) -> 0/255 bytes.
#include "haff_common.h"

namespace {

constexpr int C1 = 64;   // channels after first transposed conv
constexpr int C2 = 32;   // channels after second transposed conv

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename T>
__global__ __launch_bounds__(256) void upscale_mask_kernel(const T* up1, const float* ln_w, const float* ln_b,
                                                          const float* w2, const float* b2, const float* hyper,
                                                          float* out, int n_prompts, int h, int w, float eps) {
  __shared__ __attribute__((aligned(16))) float sW[C1 * 4 * C2];  // [co][tap2*32 + c2]  (32 KiB)
  __shared__ float sB[C2];
  for (int i = threadIdx.x; i < C1 * 4 * C2; i += 256) sW[i] = w2[i];
  if (threadIdx.x < C2) sB[threadIdx.x] = b2[threadIdx.x];
  __syncthreads();

  const long unit = (long)blockIdx.x * 256 + threadIdx.x;  // (prompt, pixel, tap1)
  const long total = (long)n_prompts * h * w * 4;
  if (unit >= total) return;
  const int tap1 = (int)(unit & 3);
  const long pix = unit >> 2;
  const int px = (int)(pix % w), py = (int)((pix / w) % h);
  const int pr = (int)(pix / ((long)w * h));

  float v[C1];
  const T* src = up1 + pix * (4 * C1) + tap1 * C1;
#pragma unroll
  for (int c = 0; c < C1; c += 8) {
    float t[8];
    load8(src + c, t);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[c + j] = t[j];
  }
  float mean = 0.f;
#pragma unroll
  for (int c = 0; c < C1; ++c) mean += v[c];
  mean *= (1.0f / C1);
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < C1; ++c) { const float d = v[c] - mean; var += d * d; }
  var *= (1.0f / C1);
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int c = 0; c < C1; ++c) v[c] = gelu_erf((v[c] - mean) * rstd * ln_w[c] + ln_b[c]);

  const float* hy = hyper + (long)pr * C2;
  float m[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int tap2 = 0; tap2 < 4; ++tap2) {
    float acc = 0.f;
#pragma unroll 1
    for (int c2 = 0; c2 < C2; c2 += 4) {
      float u0 = sB[c2], u1 = sB[c2 + 1], u2 = sB[c2 + 2], u3 = sB[c2 + 3];
#pragma unroll
      for (int co = 0; co < C1; ++co) {
        const float4 ww = *reinterpret_cast<const float4*>(&sW[co * (4 * C2) + tap2 * C2 + c2]);
        u0 += v[co] * ww.x; u1 += v[co] * ww.y; u2 += v[co] * ww.z; u3 += v[co] * ww.w;
      }
      acc += hy[c2] * gelu_erf(u0) + hy[c2 + 1] * gelu_erf(u1) + hy[c2 + 2] * gelu_erf(u2) + hy[c2 + 3] * gelu_erf(u3);
    }
    m[tap2] = acc;
  }
  const int dy = tap1 >> 1, dx = tap1 & 1;
  const int W4 = 4 * w;
  float* o = out + (long)pr * (4 * h) * W4;
#pragma unroll
  for (int tap2 = 0; tap2 < 4; ++tap2) {
    const int yy = 4 * py + 2 * dy + (tap2 >> 1);
    const int xx = 4 * px + 2 * dx + (tap2 & 1);
    o[(long)yy * W4 + xx] = m[tap2];
  }
}

__global__ void resize_bilinear_kernel(const float* in, float* out, int N, int Hs, int Ws, int Hc, int Wc, int Ho,
                                       int Wo) {
  const float sh = (float)Hc / (float)Ho, sw = (float)Wc / (float)Wo;
  const long total = (long)N * Ho * Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
    const long n = i / ((long)Wo * Ho);
    float fy = sh * ((float)oy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
    float fx = sw * ((float)ox + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
    int y0 = (int)fy; y0 = y0 > Hc - 1 ? Hc - 1 : y0;
    int x0 = (int)fx; x0 = x0 > Wc - 1 ? Wc - 1 : x0;
    const int y1 = y0 + (y0 < Hc - 1 ? 1 : 0), x1 = x0 + (x0 < Wc - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p = in + n * (long)Hs * Ws;
    const float a = p[(long)y0 * Ws + x0], b = p[(long)y0 * Ws + x1];
    const float c = p[(long)y1 * Ws + x0], d = p[(long)y1 * Ws + x1];
    out[i] = hy * (hx * a + lx * b) + ly * (hx * c + lx * d);
  }
}

__global__ void threshold_masks_kernel(const float* in, unsigned char* out, long total, float logit_th) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    out[i] = in[i] > logit_th ? 255 : 0;
}

struct GateArgs {
  const float* in;
  unsigned char* out;   // [n_th][plane_stride], the first `total` bytes of a plane are written
  long total, plane_stride;
  const float* taxonomy;  // device, 4 class probabilities of this prompt, or nullptr (gate always open)
  int blank_class, n_th, on_value;
  float th[8];
};

// planes[t][i] = (argmax(taxonomy) != blank_class && in[i] > th[t]) ? on_value : 0 — one read of the fp32 logits for all
// thresholds; 4 pixels per thread (16-B load, 4-B store per plane). argmax ties resolve to the first maximum (torch).
__global__ __launch_bounds__(256) void gate_threshold_kernel(GateArgs p) {
  bool open = true;
  if (p.taxonomy) {
    int best = 0;
    float bv = p.taxonomy[0];
#pragma unroll
    for (int c = 1; c < 4; ++c) {
      const float v = p.taxonomy[c];
      if (v > bv) { bv = v; best = c; }
    }
    open = best != p.blank_class;
  }
  const unsigned on = open ? (unsigned)p.on_value : 0u;
  const long n4 = p.total >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(p.in)[i];
    for (int t = 0; t < p.n_th; ++t) {
      const float th = p.th[t];
      const unsigned w = (v.x > th ? on : 0u) | (v.y > th ? on << 8 : 0u) | (v.z > th ? on << 16 : 0u) | (v.w > th ? on << 24 : 0u);
      reinterpret_cast<unsigned*>(p.out + (long)t * p.plane_stride)[i] = w;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (p.total & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    for (int t = 0; t < p.n_th; ++t) p.out[(long)t * p.plane_stride + i] = p.in[i] > p.th[t] ? (unsigned char)on : 0;
  }
}

}  // namespace

// up1: [n_prompts*h*w][4*64] (dtype 0 bf16 / 1 f32), columns (dy*2+dx)*64+co, bias already added.
// w2: fp32 [64][4*32] with column (dy2*2+dx2)*32+c2; b2 fp32 [32]; hyper fp32 [n_prompts][32];
// out: fp32 [n_prompts][4h][4w] low-res mask logits.
extern "C" int haff_upscale_mask(const void* up1, const float* ln_w, const float* ln_b, const float* w2,
                                 const float* b2, const float* hyper, float* out, int n_prompts, int h, int w,
                                 float eps, int dtype, void* stream) {
  if (n_prompts <= 0 || h <= 0 || w <= 0) return HAFF_ERR_BAD_ARG;
  const long total = (long)n_prompts * h * w * 4;
  dim3 g((unsigned)((total + 255) / 256)), b(256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == 0) hipLaunchKernelGGL((upscale_mask_kernel<bf16_t>), g, b, 0, s, (const bf16_t*)up1, ln_w, ln_b, w2, b2, hyper, out, n_prompts, h, w, eps);
  else hipLaunchKernelGGL((upscale_mask_kernel<float>), g, b, 0, s, (const float*)up1, ln_w, ln_b, w2, b2, hyper, out, n_prompts, h, w, eps);
  return haff_check_launch();
}

// in: fp32 [N][Hs][Ws] of which the top-left [Hc][Wc] crop is resampled to out fp32 [N][Ho][Wo].
extern "C" int haff_resize_bilinear(const float* in, float* out, int N, int Hs, int Ws, int Hc, int Wc, int Ho, int Wo,
                                    void* stream) {
  if (N <= 0 || Hc <= 0 || Wc <= 0 || Hc > Hs || Wc > Ws || Ho <= 0 || Wo <= 0) return HAFF_ERR_BAD_ARG;
  const long total = (long)N * Ho * Wo;
  long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, N, Hs, Ws, Hc, Wc, Ho, Wo);
  return haff_check_launch();
}

extern "C" int haff_threshold_masks(const float* in, void* out, long total, float logit_th, void* stream) {
  if (total <= 0) return HAFF_ERR_BAD_ARG;
  long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(threshold_masks_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, (unsigned char*)out, total, logit_th);
  return haff_check_launch();
}

// Output gating + thresholds of the CLIs in one pass (inference.py:276-334, chat.py:226-253): `logits` fp32 [total]
// (one postprocessed mask), `planes` uint8 [n_th][plane_stride >= total, multiple of 4]; thresholds are LOGIT thresholds (host array of n_th <= 8
// floats; the caller maps sigmoid(m) > th to m > x*(th), see postprocess.sigmoid_logit_threshold); taxonomy = device
// pointer to the prompt's 4 class probabilities or NULL; a prompt whose argmax equals blank_class gets all-zero planes.
extern "C" int haff_gate_threshold_masks(const float* logits, void* planes, long total, long plane_stride,
                                         const float* thresholds_host, int n_th, int on_value, const float* taxonomy,
                                         int blank_class, void* stream) {
  if (total <= 0 || n_th <= 0 || n_th > 8 || on_value < 0 || on_value > 255 || !thresholds_host) return HAFF_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(logits) & 15) || (reinterpret_cast<uintptr_t>(planes) & 3) || plane_stride < total ||
      (plane_stride & 3))
    return HAFF_ERR_BAD_ARG;
  GateArgs p{logits, (unsigned char*)planes, total, plane_stride, taxonomy, blank_class, n_th, on_value, {}};
  for (int t = 0; t < n_th; ++t) p.th[t] = thresholds_host[t];
  long g = ((total >> 2) + 255) / 256;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(gate_threshold_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return haff_check_launch();
}
