// Device-side frame ingest (rows a1 / a2 / f2): the two resizes the reference does on the host through Pillow, and
// CLIP's rescale + normalise, on uint8 NHWC frames already resident in HBM.
//
//   haff_resample_u8       : one axis of Pillow's antialiased resampling (src/libImaging/Resample.c,
//                            ImagingResampleHorizontal_8bpc / Vertical_8bpc): out = clip8((2^21 + sum in * k) >> 22)
//                            with the 22-bit fixed-point coefficient tables built on the host (preprocess.py:
//                            pil_resample_tables = precompute_coeffs + normalize_coeffs_8bpc). Horizontal pass first,
//                            uint8 intermediate, then vertical — bit-exact with Image.resize(BILINEAR / BICUBIC), i.e. with
//                            ResizeLongestSide.apply_image (segment_anything/utils/transforms.py:27-34) and with
//                            CLIPImageProcessor's resize (third-party transformers; call site inference.py:233-236).
//   haff_clip_normalize_u8 : centre crop + x/255 + (x - mean)/std as a 3x256 float LUT built in numpy's arithmetic,
//                            NHWC uint8 -> NCHW bf16/f32 (the images_clip tensor of LISA.py:432).
// Byte/integer work, HBM/L2-bound and tiny next to the encoders (3 MB per 1024^2 frame read once).
#include "haff_common.h"

namespace {

// axis 0: resample along W (rows kept): thread = one output pixel (3 channels); taps are contiguous bytes of one row.
__global__ __launch_bounds__(256) void resample_w_kernel(const unsigned char* in, unsigned char* out, long n_rows, int Win,
                                                         int Wout, const int* bounds, const int* coeffs, int ksize) {
  const long total = n_rows * Wout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % Wout);
    const long row = i / Wout;
    const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = coeffs + (long)xx * ksize;
    const unsigned char* p = in + (row * Win + x0) * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int t = 0; t < n; ++t) {
      const int kv = k[t];
      s0 += (int)p[3 * t] * kv;
      s1 += (int)p[3 * t + 1] * kv;
      s2 += (int)p[3 * t + 2] * kv;
    }
    unsigned char* o = out + i * 3;
    o[0] = (unsigned char)min(max(s0 >> 22, 0), 255);
    o[1] = (unsigned char)min(max(s1 >> 22, 0), 255);
    o[2] = (unsigned char)min(max(s2 >> 22, 0), 255);
  }
}

// axis 1: resample along H (columns kept): thread = one output byte (x, channel); a wave reads 64 consecutive bytes per tap.
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* in, unsigned char* out, int B, int Hin, int Hout,
                                                         long row_bytes, const int* bounds, const int* coeffs, int ksize) {
  const long total = (long)B * Hout * row_bytes;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long col = i % row_bytes;
    const long r = i / row_bytes;
    const int yy = (int)(r % Hout);
    const long b = r / Hout;
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = coeffs + (long)yy * ksize;
    const unsigned char* p = in + (b * Hin + y0) * row_bytes + col;
    int s = 1 << 21;
    for (int t = 0; t < n; ++t) s += (int)p[(long)t * row_bytes] * k[t];
    out[i] = (unsigned char)min(max(s >> 22, 0), 255);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void clip_normalize_kernel(const unsigned char* in, T* out, int B, int Hin, int Win, int top,
                                                             int left, int S, const float* lut) {
  const long total = (long)B * 3 * S * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % S), y = (int)((i / S) % S), c = (int)((i / ((long)S * S)) % 3);
    const long b = i / (3L * S * S);
    const unsigned char v = in[((b * Hin + top + y) * Win + left + x) * 3 + c];
    elem<T>::st(out + i, lut[c * 256 + v]);
  }
}

inline unsigned grid_1d(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

// in u8 [B][Hin][Win][3] -> out u8 [B][Hout][Wout][3]. axis 0: resample W (Hout must equal Hin), axis 1: resample H
// (Wout must equal Win). bounds int32 [n_out][2] = (first input index, tap count), coeffs int32 [n_out][ksize]
// (22-bit fixed point), both DEVICE pointers; every (first + count) must be <= the input extent (checked by the caller
// that built the tables: preprocess.pil_resample_tables).
extern "C" int haff_resample_u8(const void* in, void* out, int B, int Hin, int Win, int Hout, int Wout, int axis,
                                const int* bounds, const int* coeffs, int ksize, void* stream) {
  if (B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || ksize <= 0 || !bounds || !coeffs) return HAFF_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (axis == 0) {
    if (Hout != Hin) return HAFF_ERR_BAD_ARG;
    const long rows = (long)B * Hin;
    hipLaunchKernelGGL(resample_w_kernel, dim3(grid_1d(rows * Wout)), dim3(256), 0, s, (const unsigned char*)in,
                       (unsigned char*)out, rows, Win, Wout, bounds, coeffs, ksize);
  } else if (axis == 1) {
    if (Wout != Win) return HAFF_ERR_BAD_ARG;
    const long row_bytes = (long)Win * 3;
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_1d((long)B * Hout * row_bytes)), dim3(256), 0, s, (const unsigned char*)in,
                       (unsigned char*)out, B, Hin, Hout, row_bytes, bounds, coeffs, ksize);
  } else {
    return HAFF_ERR_BAD_ARG;
  }
  return haff_check_launch();
}

// in u8 [B][Hin][Win][3]; the S x S window at (top, left) -> out [B][3][S][S] (out_dtype 0 = bf16, 1 = f32) through
// lut f32 [3][256] (DEVICE pointer).
extern "C" int haff_clip_normalize_u8(const void* in, void* out, int B, int Hin, int Win, int top, int left, int S,
                                      const float* lut, int out_dtype, void* stream) {
  if (B <= 0 || S <= 0 || top < 0 || left < 0 || top + S > Hin || left + S > Win || !lut) return HAFF_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned g = grid_1d((long)B * 3 * S * S);
  if (out_dtype == 0)
    hipLaunchKernelGGL((clip_normalize_kernel<bf16_t>), dim3(g), dim3(256), 0, s, (const unsigned char*)in, (bf16_t*)out, B, Hin, Win, top, left, S, lut);
  else if (out_dtype == 1)
    hipLaunchKernelGGL((clip_normalize_kernel<float>), dim3(g), dim3(256), 0, s, (const unsigned char*)in, (float*)out, B, Hin, Win, top, left, S, lut);
  else
    return HAFF_ERR_BAD_ARG;
  return haff_check_launch();
}
