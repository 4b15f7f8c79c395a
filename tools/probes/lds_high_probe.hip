// Probe: are LDS accesses above 64 KiB of a workgroup's allocation disturbed while OTHER workgroups are launched on the
// same CU (second stream, small kernels)?  Kernel `holder` keeps one 512-thread workgroup per CU busy for a while:
// it tags the low 64 KiB and the region above it differently and keeps re-reading both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void holder(int n_words, int iters, unsigned* errors, unsigned* first_bad) {
  extern __shared__ unsigned lds[];
  const unsigned tag = (blockIdx.x + 1u) << 20;
  for (int i = threadIdx.x; i < n_words; i += 512) lds[i] = tag | (unsigned)i;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < n_words; i += 512) {
      const unsigned v = lds[i];
      if (v != (tag | (unsigned)i)) {
        if (atomicAdd(errors, 1u) == 0) { first_bad[0] = blockIdx.x; first_bad[1] = i; first_bad[2] = v; first_bad[3] = it; }
      }
    }
    // rewrite the upper half each pass (a stale or aliased write shows up on the next read)
    for (int i = threadIdx.x + n_words / 2; i < n_words; i += 512) lds[i] = tag | (unsigned)i;
    __syncthreads();
  }
}
__global__ void small(float* x, int n, int lds_words) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < lds_words; i += blockDim.x) s[i] = (float)i;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] * 1.0001f + s[(threadIdx.x * 7) % (lds_words > 0 ? lds_words : 1)];
}
int main() {
  unsigned *err, *fb; float* x;
  CK(hipMalloc(&err, 4)); CK(hipMalloc(&fb, 16)); CK(hipMalloc(&x, 4 << 20));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(holder), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int sizes[] = {32768, 65536, 71680, 98304, 131072};
  for (int with_other = 0; with_other < 2; ++with_other)
    for (int sz : sizes) {
      CK(hipMemset(err, 0, 4)); CK(hipMemset(fb, 0, 16)); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(holder, dim3(256), dim3(512), sz, s1, sz / 4, 400, err, fb);
      if (with_other)
        for (int k = 0; k < 400; ++k) hipLaunchKernelGGL(small, dim3(2048), dim3(256), 8192, s2, x, 1 << 20, 2048);
      CK(hipDeviceSynchronize());
      unsigned h = 0, f[4];
      CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, fb, 16, hipMemcpyDeviceToHost));
      printf("holder lds %6d B  %s: mismatches %u", sz, with_other ? "small kernels on a second stream" : "alone                           ", h);
      if (h) printf("  first: block %u word %u (byte %u) pass %u read 0x%08x = block tag %u word %u (byte %u)", f[0], f[1], f[1] * 4, f[3], f[2], (f[2] >> 20) - 1, f[2] & 0xfffff, (f[2] & 0xfffff) * 4);
      printf("\n");
    }
  return 0;
}
