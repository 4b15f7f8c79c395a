// Probe: LDS-DMA (global_load_lds_dwordx4) double buffer + ds_read_b128 verify, the staging skeleton of the 256x256
// GEMM tile without the MFMAs, run beside small kernels launched on a second stream. Each "tile" t of the source
// holds the value (t+1) in every dword, so a read of a stale / future / other-buffer tile is identified exactly.
//   usage: lds_dma_probe             (sweeps buffer sizes below and above 64 KiB)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

// one buffer = chunks * 8 KiB (512 threads x 16 B per DMA instruction)
__global__ __launch_bounds__(512) void dma_loop(const unsigned* src, int chunks, int n_tiles, unsigned* errors, unsigned* first_bad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int buf_bytes = chunks * 8192;
  auto stage = [&](int buf, int t) {
    for (int i = 0; i < chunks; ++i) {
      const unsigned* g = src + ((size_t)t * chunks + i) * 2048 + tid * 4;              // 16 B per lane
      unsigned char* l = smem + buf * buf_bytes + i * 8192 + wave * 1024;                 // wave-uniform base
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
    }
  };
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (n_tiles > 1) stage(1, 1);
  for (int t = 0; t < n_tiles; ++t) {
    const int cur = t & 1;
    // every thread checks a different slice of the whole buffer (written by all waves)
    unsigned bad = 0, badv = 0, badi = 0;
    for (int i = 0; i < chunks; ++i) {
      const int off = cur * buf_bytes + i * 8192 + ((tid * 16 + 4096) & 8191);            // read another wave's lanes
      const uint4 v = *reinterpret_cast<const uint4*>(smem + off);
      const unsigned want = (unsigned)t + 1u;
      if (v.x != want || v.y != want || v.z != want || v.w != want) { bad++; badv = v.x != want ? v.x : (v.y != want ? v.y : (v.z != want ? v.z : v.w)); badi = off; }
    }
    if (bad) {
      if (atomicAdd(errors, bad) == 0) { first_bad[0] = blockIdx.x; first_bad[1] = t; first_bad[2] = badv; first_bad[3] = badi; }
    }
    if (t + 1 < n_tiles) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 2 < n_tiles) stage(cur, t + 2);
    }
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
  const int n_tiles = 512, max_chunks = 8;
  unsigned *err, *fb, *src; float* x;
  CK(hipMalloc(&err, 4)); CK(hipMalloc(&fb, 16)); CK(hipMalloc(&x, 4 << 20));
  CK(hipMalloc(&src, (size_t)n_tiles * max_chunks * 8192));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_loop), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int chunks : {2, 4, 5, 8}) {
    unsigned* h = (unsigned*)malloc((size_t)n_tiles * chunks * 8192);
    for (int t = 0; t < n_tiles; ++t) for (int i = 0; i < chunks * 2048; ++i) h[(size_t)t * chunks * 2048 + i] = t + 1;
    CK(hipMemcpy(src, h, (size_t)n_tiles * chunks * 8192, hipMemcpyHostToDevice)); free(h);
    for (int with_other = 0; with_other < 2; ++with_other) {
      CK(hipMemset(err, 0, 4)); CK(hipMemset(fb, 0, 16)); CK(hipDeviceSynchronize());
      for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(dma_loop, dim3(48), dim3(512), 2 * chunks * 8192, s1, src, chunks, n_tiles, err, fb);
        if (with_other) for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(small, dim3(4096), dim3(256), 8192, s2, x, 1 << 20, 2048);
      }
      CK(hipDeviceSynchronize());
      unsigned e = 0, f[4];
      CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, fb, 16, hipMemcpyDeviceToHost));
      printf("2 x %3d KiB buffers (%3d KiB LDS) %s: bad 16-B reads %u", chunks * 8, 2 * chunks * 8, with_other ? "beside small kernels" : "alone               ", e);
      if (e) printf("  first: block %u tile %u read value %u (tile %d) at LDS byte %u", f[0], f[1], f[2], (int)f[2] - 1, f[3]);
      printf("\n");
    }
  }
  return 0;
}
