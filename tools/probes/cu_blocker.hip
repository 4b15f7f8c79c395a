// Holds n CUs for a while: n workgroups of one wave, each claiming nearly all of a CU's LDS (so no two share a CU and no
// LDS-using kernel can join them), sleeping until `ticks` of the 100 MHz wall clock have passed. Lets a tool time OTHER kernels
// on the remaining 256 - n CUs (tools/decode_on_k_cus.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o cu_blocker.so cu_blocker.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void cu_blocker_kernel(unsigned long long ticks, unsigned int* sink) {
  extern __shared__ unsigned int lds[];
  lds[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (lds[(threadIdx.x + 1) & 63] == 12345u) sink[0] = 1;   // keep the LDS claim alive
}

extern "C" int cu_blocker_launch(int n_wgs, int lds_bytes, unsigned long long ticks, unsigned int* sink, void* stream) {
  static bool once = false;
  if (!once) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cu_blocker_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    once = true;
  }
  hipLaunchKernelGGL(cu_blocker_kernel, dim3(n_wgs), dim3(64), lds_bytes, static_cast<hipStream_t>(stream), ticks, sink);
  return (int)hipGetLastError();
}
