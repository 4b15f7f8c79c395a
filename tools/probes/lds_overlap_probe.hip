// Probe: do the LDS allocations of two co-resident workgroups overlap when each asks for more than 64 KiB?
// Each workgroup fills its dynamic LDS with a tag, idles, then re-reads it; mismatches are counted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void probe(int n_words, int spin, unsigned* errors, unsigned* first_bad) {
  extern __shared__ unsigned lds[];
  const unsigned tag = (blockIdx.x + 1u) << 18;
  for (int rep = 0; rep < 4; ++rep) {
    for (int i = threadIdx.x; i < n_words; i += 256) lds[i] = tag | (unsigned)i;
    __syncthreads();
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    for (int i = threadIdx.x; i < n_words; i += 256) {
      const unsigned v = lds[i];
      if (v != (tag | (unsigned)i)) {
        if (atomicAdd(errors, 1u) == 0) { first_bad[0] = blockIdx.x; first_bad[1] = i; first_bad[2] = v; }
      }
    }
    __syncthreads();
  }
}
int main() {
  unsigned *err, *fb;
  hipMalloc(&err, 4); hipMalloc(&fb, 12);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int sizes[] = {55296, 65536, 66048, 71680, 81920, 107520};
  for (int s : sizes) {
    hipMemset(err, 0, 4); hipMemset(fb, 0, 12);
    hipLaunchKernelGGL(probe, dim3(2048), dim3(256), s, 0, s / 4, 2000, err, fb);
    hipError_t e = hipDeviceSynchronize();
    unsigned h = 0, f[3];
    hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost); hipMemcpy(f, fb, 12, hipMemcpyDeviceToHost);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 256, s);
    printf("lds %6d B/WG  occupancy %d WG/CU  rc=%d  mismatches %u  first: block %u word %u value 0x%x (block tag %u, word %u)\n",
           s, occ, (int)e, h, f[0], f[1], f[2], (f[2] >> 18) - 1, f[2] & 0x3ffff);
  }
  return 0;
}
