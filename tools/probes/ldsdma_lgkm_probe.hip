// Probe: does an in-flight LDS-DMA (global_load_lds_dwordx4) hold up "s_waitcnt lgkmcnt(0)"?
// One wave issues 8 DMAs from cold HBM addresses, then times (s_memtime) three waits in a row:
//   lgkmcnt(0)  -> if this takes about as long as an HBM miss, LDS-DMA is counted on LGKM_CNT too
//   vmcnt(0)    -> the rest of the DMA latency
// A control wave does the same with plain global loads to registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
__global__ __launch_bounds__(64) void probe(const unsigned* src, long stride_words, unsigned long long* out, int mode, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[8192 + 64];
  const int lane = threadIdx.x;
  const unsigned* g = src + (size_t)blockIdx.x * stride_words * 8 + lane * 4;
  unsigned long long t0, t1, t2, t3;
  uint4 r[8];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  if (mode == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(g + i * stride_words), (lptr_t)(smem + i * 1024), 16, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(g + i * stride_words) : "memory");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");   // waits only for the s_memtime itself... or also for the DMA?
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3) :: "memory");
  if (mode != 0) { unsigned acc = 0; for (int i = 0; i < 8; ++i) acc += r[i].x; if (acc == 0x12345678u) sink[0] = acc; }
  else if (smem[lane * 16] == 0x7f && smem[4096 + lane] == 0x7e) sink[0] = 1;
  if (lane == 0) { out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = t2 - t1; out[blockIdx.x * 4 + 2] = t3 - t2; }
}
int main() {
  const long stride_words = (1 << 20);           // 4 MiB between the 8 pieces of a block: cold lines
  const int blocks = 16;
  unsigned* src; unsigned long long* out; unsigned* sink;
  hipMalloc(&src, (size_t)blocks * stride_words * 8 * 4 + 4096); hipMalloc(&out, blocks * 32); hipMalloc(&sink, 4);
  hipMemset(src, 1, (size_t)blocks * stride_words * 8 * 4 + 4096);
  for (int mode = 0; mode < 2; ++mode) {
    hipDeviceSynchronize();
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, src, stride_words, out, mode, sink);
    hipDeviceSynchronize();
    unsigned long long h[64]; hipMemcpy(h, out, blocks * 32, hipMemcpyDeviceToHost);
    double a = 0, b = 0, c = 0;
    for (int i = 0; i < blocks; ++i) { a += h[i * 4]; b += h[i * 4 + 1]; c += h[i * 4 + 2]; }
    printf("%s: issue + first lgkmcnt(0) %6.0f clk | second memtime+lgkmcnt(0) %6.0f clk | vmcnt(0) %6.0f clk  (s_memtime clocks, mean of %d waves)\n",
           mode == 0 ? "8 x global_load_lds_dwordx4 (cold)" : "8 x global_load_dwordx4 (cold)    ", a / blocks, b / blocks, c / blocks, blocks);
  }
  return 0;
}
