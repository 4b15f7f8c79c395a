// Which row strides make a 16-row x 64-byte ds_read_b128 fragment read conflict-free on MI355X? The MFMA operand pattern of the
// attention kernels: lane (fr = lane & 15, fh = lane >> 4) reads 16 bytes at row fr, byte column fh * 16 (+ a k-step offset) of a
// tile whose rows are STRIDE bytes apart. One wave per SIMD (4 waves per workgroup, one workgroup per CU) issues 64 such reads
// back to back, 200 times; cycles per read instruction from s_memtime. Patterns: linear (lane * 16: the ideal), the GEMM's
// 128-byte rows with the XOR swizzle, and plain strides 128 ... 288.
// hipcc --offload-arch=gfx950 -O3 -o lds_b128_pattern lds_b128_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe(int mode, int stride, int waves_active, unsigned long long* out, unsigned* sink) {
  extern __shared__ unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  if (wave >= waves_active) return;
  const int fr = lane & 15, fh = lane >> 4;
  unsigned addr;
  if (mode == 0) addr = lane * 16;
  else if (mode == 1) addr = fr * 128 + ((fh ^ (fr & 7)) << 4);
  else addr = fr * stride + fh * 16;
  addr += wave * 16384;   // each wave its own 16 KiB region
  unsigned acc = 0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < 200; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      u32x4 v;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
      acc += v[0];
    }
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}
// the same with 8 reads in flight before the wait (throughput rather than latency)
__global__ __launch_bounds__(256) void probe_tp(int mode, int stride, int waves_active, unsigned long long* out, unsigned* sink) {
  extern __shared__ unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  if (wave >= waves_active) return;
  const int fr = lane & 15, fh = lane >> 4;
  unsigned addr;
  if (mode == 0) addr = lane * 16;
  else if (mode == 1) addr = fr * 128 + ((fh ^ (fr & 7)) << 4);
  else addr = fr * stride + fh * 16;
  addr += wave * 16384;
  unsigned acc = 0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < 200; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(addr));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k][0];
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  unsigned long long* d;
  unsigned* sink;
  hipMalloc(&d, 256 * 4 * 8);
  hipMalloc(&sink, 64);
  std::vector<unsigned long long> h(256 * 4);
  struct C { const char* name; int mode, stride; };
  std::vector<C> cs = {{"linear (lane * 16)", 0, 0}, {"128-B rows, XOR swizzle (GEMM)", 1, 0}};
  static char names[32][32];
  int ni = 0;
  for (int s = 128; s <= 288; s += 16) { snprintf(names[ni], 32, "stride %d B", s); cs.push_back({names[ni], 2, s}); ++ni; }
  for (int waves = 1; waves <= 4; waves += 3)
    for (auto& c : cs) {
      double r[2];
      for (int tp = 0; tp < 2; ++tp) {
        hipMemset(d, 0, 256 * 4 * 8);
        if (tp == 0) hipLaunchKernelGGL(probe, dim3(256), dim3(256), 65536, 0, c.mode, c.stride, waves, d, sink);
        else hipLaunchKernelGGL(probe_tp, dim3(256), dim3(256), 65536, 0, c.mode, c.stride, waves, d, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(h.data(), d, 256 * 4 * 8, hipMemcpyDeviceToHost);
        double s = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { s += (double)h[b * 4 + w]; ++n; }
        r[tp] = s / n / (200.0 * (tp ? 8 : 16));
      }
      printf("%d wave(s) per CU  %-32s  %6.1f clocks per read, one at a time  %6.1f with 8 in flight\n", waves, c.name, r[0], r[1]);
    }
  return 0;
}
