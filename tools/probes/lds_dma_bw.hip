// Per-CU operand fetch rate probe: every workgroup repeatedly pulls `batch` KiB per wave (global_load_lds b128, 1 KiB
// per instruction) from a private window of `win_kb` KiB (L2-resident when small, HBM-streaming when large) and waits with
// vmcnt(0), the way the GEMM tiles stage operands. Prints GB/s per CU and B/clk/CU for workgroups/CU x waves x batch.
// hipcc --offload-arch=gfx950 -O3 -o lds_dma_bw lds_dma_bw.hip && ./lds_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BATCH, bool REG>
__global__ void probe(const char* src, long win_bytes, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const char* base = src + (long)blockIdx.x * win_bytes;
  long off = (long)wave * BATCH * 1024;
  const long step = (long)nw * BATCH * 1024;
  uint4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (off + BATCH * 1024 > win_bytes) off = (long)wave * BATCH * 1024;
    if constexpr (REG) {
      uint4 v[BATCH];
#pragma unroll
      for (int i = 0; i < BATCH; ++i) v[i] = *reinterpret_cast<const uint4*>(base + off + i * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < BATCH; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
    } else {
#pragma unroll
      for (int i = 0; i < BATCH; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off + i * 1024 + lane * 16), (lptr_t)(smem + (wave * BATCH + i) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    off += step;
  }
  if (REG && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1.f;
}

template <int BATCH, bool REG>
static void run(const char* src, long total_bytes, int wgs, int waves, long win_kb, int clk_mhz) {
  const long win = win_kb * 1024;
  if ((long)wgs * win > total_bytes) return;
  const int iters = 2000;
  float* sink;
  hipMalloc(&sink, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const size_t lds = REG ? 0 : (size_t)waves * BATCH * 1024;
  hipLaunchKernelGGL((probe<BATCH, REG>), dim3(wgs), dim3(waves * 64), lds, 0, src, win, 200, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<BATCH, REG>), dim3(wgs), dim3(waves * 64), lds, 0, src, win, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * waves * BATCH * 1024.0 * iters;
  const double per_cu = bytes / (ms * 1e-3) / (wgs < 256 ? wgs : 256.0);
  printf("%s wgs %4d waves %2d batch %2d KiB/wave window %6ld KiB/wg : %7.2f TB/s total, %6.1f GB/s per CU, %5.1f B/clk/CU\n",
         REG ? "reg" : "dma", wgs, waves, BATCH, win_kb, bytes / (ms * 1e-3) / 1e12, per_cu / 1e9, per_cu / (clk_mhz * 1e6));
  hipFree(sink);
}

int main() {
  const long total = 8L << 30;
  char* src;
  hipMalloc(&src, total);
  hipMemset(src, 1, total);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int clk = prop.clockRate / 1000;
  printf("%s, %d CUs, %d MHz\n", prop.name, prop.multiProcessorCount, clk);
  for (long win_kb : {64L}) {            // 64 KiB/wg: L2 (and L1-missing: > 32 KiB); 16 MiB/wg: HBM stream
    for (int wgs : {256, 512}) {
      for (int waves : {4, 8}) {
        if (wgs == 512 && waves == 8) continue;
        run<4, false>(src, total, wgs, waves, win_kb, clk);
        run<8, false>(src, total, wgs, waves, win_kb, clk);
        run<16, false>(src, total, wgs, waves, win_kb, clk);
        run<8, true>(src, total, wgs, waves, win_kb, clk);
        run<16, true>(src, total, wgs, waves, win_kb, clk);
      }
    }
  }
  for (int wgs : {1024, 2048}) run<8, true>(src, total, wgs, 4, 2048, clk);
  // footprints between the 32 MiB of L2 and the 256 MiB memory-side cache: re-read every pass (MALL-resident after the first)
  for (long win_kb : {256L, 512L, 768L, 1024L}) {
    run<16, true>(src, total, 256, 4, win_kb, clk);
    run<16, false>(src, total, 256, 4, win_kb, clk);
  }
  // few workgroups streaming from HBM: what one CU can pull on misses, by bytes in flight
  for (int wgs : {32, 96, 128, 192}) {
    run<4, false>(src, total, wgs, 4, 16384, clk);
    run<8, false>(src, total, wgs, 4, 16384, clk);
    run<16, false>(src, total, wgs, 4, 16384, clk);
    run<16, true>(src, total, wgs, 4, 16384, clk);
    run<16, false>(src, total, wgs, 8, 16384, clk);
  }
  hipFree(src);
  return 0;
}
