// How fast can the CUs of an MI355X get a GEMM tile's epilogue out? One workgroup per CU (8 waves, 130 KiB of LDS claimed),
// every `period_us` each ACTIVE workgroup writes one 256 x 256 bf16 tile (128 KiB) of a [131072][3840] output in the register
// epilogue's access pattern (16-B stores, four lanes cover 64 contiguous bytes of a row, 16 rows per instruction, 16
// instructions per wave), then idles until its next slot. Measured per burst (100 MHz wall clock, wave 0): first store ->
// last store ISSUED (what the wave is blocked for) and -> all stores retired (vmcnt(0)), maximum over the workgroup's waves.
//   who is active:   all 256 | one XCD (32 CUs) | one CU per XCD | one CU | 8 / 16 CUs per XCD
//   phases:          1 = every active workgroup bursts at the same moment (what the tile kernel does: all CUs reach their
//                    epilogue together); P > 1 = the active workgroups of an XCD are split into P classes (slot % P) that
//                    burst period / P apart
//   store flavour:   plain | nt | sc1 | sc0 sc1
// hipcc --offload-arch=gfx950 -O3 -o store_burst store_burst.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void st16(char* p, u32x4 v) {
  if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512) void burst(char* out, long ldc_bytes, int tiles_n, int xcds, int per_xcd_stride, int phases,
                                             int period_ticks, int rounds, unsigned long long* t) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (xcd >= xcds || (slot % per_xcd_stride) != 0) return;
  if (tid == 0) lds[0] = 1;   // keep the allocation
  const int cls = (slot / per_xcd_stride) % phases;
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fh = lane >> 4;
  const int coff = 16 * (fh & 1) + 4 * (fh & 2);   // bf16 columns: the register epilogue's lane -> column map
  const unsigned long long t_launch = wall_clock64();
  for (int r = 0; r < rounds; ++r) {
    const unsigned long long due = t_launch + (unsigned long long)period_ticks * r + (unsigned long long)period_ticks * cls / phases + 200;
    while (wall_clock64() < due) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    const int tile = blockIdx.x + r * 256;
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    char* base = out + ((long)tm * 256 + wm * 128 + fr) * ldc_bytes + ((long)tn * 256 + wn * 64 + coff) * 2;
    const unsigned long long t0 = wall_clock64();
    u32x4 v = {(unsigned)tile, (unsigned)lane, (unsigned)r, 7u};
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j) st16<MODE>(base + (long)mi * 16 * ldc_bytes + 64 * j, v);
    const unsigned long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = wall_clock64();
    // the workgroup's LAST wave is what a tile boundary waits for: maximum over the 8 waves (wave 0, the oldest, is served first
    // and alone looks twice as fast as the CU is)
    unsigned* mx = reinterpret_cast<unsigned*>(lds + 64);
    if (tid < 2) mx[tid] = 0;
    __syncthreads();
    if (lane == 0) {
      atomicMax(&mx[0], (unsigned)(t1 - t0));
      atomicMax(&mx[1], (unsigned)(t2 - t0));
    }
    __syncthreads();
    if (tid == 0) {
      t[(blockIdx.x * rounds + r) * 2 + 0] = mx[0];
      t[(blockIdx.x * rounds + r) * 2 + 1] = mx[1];
    }
  }
}


// Second experiment: the register epilogue's rhythm. Every wave alternates `valu` dependent-free v_fma_f32 (4 cycles each) with
// `group` stores, 16 stores per wave and burst in all (one 256 x 256 tile per workgroup as above). Measured per burst on wave 0
// and wave 4 (SIMD partners): first instruction -> last store issued, and the VALU-only time of the same loop without stores.
// ORDER 0: the two 64-B halves of a row's 128-B line are written 8 stores apart (column half outer, row pass inner);
// ORDER 1: the two halves by consecutive stores (row pass outer); ORDER 2: whole 128-B lines per instruction (8 lanes x 16 B per
// row, 8 rows per instruction).
template <int GROUP, bool STORES, int ORDER = 0>
__global__ __launch_bounds__(512) void rhythm(char* out, long ldc_bytes, int tiles_n, int valu, int period_ticks, int rounds,
                                              unsigned long long* t, float* sink) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) lds[0] = 1;
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fh = lane >> 4;
  const int coff = 16 * (fh & 1) + 4 * (fh & 2);
  const unsigned long long t_launch = wall_clock64();
  float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3;
  for (int r = 0; r < rounds; ++r) {
    const unsigned long long due = t_launch + (unsigned long long)period_ticks * r + 200;
    while (wall_clock64() < due) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    const int tile = blockIdx.x + r * 256;
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    char* base = out + ((long)tm * 256 + wm * 128 + fr) * ldc_bytes + ((long)tn * 256 + wn * 64 + coff) * 2;
    char* base2 = out + ((long)tm * 256 + wm * 128 + (lane >> 3)) * ldc_bytes + ((long)tn * 256 + wn * 64) * 2 + (lane & 7) * 16;
    (void)base2;
    const unsigned long long t0 = wall_clock64();
#pragma unroll 1
    for (int g = 0; g < 16 / GROUP; ++g) {
#pragma unroll 1
      for (int i = 0; i < valu * GROUP; i += 4) {
        asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      }
      if (STORES) {
        u32x4 v = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1), (unsigned)r, 7u};
#pragma unroll
        for (int k = 0; k < GROUP; ++k) {
          const int s = g * GROUP + k;   // store index 0..15
          if (ORDER == 0) st16<0>(base + (long)(s & 7) * 16 * ldc_bytes + 64 * (s >> 3), v);
          if (ORDER == 1) st16<0>(base + (long)(s >> 1) * 16 * ldc_bytes + 64 * (s & 1), v);
          if (ORDER == 2) st16<0>(base2 + (long)s * 8 * ldc_bytes, v);
        }
      }
    }
    const unsigned long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = wall_clock64();
    if (lane == 0 && (wave == 0 || wave == 4)) {
      t[((blockIdx.x * rounds + r) * 2 + (wave >> 2)) * 2 + 0] = t1 - t0;
      t[((blockIdx.x * rounds + r) * 2 + (wave >> 2)) * 2 + 1] = t2 - t0;
    }
  }
  if (a0 + a1 + a2 + a3 == 12345.f) sink[0] = a0;
}

// Third experiment: does the shape of a store instruction's footprint change what a CU can write? One workgroup per CU writes
// its 256 x 256 bf16 tile (128 KiB) with 16 dwordx4 stores per wave, the 1 KiB of an instruction laid out as
//   PAT 0: 16 rows x 64 B (register epilogue)   PAT 1: 8 rows x 128 B   PAT 2: 4 rows x 256 B   PAT 3: 2 rows x 512 B (whole tile rows)
//   PAT 4: 1 KiB contiguous (the tile as a dense 128 KiB block: what a tiled output layout would allow)
// max over the workgroup's waves of first store -> all retired.
template <int PAT>
__global__ __launch_bounds__(512) void footprint(char* out, long ldc_bytes, int tiles_n, int period_ticks, int rounds, unsigned long long* t) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned* mx = reinterpret_cast<unsigned*>(lds + 64);
  const unsigned long long t_launch = wall_clock64();
  for (int r = 0; r < rounds; ++r) {
    const unsigned long long due = t_launch + (unsigned long long)period_ticks * r + 200;
    while (wall_clock64() < due) __builtin_amdgcn_s_sleep(4);
    if (tid < 2) mx[tid] = 0;
    __syncthreads();
    const int tile = blockIdx.x + r * 256;
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    char* tile0 = out + (long)tm * 256 * ldc_bytes + (long)tn * 512;
    const unsigned long long t0 = wall_clock64();
    u32x4 v = {(unsigned)tile, (unsigned)lane, (unsigned)r, 7u};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      char* p;
      const int i = wave * 16 + s;   // instruction index in the tile, 0..127
      if (PAT == 0) p = tile0 + (long)((i >> 3) * 16 + (lane & 15)) * ldc_bytes + (i & 7) * 64 + (lane >> 4) * 16;
      if (PAT == 1) p = tile0 + (long)((i >> 2) * 8 + (lane >> 3)) * ldc_bytes + (i & 3) * 128 + (lane & 7) * 16;
      if (PAT == 2) p = tile0 + (long)((i >> 1) * 4 + (lane >> 4)) * ldc_bytes + (i & 1) * 256 + (lane & 15) * 16;
      if (PAT == 3) p = tile0 + (long)(i * 2 + (lane >> 5)) * ldc_bytes + (lane & 31) * 16;
      if (PAT == 4) p = out + (long)tile * 131072 + (long)i * 1024 + lane * 16;
      st16<0>(p, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = wall_clock64();
    if (lane == 0) atomicMax(&mx[1], (unsigned)(t2 - t0));
    __syncthreads();
    if (tid == 0) t[blockIdx.x * rounds + r] = mx[1];
  }
}

int main() {
  const long M = 131072, N = 3840;
  char* out;
  unsigned long long* t;
  const int rounds = 12;
  if (hipMalloc(&out, M * N * 2) != hipSuccess) return 1;
  hipMalloc(&t, 256 * rounds * 2 * sizeof(unsigned long long));
  std::vector<unsigned long long> h(256 * rounds * 2);
  struct Cfg { const char* name; int xcds, stride, phases; };
  const Cfg cfgs[] = {{"all 256 CUs, together", 8, 1, 1},       {"all 256 CUs, 2 phases per XCD", 8, 1, 2},
                      {"all 256 CUs, 4 phases per XCD", 8, 1, 4}, {"all 256 CUs, 8 phases per XCD", 8, 1, 8},
                      {"one XCD (32 CUs), together", 1, 1, 1},  {"16 CUs per XCD (128), together", 8, 2, 1},
                      {"8 CUs per XCD (64), together", 8, 4, 1}, {"1 CU per XCD (8)", 8, 32, 1},
                      {"1 CU", 1, 32, 1}};
  const char* modes[] = {"plain", "nt", "sc1", "sc0 sc1"};
  const int period_ticks = 3600;   // 36 us: one K = 1280 tile
  for (int mode = 0; mode < 3; mode += 2)
    for (const Cfg& c : cfgs) {
      hipMemset(t, 0, h.size() * 8);
      auto launch = [&](auto k) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 130 * 1024, 0, out, N * 2, (int)(N / 256), c.xcds, c.stride, c.phases, period_ticks, rounds, t);
      };
      if (mode == 0) launch(burst<0>);
      if (mode == 1) launch(burst<1>);
      if (mode == 2) launch(burst<2>);
      if (mode == 3) launch(burst<3>);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
      double si = 0, sd = 0, mi = 0, md = 0;
      int n = 0;
      for (int b = 0; b < 256; ++b)
        for (int r = 2; r < rounds; ++r) {
          const double a = h[(b * rounds + r) * 2] * 0.01, d = h[(b * rounds + r) * 2 + 1] * 0.01;
          if (d == 0) continue;
          si += a; sd += d; n++;
          if (a > mi) mi = a;
          if (d > md) md = d;
        }
      printf("%-8s %-34s issue %6.2f us (max %6.2f)  drained %6.2f us (max %6.2f)  = %5.1f GB/s per CU while bursting\n", modes[mode], c.name,
             si / n, mi, sd / n, md, 131072.0 / (sd / n) * 1e-3);
      fflush(stdout);
    }
  {
    const char* names[] = {"16 rows x 64 B", "8 rows x 128 B", "4 rows x 256 B", "2 rows x 512 B", "1 KiB contiguous"};
    for (int pat = 0; pat < 5; ++pat) {
      hipMemset(t, 0, h.size() * 8);
      auto launch = [&](auto k) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 130 * 1024, 0, out, N * 2, (int)(N / 256), period_ticks, rounds, t);
      };
      if (pat == 0) launch(footprint<0>);
      if (pat == 1) launch(footprint<1>);
      if (pat == 2) launch(footprint<2>);
      if (pat == 3) launch(footprint<3>);
      if (pat == 4) launch(footprint<4>);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
      double sd = 0; int n = 0;
      for (int b = 0; b < 256; ++b) for (int r = 2; r < rounds; ++r) { sd += h[b * rounds + r] * 0.01; n++; }
      printf("footprint: %-18s all 256 CUs: tile drained in %5.2f us = %5.1f GB/s per CU\n", names[pat], sd / n, 131072.0 / (sd / n) * 1e-3);
      fflush(stdout);
    }
  }
  {
    unsigned long long* t2;
    float* sink;
    hipMalloc(&t2, 256 * rounds * 4 * sizeof(unsigned long long));
    hipMalloc(&sink, 64);
    std::vector<unsigned long long> h2(256 * rounds * 4);
    const int valus[] = {0, 16, 32, 64};
    for (int valu : valus)
      for (int variant = 0; variant < 9; ++variant) {
        hipMemset(t2, 0, h2.size() * 8);
        auto launch = [&](auto k) {
          hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024);
          hipLaunchKernelGGL(k, dim3(256), dim3(512), 130 * 1024, 0, out, N * 2, (int)(N / 256), valu, period_ticks, rounds, t2, sink);
        };
        const char* name = "";
        if (variant == 0) { launch(rhythm<1, false>); name = "VALU only"; }
        if (variant == 1) { launch(rhythm<1, true>); name = "1 store per step"; }
        if (variant == 2) { launch(rhythm<2, true>); name = "2 stores per 2 steps"; }
        if (variant == 3) { launch(rhythm<4, true>); name = "4 stores per 4 steps"; }
        if (variant == 4) { launch(rhythm<16, true>); name = "16 stores at the end"; }
        if (variant == 5) { launch(rhythm<2, true, 1>); name = "line halves adjacent /2"; }
        if (variant == 6) { launch(rhythm<1, true, 1>); name = "line halves 1 step apart"; }
        if (variant == 7) { launch(rhythm<1, true, 2>); name = "whole lines, 1 per step"; }
        if (variant == 8) { launch(rhythm<2, true, 2>); name = "whole lines, 2 per 2"; }
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(h2.data(), t2, h2.size() * 8, hipMemcpyDeviceToHost);
        double s0 = 0, s1 = 0, d0 = 0, d1 = 0;
        int n = 0;
        for (int b = 0; b < 256; ++b)
          for (int r = 2; r < rounds; ++r) {
            const unsigned long long* e = &h2[(b * rounds + r) * 4];
            s0 += e[0] * 0.01; d0 += e[1] * 0.01; s1 += e[2] * 0.01; d1 += e[3] * 0.01; n++;
          }
        printf("rhythm: %3d v_fma per store, %-22s wave 0: loop %6.2f us, drained %6.2f | wave 4: loop %6.2f us, drained %6.2f\n", valu, name,
               s0 / n, d0 / n, s1 / n, d1 / n);
        fflush(stdout);
      }
  }
  return 0;
}
