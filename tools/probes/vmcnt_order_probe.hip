// Probe: do LDS-DMA operations (global_load_lds_dwordx4) leave a wave's vmcnt in ISSUE order on gfx950?
// DESIGN.md section 10a recorded a 128x128 GEMM loop whose counted s_waitcnt vmcnt(8) over LDS-DMA read stale LDS beside a
// second stream; the CDNA guides describe GEMM templates that rely on counted waits. This probe asks the hardware directly.
//
//  test A (one wave, no barrier): sentinel -> slot 0; DMA of a COLD 1-KiB line (HBM miss, address used once) -> slot 0;
//      NNEW DMAs of HOT lines (L2/L1 hits) or other cold lines -> slots 1..NNEW; s_waitcnt vmcnt(NNEW); ds_read slot 0.
//      In-order retirement => slot 0 always holds the cold line. A sentinel read => a newer DMA left vmcnt before the older.
//  test B (8 waves, ring of 3 x 32 KiB slots, barrier, readers verify chunks staged by OTHER waves): the wait before the
//      barrier is vmcnt(NI) (stage t+1 stays in flight across the barrier) or vmcnt(0) (control); run alone and beside the
//      (layernorm, qkv GEMM) aggressor on a second stream that exposed the GEMM failure.
// build: hipcc -O2 --offload-arch=gfx950 vmcnt_order_probe.hip -I../../include -L../../2handedafforder_amd/lib -lhaff_hip
//        -Wl,-rpath,'$ORIGIN/../../2handedafforder_amd/lib' -o vmcnt_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" {
#include "haff_hip.h"
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__device__ __forceinline__ unsigned cold_value(unsigned d) { return d * 2654435761u ^ 0x5bd1e995u; }

__global__ void fill_cold(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = cold_value((unsigned)i);
}
__global__ void fill_hot(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0x11110000u + (unsigned)i;
}

__device__ __forceinline__ uint4 lds_read16(unsigned off) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
  return v;
}

template <int NNEW, bool NEW_COLD>
__global__ __launch_bounds__(256) void order_same_wave(const unsigned* cold, unsigned cold_chunks, const unsigned* hot, int iters,
                                                      unsigned* errors, unsigned* info) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 9 * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* base = smem + wave * 9 * 1024;
  const unsigned slot0 = (unsigned)(uintptr_t)(lptr_t)(base + lane * 16);
  const unsigned gw = blockIdx.x * 4 + wave;
  unsigned long long rng = gw * 0x9E3779B97F4A7C15ull + 12345ull;
  unsigned bad = 0, bad_it = 0, bad_val = 0;
  for (int it = 0; it < iters; ++it) {
    *reinterpret_cast<uint4*>(base + lane * 16) = uint4{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    const unsigned chunk = __builtin_amdgcn_readfirstlane((unsigned)((rng >> 24) % cold_chunks));
    __builtin_amdgcn_global_load_lds((gptr_t)(cold + (size_t)chunk * 256 + lane * 4), (lptr_t)base, 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NNEW; ++i) {
      const unsigned* g = NEW_COLD ? cold + (size_t)((chunk + 7919u * (i + 1)) % cold_chunks) * 256 + lane * 4
                                   : hot + ((gw + i) & 63) * 256 + lane * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(base + (1 + i) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NNEW) : "memory");
    const uint4 v = lds_read16(slot0);
    const unsigned d = chunk * 256u + lane * 4;
    if (v.x != cold_value(d) || v.y != cold_value(d + 1) || v.z != cold_value(d + 2) || v.w != cold_value(d + 3)) {
      bad++; bad_it = it; bad_val = v.x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (bad && atomicAdd(errors, bad) == 0) { info[0] = gw; info[1] = bad_it; info[2] = bad_val; info[3] = lane; }
}

// test B: ring of 3 slots x 32 KiB, 512 threads. Stage t of block b: 256 rows x 128 B, rows one K-row (n_stages * 128 B) apart
// inside the block's own 256-row region; every dword of stage t holds t + 1.
template <bool COUNTED>
__global__ __launch_bounds__(512) void ring_loop(const unsigned* src, int n_stages, unsigned* errors, unsigned* info) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SLOT = 32768, NI = SLOT / (512 * 16);   // 4 DMA instructions per wave per stage
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned* region = src + (size_t)blockIdx.x * 256 * n_stages * 32;
  auto stage = [&](int slot, int t) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int id = i * 512 + tid;
      const unsigned* g = region + (size_t)(id >> 3) * (n_stages * 32) + t * 32 + (id & 7) * 4;
      unsigned char* l = smem + slot * SLOT + (i * 512 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
    }
  };
  stage(0, 0);
  if (n_stages > 1) stage(1, 1);
  unsigned bad = 0, badv = 0, badt = 0, bado = 0;
  for (int t = 0; t < n_stages; ++t) {
    if (COUNTED && t + 1 < n_stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");   // stage t landed; t+1 stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < n_stages) stage((t + 2) % 3, t + 2);   // slot of stage t-1: every wave is past the barrier, its reads are done
    const unsigned want = (unsigned)t + 1u;
    const unsigned sbase = (unsigned)(uintptr_t)(lptr_t)(smem + (t % 3) * SLOT);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned off = ((i * 512 + tid) * 16 + 16384 + 1024) % SLOT;   // chunks staged by other waves
      const uint4 v = lds_read16(sbase + off);
      if (v.x != want || v.y != want || v.z != want || v.w != want) { bad++; badv = v.x; badt = t; bado = off; }
    }
  }
  if (bad && atomicAdd(errors, bad) == 0) { info[0] = blockIdx.x; info[1] = badt; info[2] = badv; info[3] = bado; }
}

static int report(const char* name, unsigned* err, unsigned* info) {
  unsigned e = 0, f[4];
  CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, info, 16, hipMemcpyDeviceToHost));
  printf("%-64s bad 16-B reads %u", name, e);
  if (e) printf("  first: id %u iter/stage %u value 0x%x aux %u", f[0], f[1], f[2], f[3]);
  printf("\n"); fflush(stdout);
  return 0;
}

int main() {
  unsigned *err, *info, *cold, *hot, *src;
  const size_t cold_bytes = 4ull << 30;
  const unsigned cold_chunks = (unsigned)(cold_bytes / 1024);
  CK(hipMalloc(&err, 4)); CK(hipMalloc(&info, 16));
  CK(hipMalloc(&cold, cold_bytes)); CK(hipMalloc(&hot, 64 * 1024));
  hipLaunchKernelGGL(fill_cold, dim3(4096), dim3(256), 0, 0, cold, cold_bytes / 4);
  hipLaunchKernelGGL(fill_hot, dim3(16), dim3(256), 0, 0, hot, (size_t)16 * 1024);
  CK(hipDeviceSynchronize());

  // aggressor operands (the SAM block's layernorm + qkv GEMM through the product ABI)
  const int M = 9800;
  void *x, *hbuf, *q, *w; float *lw, *lb;
  CK(hipMalloc(&x, (size_t)M * 1280 * 2)); CK(hipMalloc(&hbuf, (size_t)M * 1280 * 2)); CK(hipMalloc(&q, (size_t)M * 3840 * 2));
  CK(hipMalloc(&w, (size_t)3840 * 1280 * 2)); CK(hipMalloc(&lw, 1280 * 4)); CK(hipMalloc(&lb, 1280 * 4));
  {
    std::vector<unsigned short> hx((size_t)M * 1280), hw((size_t)3840 * 1280);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0x3c00 + (unsigned short)(rand() & 0x3ff);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (unsigned short)(rand() & 0x3ff);
    std::vector<float> ones(1280, 1.f), zeros(1280, 0.f);
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(lw, ones.data(), 1280 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(lb, zeros.data(), 1280 * 4, hipMemcpyHostToDevice));
  }
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto aggressor = [&](int n) {
    for (int k = 0; k < n; ++k) {
      if (haff_layernorm(x, 1280, hbuf, 1280, lw, lb, nullptr, M, 1280, 1e-6f, 0, s2)) return 2;
      if (haff_gemm_bf16(hbuf, 1280, w, 1280, q, 3840, nullptr, nullptr, 0, nullptr, M, 3840, 1280, 0, 0, 0, s2)) return 2;
    }
    return 0;
  };

  // ---- test A ----
  for (int with_other = 0; with_other < 2; ++with_other) {
    auto runA = [&](const char* name, auto kern, int grid) -> int {
      CK(hipMemset(err, 0, 4)); CK(hipMemset(info, 0, 16)); CK(hipDeviceSynchronize());
      if (with_other && aggressor(60)) return 2;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s1, cold, cold_chunks, hot, 4000, err, info);
      CK(hipDeviceSynchronize());
      char buf[160];
      snprintf(buf, sizeof buf, "A %-40s grid %4d %s", name, grid, with_other ? "beside aggressor" : "alone");
      return report(buf, err, info);
    };
    for (int grid : {256, 2048}) {
      if (runA("cold older, 0 newer (vmcnt(0) control)", order_same_wave<0, false>, grid)) return 1;
      if (runA("cold older, 1 hot newer, vmcnt(1)", order_same_wave<1, false>, grid)) return 1;
      if (runA("cold older, 4 hot newer, vmcnt(4)", order_same_wave<4, false>, grid)) return 1;
      if (runA("cold older, 8 hot newer, vmcnt(8)", order_same_wave<8, false>, grid)) return 1;
      if (runA("cold older, 4 cold newer, vmcnt(4)", order_same_wave<4, true>, grid)) return 1;
      if (runA("cold older, 8 cold newer, vmcnt(8)", order_same_wave<8, true>, grid)) return 1;
    }
  }

  // ---- test B ----
  const int n_stages = 64, blocks = 256;
  const size_t src_bytes = (size_t)blocks * 256 * n_stages * 128;
  CK(hipMalloc(&src, src_bytes));
  {
    std::vector<unsigned> h(src_bytes / 4);
    for (size_t r = 0; r < (size_t)blocks * 256; ++r)
      for (int t = 0; t < n_stages; ++t)
        for (int i = 0; i < 32; ++i) h[(r * n_stages + t) * 32 + i] = t + 1;
    CK(hipMemcpy(src, h.data(), src_bytes, hipMemcpyHostToDevice));
  }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ring_loop<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ring_loop<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int with_other = 0; with_other < 2; ++with_other)
    for (int counted = 0; counted < 2; ++counted)
      for (int grid : {48, 256}) {
        CK(hipMemset(err, 0, 4)); CK(hipMemset(info, 0, 16)); CK(hipDeviceSynchronize());
        if (with_other && aggressor(100)) return 2;
        for (int rep = 0; rep < 300; ++rep) {
          if (counted) hipLaunchKernelGGL(ring_loop<true>, dim3(grid), dim3(512), 3 * 32768, s1, src, n_stages, err, info);
          else hipLaunchKernelGGL(ring_loop<false>, dim3(grid), dim3(512), 3 * 32768, s1, src, n_stages, err, info);
        }
        CK(hipDeviceSynchronize());
        char buf[160];
        snprintf(buf, sizeof buf, "B ring 3 x 32 KiB, %s, grid %3d x 300 launches, %s", counted ? "vmcnt(4) across barrier" : "vmcnt(0) control      ",
                 grid, with_other ? "beside aggressor" : "alone");
        report(buf, err, info);
      }
  return 0;
}
