// What can ONE CU stream of a [N][K] bf16 weight matrix against <= 16 activation rows, and in which form?
// Two persistent forms of the decode-sized product (M <= 16; out[w][m][n] = partial sum of K-quarter w, the caller adds the four):
//   reg<D>   weights and activations straight into a ring of D register slots (one 32-deep k-step each), every load of the
//            loop issued unconditionally (past the end: one shared dummy line) so hipcc counts its waits (with a conditional
//            load in the loop it drains vmcnt(0) in front of every multiply); 4 waves split K; NTL: non-temporal weight loads
//   dma<S>   weights by LDS-DMA into a wave-private ring of S 1-KB slots (a lane reads back the 16 bytes it asked for: its
//            MFMA fragment), counted vmcnt; the wave's activation K-quarter lives in registers (K <= 4096)
// A workgroup walks the 16-row weight tiles blockIdx.x, + gridDim.x, ...; the request stream runs on across tile boundaries.
// Timed by tools/stream_probe.py on k of the 256 CUs (the others held by tools/probes/cu_blocker.hip).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o stream_probe.so stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

typedef unsigned short bf16_t;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(256))) unsigned int probe_zero_page[64];

template <int D, bool NTL = false>
__global__ __launch_bounds__(256) void probe_reg(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X, float* __restrict__ out, int N, int K, int M) {
  __shared__ unsigned int claim[2048];   // 8 KB of LDS: the blocker's CUs (2 KB left) cannot take this kernel either
  claim[threadIdx.x] = threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fh = lane >> 4;
  const int kq = K / 4, steps = kq / 32, k_lo = wave * kq;
  const int tiles = N / 16;
  const bf16_t* xp = X + (long)min(fr, M - 1) * K + k_lo + fh * 8;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(probe_zero_page);
  uint4 wv[D], xv[D];
  int it_tile = blockIdx.x, it_ks = 0;
  auto issue = [&](int slot) {
    const bool live = it_tile < tiles;
    const bf16_t* wp = live ? W + (long)(it_tile * 16 + fr) * K + k_lo + it_ks * 32 + fh * 8 : zero;
    const bf16_t* xq = live ? xp + it_ks * 32 : zero;
    if constexpr (NTL) {
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(wp));
      wv[slot] = uint4{t[0], t[1], t[2], t[3]};
    } else {
      wv[slot] = *reinterpret_cast<const uint4*>(wp);
    }
    xv[slot] = *reinterpret_cast<const uint4*>(xq);
    if (++it_ks == steps) { it_ks = 0; it_tile += gridDim.x; }
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) issue(s);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int tile = blockIdx.x, ks = 0;
  const int n_my = tile < tiles ? (tiles - tile + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int total = n_my * steps;
  for (int base = 0; base < total; base += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      issue((s + D - 1) % D);               // unconditional (past the end: the dummy line), so the waits can be counted
      __builtin_amdgcn_sched_barrier(0);
      if (base + s < total) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[s]), __builtin_bit_cast(bf16x8, xv[s]), acc, 0, 0, 0);
        if (++ks == steps) {
          // lane holds D[n = 4 fh + r][m = fr]
          float* o = out + ((long)wave * 16 + fr) * N + tile * 16 + 4 * fh;
          *reinterpret_cast<float4*>(o) = float4{acc[0], acc[1], acc[2], acc[3]};
          acc = f32x4{0.f, 0.f, 0.f, 0.f};
          ks = 0;
          tile += gridDim.x;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (claim[(threadIdx.x + 1) & 255] == 0xdeadbeefu) out[0] = 1.f;
}

template <int S, int XS, int AUX = 0>
__global__ __launch_bounds__(256) void probe_dma(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X, float* __restrict__ out, int N, int K, int M) {
  __shared__ __attribute__((aligned(16))) char ring[4][S][1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 15, fh = lane >> 4;
  const int kq = K / 4, steps = kq / 32, k_lo = wave * kq;   // steps <= XS
  const int tiles = N / 16;
  const bf16_t* xp = X + (long)min(fr, M - 1) * K + k_lo + fh * 8;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(probe_zero_page);
  uint4 xv[XS];
#pragma unroll
  for (int i = 0; i < XS; ++i) xv[i] = *reinterpret_cast<const uint4*>(xp + min(i, steps - 1) * 32);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no ordinary load is outstanding while the ring runs
  int it_tile = blockIdx.x, it_ks = 0;
  auto issue = [&](int slot) {
    const bool live = it_tile < tiles;
    const bf16_t* wp = live ? W + (long)(it_tile * 16 + fr) * K + k_lo + it_ks * 32 + fh * 8 : zero;
    __builtin_amdgcn_global_load_lds((gptr_t)wp, (lptr_t)&ring[wave][slot][0], 16, 0, AUX);
    if (++it_ks == steps) { it_ks = 0; it_tile += gridDim.x; }
  };
#pragma unroll
  for (int s = 0; s < S - 1; ++s) issue(s);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int tile = blockIdx.x;
  while (tile < tiles) {
    // one tile = `steps` k-steps; the ring position advances by steps % S per tile, so the slot of k-step i is (base + i) % S
    // with base carried along: keep S | steps (K = 4096: 32 steps, S = 16) so that slots are compile-time inside the tile
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      if (i < steps) {
        constexpr int dummy = 0;
        (void)dummy;
        const int slot = i % S;
        issue((i + S - 1) % S);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S - 1) : "memory");
        uint4 wfrag;   // by hand: behind an LDS-DMA hipcc drains vmcnt(0) in front of every ds_read it emits itself
        const unsigned laddr = (unsigned)(uintptr_t)(&ring[wave][slot][0]) + lane * 16;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(wfrag) : "v"(laddr) : "memory");
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, xv[i]), acc, 0, 0, 0);
      }
    }
    float* o = out + ((long)wave * 16 + fr) * N + tile * 16 + 4 * fh;
    *reinterpret_cast<float4*>(o) = float4{acc[0], acc[1], acc[2], acc[3]};
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    tile += gridDim.x;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy fills must have landed before the LDS is handed on
}

extern "C" int stream_probe_launch(int variant, const void* W, const void* X, float* out, int N, int K, int M, int grid, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bf16_t* w = static_cast<const bf16_t*>(W);
  const bf16_t* x = static_cast<const bf16_t*>(X);
  if ((K % 128) || (N % 16) || M < 1 || M > 16) return -1;
  switch (variant) {
    case 0: hipLaunchKernelGGL(probe_reg<8>, dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 1: hipLaunchKernelGGL(probe_reg<16>, dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 2: hipLaunchKernelGGL(probe_reg<24>, dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 3: if (K / 128 > 32 || (K / 128) % 8) return -1; hipLaunchKernelGGL((probe_dma<8, 32>), dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 4: if (K / 128 > 32 || (K / 128) % 16) return -1; hipLaunchKernelGGL((probe_dma<16, 32>), dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 5: if (K / 128 > 32 || (K / 128) % 32) return -1; hipLaunchKernelGGL((probe_dma<32, 32>), dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 6: hipLaunchKernelGGL((probe_reg<8, true>), dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 7: hipLaunchKernelGGL((probe_reg<16, true>), dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 8: hipLaunchKernelGGL((probe_dma<16, 32, 1>), dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 9: hipLaunchKernelGGL((probe_dma<16, 32, 2>), dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    case 10: hipLaunchKernelGGL((probe_dma<16, 32, 3>), dim3(grid), dim3(256), 0, s, w, x, out, N, K, M); break;
    default: return -2;
  }
  return (int)hipGetLastError();
}
