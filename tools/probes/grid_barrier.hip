// Cost of a software grid barrier on MI355X (8 XCDs): NWG workgroups x 256 threads run `iters` barriers; between barriers
// each workgroup writes a few values and reads its neighbour's from the previous round (checks cross-XCD visibility).
// Variants: (0) __threadfence() release/acquire + device-scope atomics (what cooperative-groups grid.sync does);
//           (1) relaxed atomics only, payload written / read with agent-scope atomic stores / loads (no L2 write-back).
// Every spin has a timeout: a barrier that cannot complete sets an error flag and the kernel drains.
// hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>

struct Bar { unsigned count; unsigned gen; unsigned err; unsigned pad; };

template <int MODE>
__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned nwg, unsigned& my_gen) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (MODE == 0) __threadfence();
    const unsigned old = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nwg - 1) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&b->gen, 1u, MODE == 0 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      long spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000L) { __hip_atomic_store(&b->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
      }
    }
    if (MODE == 0) __threadfence();
  }
  my_gen++;
  __syncthreads();
  return ok;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(Bar* b, unsigned* payload, int iters, unsigned* bad) {
  const unsigned nwg = gridDim.x;
  unsigned my_gen = 0;
  if (threadIdx.x == 0) my_gen = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  my_gen = __shfl(my_gen, 0);   // wave 0 only; other waves never use it
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    // every thread writes one word of this workgroup's 1 KiB slot
    const unsigned v = (unsigned)it * 2654435761u + blockIdx.x * 256 + threadIdx.x;
    if (MODE == 0) payload[blockIdx.x * 256 + threadIdx.x] = v;
    else __hip_atomic_store(&payload[blockIdx.x * 256 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!grid_barrier<MODE>(b, nwg, my_gen)) return;
    const unsigned nb = (blockIdx.x + 37) % nwg;   // a workgroup on another XCD
    unsigned got;
    if (MODE == 0) got = payload[nb * 256 + threadIdx.x];
    else got = __hip_atomic_load(&payload[nb * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (got != (unsigned)it * 2654435761u + nb * 256 + threadIdx.x) wrong++;
    if (!grid_barrier<MODE>(b, nwg, my_gen)) return;   // nobody overwrites a slot before it was read
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int MODE>
static void run(int nwg, int iters) {
  Bar* b; unsigned* payload; unsigned* bad;
  hipMalloc(&b, sizeof(Bar)); hipMemset(b, 0, sizeof(Bar));
  hipMalloc(&payload, nwg * 1024); hipMemset(payload, 0, nwg * 1024);
  hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 0, 0, b, payload, 10, bad);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 0, 0, b, payload, iters, bad);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  Bar hb; unsigned hbad;
  hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost);
  hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
  printf("mode %d  %4d workgroups: %.2f us per barrier (%d barriers), timeout flag %u, stale reads %u\n", MODE, nwg,
         ms * 1e3 / (2.0 * iters), 2 * iters, hb.err, hbad);
  hipFree(b); hipFree(payload); hipFree(bad);
}

int main() {
  for (int nwg : {64, 256, 512}) {
    run<0>(nwg, 2000);
    run<1>(nwg, 2000);
  }
  return 0;
}
