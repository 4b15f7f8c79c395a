// Probe: does v_pk_fma_f32 honour op_sel / op_sel_hi on an SGPR-pair source (does the high lane read s[n+1])?
// hipcc emits "v_pk_fma_f32 v[..], v[..], s[50:51], v[..] op_sel:[0,0,1] op_sel_hi:[1,0,1]" with only s50 initialised
// in attn_fwd_kernel<96,4,..> (the splat of a scalar constant): if the hardware ignored op_sel_hi for the scalar
// operand, the high lane would multiply by whatever s51 holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(float k, float garbage, float* out) {
  f32x2 a = {1.0f, 2.0f}, c = {10.0f, 20.0f}, r1, r2, r3;
  asm volatile("s_mov_b32 s50, %3\n\ts_mov_b32 s51, %4\n\ts_nop 4\n\t"
               "v_pk_fma_f32 %0, %1, s[50:51], %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]"
               : "=v"(r1) : "v"(a), "v"(c), "s"(k), "s"(garbage) : "s50", "s51");
  asm volatile("s_mov_b32 s50, %3\n\ts_mov_b32 s51, %4\n\ts_nop 4\n\t"
               "v_pk_fma_f32 %0, %1, s[50:51], %2 op_sel_hi:[1,0,0]"
               : "=v"(r2) : "v"(a), "v"(c), "s"(k), "s"(garbage) : "s50", "s51");
  asm volatile("s_mov_b32 s50, %2\n\ts_mov_b32 s51, %3\n\ts_nop 4\n\t"
               "v_pk_mul_f32 %0, %1, s[50:51] op_sel_hi:[1,0]"
               : "=v"(r3) : "v"(a), "s"(k), "s"(garbage) : "s50", "s51");
  if (threadIdx.x == 0) { out[0] = r1[0]; out[1] = r1[1]; out[2] = r2[0]; out[3] = r2[1]; out[4] = r3[0]; out[5] = r3[1]; }
}
int main() {
  float* d; hipMalloc(&d, 32);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 3.0f, 1000.0f, d);
  float h[6]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("a = (1,2), c = (10,20), s50 = 3, s51 = 1000\n");
  printf("pk_fma op_sel:[0,0,1] op_sel_hi:[1,0,1]: (%g, %g)   expected if op_sel honoured: (23, 26); if hi lane read s51: (23, 2020)\n", h[0], h[1]);
  printf("pk_fma op_sel_hi:[1,0,0]               : (%g, %g)   expected if honoured: (13, 16); if hi lane read s51: (13, 2010)\n", h[2], h[3]);
  printf("pk_mul op_sel_hi:[1,0]                 : (%g, %g)   expected if honoured: (3, 6);   if hi lane read s51: (3, 2000)\n", h[4], h[5]);
  return 0;
}
