// VALU issue-rate / clock calibration for gfx950: how many cycles does one wave64 fp32 VALU
// instruction take with W waves per SIMD?  (s_memtime ticks = shader cycles.)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1, e = a + 2, f = a + 3, g = a + 4, h = a + 5;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 A = {a, d}, B = {b, b}, C = {c, c}, D = {e, f}, E = {g, h}, F = {a, g};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent scalar fma
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c); e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c);
        g = __builtin_fmaf(g, b, c); h = __builtin_fmaf(h, b, c); A[0] = __builtin_fmaf(A[0], b, c); A[1] = __builtin_fmaf(A[1], b, c);
      }
    } else if (MODE == 1) {  // 4 independent packed fma (8 flops-lanes)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        A = __builtin_elementwise_fma(A, B, C); D = __builtin_elementwise_fma(D, B, C);
        E = __builtin_elementwise_fma(E, B, C); F = __builtin_elementwise_fma(F, B, C);
      }
    } else {  // min/max mix
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(d) : "v"(c));
        asm volatile("v_min_f32 %0, %0, %1" : "+v"(e) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(f) : "v"(c));
        asm volatile("v_min_f32 %0, %0, %1" : "+v"(g) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(h) : "v"(c));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(A[0]) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(A[1]) : "v"(b));
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h + A[0] + A[1] + D[0] + D[1] + E[0] + E[1] + F[0] + F[1];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int threads, int ninstr_per_iter) {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<8, threads>>>(out, cyc, 100);
  hipEventRecord(e0); k<MODE><<<8, threads>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double per = (double)c / iters / ninstr_per_iter;
  printf("%-10s threads %4d (waves/SIMD %d): %.2f counter-ticks per wave-instr (per SIMD: %.2f), wall %.3f ms, ticks/us %.1f\n",
         name, threads, threads / 256, per, per / (threads / 256.0), ms, c / (ms * 1e3));
}
int main() {
  for (int th : {256, 512, 1024}) { run<0>("fma", th, 32); run<1>("pk_fma", th, 16); run<2>("minmax", th, 32); }
  return 0;
}
