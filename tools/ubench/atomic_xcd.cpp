// Are workgroup-scope atomics (performed in the XCD's own L2) on per-XCD copies of an accumulator faster than
// agent-scope ones (performed at the memory side) - and do the copies add up?  The copy is chosen by the XCC_ID
// hardware register, not by the workgroup index.  Build (on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/ubench/atomic_xcd.cpp -o /tmp/atomic_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// mode 0: agent-scope adds to one array; 1: workgroup-scope adds to copy[xcc]; 2: agent-scope adds to copy[xcc]
__global__ __launch_bounds__(512) void flush_k(float* dst, int n, int mode, unsigned* xcc_hist) {
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(xcc_hist + (blockIdx.x % 8) * 16 + x, 1u);
  float* d = mode == 0 ? dst : dst + (size_t)x * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (mode == 1) __hip_atomic_fetch_add(d + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(d + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int grid = 256, reps = 20;
  for (int n : {384, 16384, 32768}) {
    for (int mode = 0; mode < 3; ++mode) {
      float* d; unsigned* h;
      hipMalloc(&d, (size_t)16 * n * 4); hipMalloc(&h, 8 * 16 * 4);
      hipMemset(d, 0, (size_t)16 * n * 4); hipMemset(h, 0, 8 * 16 * 4);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(flush_k, dim3(grid), dim3(512), 0, 0, d, n, mode, h);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(flush_k, dim3(grid), dim3(512), 0, 0, d, n, mode, h);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<float> host((size_t)16 * n);
      hipMemcpy(host.data(), d, host.size() * 4, hipMemcpyDeviceToHost);
      double tot = 0, mn = 1e30, mx = -1e30;
      for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int c = 0; c < 16; ++c) s += host[(size_t)c * n + i];
        tot += s; mn = s < mn ? s : mn; mx = s > mx ? s : mx;
      }
      std::vector<unsigned> hh(128);
      hipMemcpy(hh.data(), h, 512, hipMemcpyDeviceToHost);
      printf("n %6d mode %d (%s): %7.1f us/launch; per-element total over copies min %.0f max %.0f (expected %d)\n", n, mode,
             mode == 0 ? "agent scope, one array" : mode == 1 ? "workgroup scope, copy[XCC_ID]" : "agent scope, copy[XCC_ID]",
             ms * 1e3 / reps, mn, mx, grid * (reps + 1));
      if (n == 384 && mode == 0) {
        printf("  XCC_ID by blockIdx %% 8 (rows) x XCC_ID (cols):\n");
        for (int r = 0; r < 8; ++r) { printf("   "); for (int c = 0; c < 8; ++c) printf(" %4u", hh[r * 16 + c]); printf("\n"); }
      }
      hipFree(d); hipFree(h);
    }
  }
  return 0;
}
