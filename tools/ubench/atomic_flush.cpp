// Cost of the "every workgroup adds its partial result into one shared array" flush, as the fused backward kernels end
// (dW: 256 x 128 floats from each of 256 workgroups) and as group_first_bwd_k ended before round 6 (3 x C1 floats,
// dw_ld apart).  Build (on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/ubench/atomic_flush.cpp -o /tmp/atomic_flush
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// mode 0: contiguous, all workgroups in the same order; 1: contiguous, workgroup w starts w/gridDim of the way in;
// 2: elements `stride` floats apart (one cache line per lane); 3: plain stores to a private slice (the floor)
template <typename T>
__global__ __launch_bounds__(512) void flush_k(T* dst, int n, int mode, int stride, T* priv) {
  const int w = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (mode == 0) atomicAdd(dst + i, (T)1);
    else if (mode == 1) { int j = i + (int)((long long)w * n / gridDim.x); if (j >= n) j -= n; atomicAdd(dst + j, (T)1); }
    else if (mode == 2) atomicAdd(dst + (size_t)i * stride, (T)1);
    else priv[(size_t)w * n + i] = (T)1;
  }
}

template <typename T>
static void run(const char* name, int grid, int n, int mode, int stride) {
  T *d, *priv;
  hipMalloc(&d, (size_t)n * (mode == 2 ? stride : 1) * sizeof(T) + 256);
  hipMalloc(&priv, (size_t)grid * n * sizeof(T));
  hipMemset(d, 0, (size_t)n * (mode == 2 ? stride : 1) * sizeof(T));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(flush_k<T>, dim3(grid), dim3(512), 0, 0, d, n, mode, stride, priv);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(flush_k<T>, dim3(grid), dim3(512), 0, 0, d, n, mode, stride, priv);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%-34s grid %4d n %6d: %8.1f us  (%.2f G lane-adds/s)\n", name, grid, n, us, (double)grid * n / us * 1e-3);
  hipFree(d); hipFree(priv);
}

int main() {
  for (int grid : {256, 1024}) {
    for (int n : {384, 16384, 32768}) {
      run<float>("f32 contiguous same order", grid, n, 0, 0);
      run<float>("f32 contiguous rotated start", grid, n, 1, 0);
      if (n <= 16384) run<float>("f32 stride 131 floats", grid, n, 2, 131);
      run<double>("f64 contiguous same order", grid, n, 0, 0);
      run<float>("f32 plain stores (private)", grid, n, 3, 0);
    }
  }
  return 0;
}
