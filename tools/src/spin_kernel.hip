// Measurement helper (tools/queue_residency.py): a workgroup that stays resident for a given
// number of microseconds, with a small or a CU-filling register footprint.
#include <hip/hip_runtime.h>
template <bool FAT>
__global__ __launch_bounds__(1024) void spin_k(long long ticks, int* out) {
  if constexpr (FAT) asm volatile("" ::: "v127");   // 128 VGPRs x 16 waves = the CU's register file
  const long long t0 = wall_clock64();
  int n = 0;
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(32); ++n; }
  if (threadIdx.x == 0 && out) out[blockIdx.x] = n;
}
extern "C" int spin_launch(int blocks, int threads, int fat, long long ticks, int* out, void* stream) {
  if (fat) hipLaunchKernelGGL(spin_k<true>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, ticks, out);
  else hipLaunchKernelGGL(spin_k<false>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, ticks, out);
  return (int)hipGetLastError();
}
