// Per-phase cycle accounting of fps_pair_kernel (per-pair cached maxima), every wave of scene 0.  Build (on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -DDEMF_FPS_PROFILE tools/ubench/fps_pair_prof.cpp -o /tmp/fps_pair_prof
#include "../../demf_amd/csrc/fps.hip"
#include "../../demf_amd/csrc/capi.hip"
#include <vector>
#include <random>
int main() {
  const int B = 8, N = 20000, M = 2048;
  std::vector<float> h(B * N * 3); std::mt19937 g(1);
  std::uniform_real_distribution<float> ux(-3, 3), uz(0, 3);
  for (int i = 0; i < B * N; ++i) { h[3 * i] = ux(g); h[3 * i + 1] = ux(g); h[3 * i + 2] = uz(g); }
  float *d, *temp; int* idx;
  hipMalloc(&d, h.size() * 4); hipMalloc(&idx, B * M * 4); hipMalloc(&temp, (size_t)B * N * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  demf_fps_ws_f32(B, N, M, d, temp, (long long)B * N, idx, nullptr); hipDeviceSynchronize();
  hipEventRecord(e0); demf_fps_ws_f32(B, N, M, d, temp, (long long)B * N, idx, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long prof[16][8]; hipMemcpyFromSymbol(prof, HIP_SYMBOL(demf::g_prune_prof), sizeof(prof));
  const double R = M - 1;
  printf("N=%d M=%d: %.3f ms (sort + chain), %.0f cycles/round at 2.4 GHz\n", N, M, ms, ms * 1e-3 / R * 2.4e9);
  printf("wave: cycles/round [test+update+pairs | publish+barrier | reduce]  rounds with update / with a lost pair best; pairs updated / re-searched per round | cycles per round WITH update\n");
  for (int w = 0; w < 16; ++w)
    printf("%2d: %6.0f %6.0f %6.0f   %5.3f %5.3f  %5.2f %5.2f | %6.0f\n", w, prof[w][0] / R, prof[w][1] / R, prof[w][2] / R,
           prof[w][3] / R, prof[w][4] / R, prof[w][5] / R, prof[w][6] / R, prof[w][7] / double(prof[w][3] ? prof[w][3] : 1));
}
