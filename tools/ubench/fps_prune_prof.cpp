// Per-phase cycle accounting of the box-pruned FPS round, every wave of scene 0.  Build (on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -DDEMF_FPS_PROFILE tools/ubench/fps_prune_prof.cpp -o /tmp/fps_prune_prof
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
  printf("N=%d M=%d: %.3f ms (sort + chain), %.0f ns/round\n", N, M, ms, ms * 1e6 / (M - 1));
  printf("wave: cycles/round [test+update+search | publish+barrier | reduce]  rounds with update / with re-search, pairs updated per round | cycles per SEARCH, cycles of test+update per round WITH update\n");
  for (int w = 0; w < 16; ++w)
    printf("%2d: %6.0f %6.0f %6.0f   %5.3f %5.3f %5.2f\n", w, prof[w][0] / double(M - 1), prof[w][1] / double(M - 1),
           prof[w][2] / double(M - 1), prof[w][3] / double(M - 1), prof[w][4] / double(M - 1), prof[w][5] / double(M - 1));
  for (int w = 0; w < 16; ++w)
    printf("%2d: search %6.0f cycles each (%lld), test+update %6.0f cycles each (%lld)\n", w,
           prof[w][6] / double(prof[w][4] ? prof[w][4] : 1), prof[w][4], prof[w][7] / double(prof[w][3] ? prof[w][3] : 1), prof[w][3]);
}
