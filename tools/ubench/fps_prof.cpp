// Per-phase cycle accounting of the FPS round (wave 0 of scene 0). Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -DDEMF_FPS_PROFILE tools/ubench/fps_prof.cpp -o tools/ubench/fps_prof
#include "../../demf_amd/csrc/fps.hip"
#include "../../demf_amd/csrc/capi.hip"
#include <vector>
#include <random>
int main() {
  const int B = 8;
  for (auto nm : {std::pair<int,int>{20000, 2048}, {2048, 1024}, {1024, 512}}) {
    int N = nm.first, M = nm.second;
    std::vector<float> h(B * N * 3); std::mt19937 g(1); std::uniform_real_distribution<float> u(-3, 3);
    for (auto& v : h) v = u(g);
    float* d; int* idx; hipMalloc(&d, h.size() * 4); hipMalloc(&idx, B * M * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    demf_fps_f32(B, N, M, d, nullptr, idx, nullptr); hipDeviceSynchronize();
    hipEventRecord(e0); demf_fps_f32(B, N, M, d, nullptr, idx, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long prof[8]; hipMemcpyFromSymbol(prof, HIP_SYMBOL(demf::g_fps_prof), sizeof(prof));
    printf("N=%d M=%d: %.3f ms, %.0f ns/round | cycles/round: compute %.0f wave-max %.0f barrier1+pick %.0f search %.0f barrier2+bcast %.0f\n", N, M, ms,
           ms * 1e6 / (M - 1), prof[0] / double(M - 1), prof[1] / double(M - 1), prof[2] / double(M - 1), prof[3] / double(M - 1), prof[4] / double(M - 1));
  }
}
