// Ball query for gfx950.
//
// Replaces mmdet3d.ops.ball_query as used by QueryAndGroup inside the
// PointSAModules built at demf/modeling/heads/class_agnostic_vote_head.py:383
// and by the PointNet2SASSG backbone (configs/demf/demf_votenet.py:48-62).
//
// Upstream runs one thread per centre that walks the N candidates serially.
// Here one wave owns CW centres: the 64 lanes each hold one candidate point in
// registers (a coalesced 768-byte read per chunk), test it against the CW
// centres held in SGPRs, and append hits with a ballot + prefix popcount so the
// "first nsample hits in index order" contract is kept while testing 64
// candidates per step.  A wave leaves the scan as soon as all its centres are
// full.  Candidate traffic is N*12 bytes per wave from L2; only idx is written.
#include "common.h"

namespace demf {

template <int CW, bool MIN0>
__global__ __launch_bounds__(256) void ball_query_kernel(
    int N, int M, float min_r2, float max_r2, int ns, const float* __restrict__ center,
    const float* __restrict__ xyz, int* __restrict__ idx) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave_in_grid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int m0 = wave_in_grid * CW;
  if (m0 >= M) return;
  xyz += (size_t)b * N * 3;
  center += (size_t)b * M * 3;
  idx += (size_t)b * M * ns;

  float cx[CW], cy[CW], cz[CW];
  int cnt[CW], first[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const int m = min(m0 + c, M - 1);
    cx[c] = center[3 * m + 0];
    cy[c] = center[3 * m + 1];
    cz[c] = center[3 * m + 2];
    cnt[c] = (m0 + c < M) ? 0 : ns;  // out-of-range centres are born full
    first[c] = 0;
  }
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  for (int k0 = 0; k0 < N; k0 += 64) {
    bool all_full = true;
#pragma unroll
    for (int c = 0; c < CW; ++c) all_full = all_full && (cnt[c] >= ns);
    if (all_full) break;
    const int k = k0 + lane;
    const bool ok = k < N;
    const float x = ok ? xyz[3 * k + 0] : 3.0e38f;  // pad lanes fail every test
    const float y = ok ? xyz[3 * k + 1] : 3.0e38f;
    const float z = ok ? xyz[3 * k + 2] : 3.0e38f;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const float d2 = dist2(cx[c] - x, cy[c] - y, cz[c] - z);
      bool hit;
      if constexpr (MIN0)
        hit = d2 < max_r2;  // min_r == 0: d2 >= 0 always, d2 == 0 implies d2 < max_r2
      else
        hit = (d2 == 0.f) || (d2 >= min_r2 && d2 < max_r2);
      hit = hit && ok;
      const unsigned long long mask = __ballot(hit);
      if (mask != 0ull && cnt[c] < ns) {
        if (cnt[c] == 0) first[c] = k0 + __builtin_ctzll(mask);
        const int pos = cnt[c] + __builtin_popcountll(mask & lt_mask);
        if (hit && pos < ns) idx[(size_t)(m0 + c) * ns + pos] = k;
        cnt[c] += __builtin_popcountll(mask);
      }
    }
  }
  // slots never reached keep the first hit (or 0 when the ball is empty)
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    if (m0 + c < M) {
      const int filled = min(cnt[c], ns);
      for (int s = filled + lane; s < ns; s += 64)
        idx[(size_t)(m0 + c) * ns + s] = first[c];
    }
  }
}

}  // namespace demf

using namespace demf;

extern "C" int demf_ball_query_f32(int B, int N, int M, float min_radius, float max_radius,
                                   int nsample, const float* center_xyz, const float* xyz,
                                   int* idx, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && M >= 0 && nsample >= 1,
               "ball_query: bad sizes B=%d N=%d M=%d ns=%d", B, N, M, nsample);
  if (B == 0 || M == 0) return DEMF_OK;
  DEMF_REQUIRE(center_xyz && xyz && idx, "ball_query: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const float min_r2 = min_radius * min_radius;
  const float max_r2 = max_radius * max_radius;
  constexpr int CW = 4;
  const int waves = cdiv(M, CW);
  dim3 grid(cdiv(waves, 4), B), block(256);
  if (min_radius == 0.f && max_r2 > 0.f)
    hipLaunchKernelGGL((ball_query_kernel<CW, true>), grid, block, 0, s, N, M, min_r2,
                       max_r2, nsample, center_xyz, xyz, idx);
  else
    hipLaunchKernelGGL((ball_query_kernel<CW, false>), grid, block, 0, s, N, M, min_r2,
                       max_r2, nsample, center_xyz, xyz, idx);
  return check_launch("ball_query");
}
