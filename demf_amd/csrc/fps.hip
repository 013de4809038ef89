// Furthest point sampling for gfx950.
//
// Replaces mmdet3d.ops.furthest_point_sample as called at
// demf/modeling/heads/class_agnostic_vote_head.py:429-430 and inside every
// PointSAModule (configs/demf/demf_votenet.py:48-62,155-162).
//
// FPS is a chain of M-1 dependent arg-max rounds: it is bound by the latency of
// one round, not by HBM.  Design for CDNA4:
//   * one workgroup per scene, the scene's points and their running min-distance
//     live in VGPRs for the whole kernel (20 000 points = 20 per lane at 1024
//     threads) - global memory is touched once on entry and once per selected
//     index on exit;
//   * the per-round arg-max is a value-only v_max chain (8 VALU per point), the
//     index is recovered afterwards from ballots of (temp == max) on the scalar
//     unit, so no per-point select of the index is paid;
//   * one LDS exchange and ONE barrier per round (slots double-buffered by round
//     parity); every wave redundantly reduces the 16 wave slots with DPP row
//     operations and broadcasts the winner's coordinates with v_readlane, so the
//     next round starts from SGPRs.
//   * thread t owns points t, t+BS, t+2BS, ... with BS = min(1024, 2^floor(log2 N))
//     - the same ownership as the upstream block reduction - so "first maximum in
//     the thread, then lowest thread id" reproduces upstream tie-breaking exactly.
#include "common.h"

namespace demf {

struct __attribute__((aligned(32))) FpsSlot {
  float v;
  int i;
  float x, y, z;
  int pad[3];
};

#define FPS_CASE(c)                         \
  case c:                                   \
    if constexpr (c < PPT) {                \
      wx = readlane_f(px[c], wl);           \
      wy = readlane_f(py[c], wl);           \
      wz = readlane_f(pz[c], wl);           \
    }                                       \
    break;

template <int BS, int PPT>
__global__ __launch_bounds__(BS) void fps_reg_kernel(int N, int M,
                                                     const float* __restrict__ xyz,
                                                     int* __restrict__ idx) {
  constexpr int NW = BS / 64;
  __shared__ FpsSlot slots[2][16];
  const int b = blockIdx.x;
  xyz += (size_t)b * N * 3;
  idx += (size_t)b * M;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  float px[PPT], py[PPT], pz[PPT], tmp[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int k = tid + p * BS;
    const bool ok = k < N;
    px[p] = ok ? xyz[3 * k + 0] : 0.f;
    py[p] = ok ? xyz[3 * k + 1] : 0.f;
    pz[p] = ok ? xyz[3 * k + 2] : 0.f;
    tmp[p] = ok ? 1e10f : -2.f;  // pad slots can never reach the maximum
  }
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (tid == 0) idx[0] = 0;

  for (int j = 1; j < M; ++j) {
    float best = -1.f;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const float d = dist2(px[p] - x1, py[p] - y1, pz[p] - z1);
      const float t = fminf(d, tmp[p]);
      tmp[p] = t;
      best = fmaxf(best, t);
    }
    const float vmax = wave_allmax(best);
    // lowest lane holding the maximum, and that lane's lowest slot
    int wl = 64, wp = 0;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const unsigned long long m = __ballot(tmp[p] == vmax);
      const int l = m ? __builtin_ctzll(m) : 64;
      if (l < wl) {
        wl = l;
        wp = p;
      }
    }
    wl = __builtin_amdgcn_readfirstlane(wl);
    wp = __builtin_amdgcn_readfirstlane(wp);
    float wx = 0.f, wy = 0.f, wz = 0.f;
    switch (wp) {
      FPS_CASE(0) FPS_CASE(1) FPS_CASE(2) FPS_CASE(3) FPS_CASE(4) FPS_CASE(5)
      FPS_CASE(6) FPS_CASE(7) FPS_CASE(8) FPS_CASE(9) FPS_CASE(10) FPS_CASE(11)
      FPS_CASE(12) FPS_CASE(13) FPS_CASE(14) FPS_CASE(15) FPS_CASE(16) FPS_CASE(17)
      FPS_CASE(18) FPS_CASE(19) FPS_CASE(20) FPS_CASE(21) FPS_CASE(22) FPS_CASE(23)
      default: break;
    }
    const int wi = wave * 64 + wl + wp * BS;
    int old;
    if constexpr (NW == 1) {
      old = wi;
      x1 = wx;
      y1 = wy;
      z1 = wz;
    } else {
      const int par = j & 1;
      if (lane == 0) {
        FpsSlot s;
        s.v = vmax;
        s.i = wi;
        s.x = wx;
        s.y = wy;
        s.z = wz;
        slots[par][wave] = s;
      }
      __syncthreads();
      const FpsSlot s = slots[par][lane & (NW - 1)];
      const float rmax = row16_allmax(s.v);
      const unsigned long long m = __ballot(s.v == rmax);
      const int w = __builtin_ctzll(m);  // lowest wave holding the maximum
      old = readlane_i(s.i, w);
      x1 = readlane_f(s.x, w);
      y1 = readlane_f(s.y, w);
      z1 = readlane_f(s.z, w);
    }
    if (tid == 0) idx[j] = old;
  }
}

// Fallback for N beyond the register budget (or N < 64): the running distance
// lives in the caller's `temp` scratch, ownership/tie rule identical.
// BSREF = upstream block size (power of two, may be < 64); the launch uses
// max(64, BSREF) threads and the surplus lanes idle.
__global__ __launch_bounds__(1024) void fps_generic_kernel(int N, int M, int BSREF,
                                                           const float* __restrict__ xyz,
                                                           float* __restrict__ temp,
                                                           int* __restrict__ idx) {
  __shared__ FpsSlot slots[2][16];
  const int b = blockIdx.x;
  xyz += (size_t)b * N * 3;
  temp += (size_t)b * N;
  idx += (size_t)b * M;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int NW = (blockDim.x + 63) / 64;
  const bool active = tid < BSREF;
  if (active)
    for (int k = tid; k < N; k += BSREF) temp[k] = 1e10f;
  int old = 0;
  if (tid == 0) idx[0] = 0;
  for (int j = 1; j < M; ++j) {
    const float x1 = xyz[3 * old], y1 = xyz[3 * old + 1], z1 = xyz[3 * old + 2];
    float best = -1.f;
    int besti = 0;
    if (active) {
      for (int k = tid; k < N; k += BSREF) {
        const float d = dist2(xyz[3 * k] - x1, xyz[3 * k + 1] - y1, xyz[3 * k + 2] - z1);
        const float t = fminf(d, temp[k]);
        temp[k] = t;
        if (t > best) {
          best = t;
          besti = k;
        }
      }
    }
    const float vmax = wave_allmax(best);
    const unsigned long long m = __ballot(best == vmax);
    const int wl = __builtin_ctzll(m);
    const int wi = readlane_i(besti, wl);
    const int par = j & 1;
    if (lane == 0) {
      slots[par][wave].v = vmax;
      slots[par][wave].i = wi;
    }
    __syncthreads();
    const int sl = lane < NW ? lane : 0;
    const float sv = slots[par][sl].v;
    const int si = slots[par][sl].i;
    const float rmax = wave_allmax(sv);
    const unsigned long long m2 = __ballot(sv == rmax);
    old = readlane_i(si, __builtin_ctzll(m2));
    if (tid == 0) idx[j] = old;
  }
}

template <int BS, int PPT>
static void launch_reg(int B, int N, int M, const float* xyz, int* idx, hipStream_t s) {
  hipLaunchKernelGGL((fps_reg_kernel<BS, PPT>), dim3(B), dim3(BS), 0, s, N, M, xyz, idx);
}

}  // namespace demf

using namespace demf;

extern "C" int demf_fps_f32(int B, int N, int M, const float* xyz, float* temp, int* idx,
                            demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && M >= 0, "fps: bad sizes B=%d N=%d M=%d", B, N, M);
  if (B == 0 || M == 0) return DEMF_OK;
  DEMF_REQUIRE(xyz && idx, "fps: null pointer");
  hipStream_t s = (hipStream_t)stream;
  int bs = 1;
  while (bs * 2 <= N && bs < 1024) bs *= 2;  // upstream opt_n_threads()
  const int ppt = cdiv(N, bs);
  bool done = true;
  if (bs == 1024) {
    if (ppt <= 1) launch_reg<1024, 1>(B, N, M, xyz, idx, s);
    else if (ppt <= 2) launch_reg<1024, 2>(B, N, M, xyz, idx, s);
    else if (ppt <= 3) launch_reg<1024, 3>(B, N, M, xyz, idx, s);
    else if (ppt <= 4) launch_reg<1024, 4>(B, N, M, xyz, idx, s);
    else if (ppt <= 6) launch_reg<1024, 6>(B, N, M, xyz, idx, s);
    else if (ppt <= 8) launch_reg<1024, 8>(B, N, M, xyz, idx, s);
    else if (ppt <= 12) launch_reg<1024, 12>(B, N, M, xyz, idx, s);
    else if (ppt <= 16) launch_reg<1024, 16>(B, N, M, xyz, idx, s);
    else if (ppt <= 20) launch_reg<1024, 20>(B, N, M, xyz, idx, s);
    else if (ppt <= 24) launch_reg<1024, 24>(B, N, M, xyz, idx, s);
    else done = false;
  } else if (bs == 512) {
    if (ppt <= 1) launch_reg<512, 1>(B, N, M, xyz, idx, s);
    else launch_reg<512, 2>(B, N, M, xyz, idx, s);
  } else if (bs == 256) {
    if (ppt <= 1) launch_reg<256, 1>(B, N, M, xyz, idx, s);
    else launch_reg<256, 2>(B, N, M, xyz, idx, s);
  } else if (bs == 128) {
    if (ppt <= 1) launch_reg<128, 1>(B, N, M, xyz, idx, s);
    else launch_reg<128, 2>(B, N, M, xyz, idx, s);
  } else if (bs == 64) {
    if (ppt <= 1) launch_reg<64, 1>(B, N, M, xyz, idx, s);
    else launch_reg<64, 2>(B, N, M, xyz, idx, s);
  } else {
    done = false;
  }
  if (!done) {
    DEMF_REQUIRE(temp != nullptr, "fps: N=%d needs the (B,N) temp scratch", N);
    const int threads = bs < 64 ? 64 : bs;
    hipLaunchKernelGGL(fps_generic_kernel, dim3(B), dim3(threads), 0, s, N, M, bs, xyz,
                       temp, idx);
  }
  return check_launch("fps");
}
