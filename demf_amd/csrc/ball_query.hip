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

// ---- large clouds (SA1: 20 000 points, 2 048 centres, r = 0.2): hashed uniform grid -----------------
// The scan above tests every point against every centre (N*M distances: 328 M per batch of 8 scenes, a
// chip-filling ~200 us launch that the training step's own kernels have to share the CUs with).  Here
// the points are first binned into cubic cells of edge 2r*(1+1e-3) (hashed into BQ_T buckets, one
// workgroup per scene, counting sort in LDS); a centre then only meets the points of the 2 x 2 x 2 cells
// its ball can reach - the cell it lies in and, per axis, the neighbour on the side of the nearer face.
// Any point within r of the centre lies in one of those cells (|p - c| < r = (0.5 - 5e-4) cell edges per
// axis, far above the rounding of p/edge), hash collisions only ADD candidates, and every candidate goes
// through the same dist2() predicate as the scan, so the hit SET is identical.  The reference's order -
// the first `ns` hits by point index - is restored by a per-wave bitmap over the point indices in LDS
// (hits set their bit; the set bits are read back in ascending order), which also makes duplicate visits
// of a bucket harmless.
constexpr int BQ_T = 16384;                 // buckets per scene (power of two)
constexpr int BQ_MAXN = 32768;              // bitmap: 1024 words per wave

__device__ __forceinline__ int bq_hash(int ix, int iy, int iz) {
  return (int)(((unsigned)ix * 73856093u) ^ ((unsigned)iy * 19349663u) ^ ((unsigned)iz * 83492791u)) & (BQ_T - 1);
}
__device__ __forceinline__ int bq_cell(float v, float inv) { return (int)floorf(v * inv); }

// one workgroup per scene: start (B, BQ_T + 1) bucket offsets, cells (B, N) = {x, y, z, index}
__global__ __launch_bounds__(1024) void bq_grid_build_k(int N, float inv, const float* __restrict__ xyz,
                                                        int* __restrict__ start, float4* __restrict__ cells) {
  __shared__ int hist[BQ_T];
  __shared__ int wsum[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  xyz += (size_t)b * N * 3;
  start += (size_t)b * (BQ_T + 1);
  cells += (size_t)b * N;
  for (int i = tid; i < BQ_T; i += 1024) hist[i] = 0;
  __syncthreads();
  for (int i = tid; i < N; i += 1024)
    atomicAdd(&hist[bq_hash(bq_cell(xyz[3 * i], inv), bq_cell(xyz[3 * i + 1], inv), bq_cell(xyz[3 * i + 2], inv))], 1);
  __syncthreads();
  // exclusive scan: 16 consecutive buckets per thread, wave scan of the thread sums, then the 16 wave sums
  constexpr int PER = BQ_T / 1024;
  int loc[PER], tsum = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) { loc[j] = hist[tid * PER + j]; tsum += loc[j]; }
  int inc = tsum;
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if ((tid & 63) >= d) inc += o;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = inc;
  __syncthreads();
  int base = inc - tsum;
  for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    hist[tid * PER + j] = base;
    start[tid * PER + j] = base;
    base += loc[j];
  }
  if (tid == 1023) start[BQ_T] = base;
  __syncthreads();
  for (int i = tid; i < N; i += 1024) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const int pos = atomicAdd(&hist[bq_hash(bq_cell(x, inv), bq_cell(y, inv), bq_cell(z, inv))], 1);
    cells[pos] = make_float4(x, y, z, __builtin_bit_cast(float, i));
  }
}

// one wave per centre
__global__ __launch_bounds__(256) void ball_query_grid_k(int N, int M, float inv, float max_r2, int ns,
                                                         const float* __restrict__ center,
                                                         const int* __restrict__ start,
                                                         const float4* __restrict__ cells,
                                                         int* __restrict__ idx) {
  extern __shared__ unsigned bq_bits[];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  if (m >= M) return;
  const int W = (N + 31) >> 5;
  unsigned* bm = bq_bits + wave * W;
  center += ((size_t)b * M + m) * 3;
  start += (size_t)b * (BQ_T + 1);
  cells += (size_t)b * N;
  idx += ((size_t)b * M + m) * ns;
  const float cx = center[0], cy = center[1], cz = center[2];
  for (int w = lane; w < W; w += 64) bm[w] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the wave's own LDS traffic, in order
  __builtin_amdgcn_wave_barrier();
  // lanes 0..7: one of the 2 x 2 x 2 cells each -> its bucket's range
  int rs = 0, re = 0;
  if (lane < 8) {
    const float fx = cx * inv, fy = cy * inv, fz = cz * inv;
    const int ix = (int)floorf(fx), iy = (int)floorf(fy), iz = (int)floorf(fz);
    const int sx = (fx - (float)ix) < 0.5f ? -1 : 1, sy = (fy - (float)iy) < 0.5f ? -1 : 1,
              sz = (fz - (float)iz) < 0.5f ? -1 : 1;
    const int h = bq_hash(ix + ((lane & 1) ? sx : 0), iy + ((lane & 2) ? sy : 0), iz + ((lane & 4) ? sz : 0));
    rs = start[h];
    re = start[h + 1];
  }
#pragma unroll 1
  for (int c = 0; c < 8; ++c) {
    const int s0 = __shfl(rs, c, 64), e0 = __shfl(re, c, 64);
    // (two of the 8 cells in one bucket: the second visit would only set the same bits again - skipped)
    bool dup = false;
    for (int j = 0; j < c; ++j) dup = dup || (__shfl(rs, j, 64) == s0 && __shfl(re, j, 64) == e0);
    if (dup || s0 == e0) continue;
    for (int k = s0 + lane; k < e0; k += 64) {
      const float4 q = cells[k];
      if (dist2(cx - q.x, cy - q.y, cz - q.z) < max_r2) {
        const int i = __builtin_bit_cast(int, q.w);
        atomicOr(&bm[i >> 5], 1u << (i & 31));
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // the set bits in ascending order: the first ns hits by point index
  int cnt = 0, first = 0;
  for (int w0 = 0; w0 < W && cnt < ns; w0 += 64) {
    const int w = w0 + lane;
    unsigned word = w < W ? bm[w] : 0u;
    const unsigned long long nz = __ballot(word != 0u);
    if (nz == 0ull) continue;
    if (cnt == 0) {
      const int l0 = __builtin_ctzll(nz);
      first = (w0 + l0) * 32 + __builtin_ctz(__shfl(word, l0, 64));
    }
    const int pc = __builtin_popcount(word);
    int inc = pc;
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    int pos = cnt + inc - pc;
    while (word != 0u && pos < ns) {
      const int bit = __builtin_ctz(word);
      idx[pos++] = w * 32 + bit;
      word &= word - 1u;
    }
    cnt += __shfl(inc, 63, 64);
  }
  // slots never reached keep the first hit (or 0 when the ball is empty)
  for (int s = min(cnt, ns) + lane; s < ns; s += 64) idx[s] = first;
}

}  // namespace demf

using namespace demf;

extern "C" int demf_ball_query_grid_ws(int B, int N, long long* start_ints, long long* cell_floats) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && start_ints && cell_floats, "ball_query_grid_ws: bad arguments");
  *start_ints = (long long)B * (BQ_T + 1);
  *cell_floats = 4ll * B * N;
  return DEMF_OK;
}

extern "C" int demf_ball_query_grid_f32(int B, int N, int M, float max_radius, int nsample,
                                        const float* center_xyz, const float* xyz, int* idx,
                                        int* ws_start, float* ws_cells, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && N <= BQ_MAXN && M >= 0 && nsample >= 1 && max_radius > 0.f,
               "ball_query_grid: bad sizes B=%d N=%d M=%d ns=%d (N <= %d, radius > 0)", B, N, M, nsample, BQ_MAXN);
  if (B == 0 || M == 0) return DEMF_OK;
  DEMF_REQUIRE(center_xyz && xyz && idx && ws_start && ws_cells, "ball_query_grid: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const float inv = 1.0f / (2.0f * max_radius * 1.001f);
  hipLaunchKernelGGL(bq_grid_build_k, dim3(B), dim3(1024), 0, s, N, inv, xyz, ws_start,
                     reinterpret_cast<float4*>(ws_cells));
  if (int e = check_launch("ball_query_grid(build)")) return e;
  const size_t lds = sizeof(unsigned) * 4 * ((N + 31) / 32);
  hipLaunchKernelGGL(ball_query_grid_k, dim3(cdiv(M, 4), B), dim3(256), lds, s, N, M, inv,
                     max_radius * max_radius, nsample, center_xyz, ws_start,
                     reinterpret_cast<const float4*>(ws_cells), idx);
  return check_launch("ball_query_grid");
}

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
