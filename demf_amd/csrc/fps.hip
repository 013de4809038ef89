// Furthest point sampling for gfx950.
//
// Replaces mmdet3d.ops.furthest_point_sample as called at
// demf/modeling/heads/class_agnostic_vote_head.py:429-430 and inside every
// PointSAModule (configs/demf/demf_votenet.py:48-62,155-162).
//
// FPS is a chain of M-1 dependent arg-max rounds: it is bound by the latency of
// one round, not by HBM.  Design for CDNA4:
//   * one workgroup per scene; the scene's points and their running min-distance
//     live in VGPRs for the whole kernel (20 000 points = 20 per lane at 1024
//     threads): global memory is read once on entry and written once on exit;
//   * per round each lane does 8 VALU per point (3 sub, mul, 2 fma, min, max - the
//     min/max are raw v_min/v_max, no canonicalisation) and keeps only the VALUE of
//     its maximum; the wave maximum is 6 in-place v_max_f32_dpp;
//   * round synchronisation is two raw s_barriers with LDS-only waits (a
//     __syncthreads() would also drain vmcnt, i.e. wait for global stores):
//     (1) the 16 wave maxima are exchanged, every wave reduces them redundantly and
//     learns which wave owns the winner; (2) ONLY that wave searches its registers
//     for the winner's slot (compare/select chain on an otherwise idle SIMD) and
//     publishes {index, x, y, z}; everyone picks it up after the second barrier;
//   * selected indices are buffered in LDS and flushed coalesced (no per-round
//     global store on the critical path).
//   * thread t owns points t, t+BS, t+2BS, ... with BS = min(1024, 2^floor(log2 N))
//     - the same ownership as the upstream block reduction - so "first maximum in
//     the thread, then lowest thread id" reproduces upstream tie-breaking exactly.
#include <type_traits>
#include "common.h"

namespace demf {

__device__ __forceinline__ float raw_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float raw_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float raw_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// in-place wave64 max; lane 63 ends up with the maximum (returned wave-uniform)
__device__ __forceinline__ float wave_max_dpp(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return readlane_f(v, 63);
}

// max over each 16-lane row, every lane of the row gets it
__device__ __forceinline__ float row16_max_dpp(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}


using f2 = float __attribute__((ext_vector_type(2)));

constexpr int FPS_IDX_CHUNK = 2048;  // selected indices buffered in LDS between flushes

#ifdef DEMF_FPS_PROFILE  // tools/ubench/fps_prof.cpp: per-phase shader-cycle accounting (wave 0)
__device__ long long g_fps_prof[8];
__device__ long long g_prune_prof[16][8];   // tools/ubench/fps_prune_prof.cpp: per wave of scene 0
#define FPS_T(k) const long long t##k = __builtin_readcyclecounter();
#define FPS_ACC(i, a, b) prof[i] += (b) - (a);
#else
#define FPS_T(k)
#define FPS_ACC(i, a, b)
#endif

constexpr int FPS_PREFIX_MAX = 1024;   // largest M for the ordered-input check (12 KB of LDS)

// ---- already-ordered input?  The points of SA level l+1 are the FPS order of level l, and the
// first M points of an FPS order are the FPS order of that cloud again - UNLESS a tie is broken
// differently in the subset.  That can be verified without the M-round dependency chain: with
// E[k] = min_{i<k} d(q_i, q_k), the sequential algorithm picks k at step k iff E[k] is strictly
// greater than min_{i<k} d(q_i, q_j) for every j > k - N*M/1024 independent distance evaluations
// per thread, no barrier inside.  A first pass over the first 16 steps rejects unordered clouds
// in a microsecond.  On success idx = 0..M-1 is written and flag[b] = 1 makes the FPS kernel
// return at once; any tie or violation leaves flag[b] = 0 and the real loop runs, so the result is
// always the reference's.
__global__ __launch_bounds__(1024) void fps_ordered_check_k(int N, int M,
                                                            const float* __restrict__ xyz,
                                                            int* __restrict__ idx,
                                                            int* __restrict__ flag) {
  constexpr int BS = 1024;
  asm volatile("" ::: "v127");                 // whole register file of the CU, as fps_reg_kernel
  __shared__ float4 s_q[FPS_PREFIX_MAX];       // {x, y, z, E[k]}: one broadcast 16-byte read per k
  __shared__ int s_bad;
  const int b = blockIdx.x, tid = threadIdx.x;
  xyz += (size_t)b * N * 3;
  idx += (size_t)b * M;
  for (int t = tid; t < M; t += BS) s_q[t] = make_float4(xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2], 0.f);
  if (tid == 0) { s_bad = 0; flag[b] = 0; }
  __syncthreads();
  auto e_of = [&](int k) {
    const float4 q = s_q[k];
    float e = 1e10f;
    int i = 0;
    for (; i + 8 <= k; i += 8) {
      float d[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const float4 o = s_q[i + v];
        d[v] = dist2(o.x - q.x, o.y - q.y, o.z - q.z);
      }
      e = fminf(e, fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])),
                         fminf(fminf(d[4], d[5]), fminf(d[6], d[7]))));
    }
    for (; i < k; ++i) {
      const float4 o = s_q[i];
      e = fminf(e, dist2(o.x - q.x, o.y - q.y, o.z - q.z));
    }
    return e;
  };
  // every j against the steps [k0, k1) given the running minimum over the steps before k0
  auto test = [&](int k0, int k1) {
    bool bad = false;
    for (int j = tid; j < N && !bad; j += BS) {
      const float qx = xyz[3 * j], qy = xyz[3 * j + 1], qz = xyz[3 * j + 2];
      float r = 1e10f;
      const int kmax = j < k1 ? j : k1;
      int k = 0;
      for (; k + 8 <= kmax && !bad; k += 8) {
        float d[8], e[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          const float4 o = s_q[k + v];
          d[v] = dist2(o.x - qx, o.y - qy, o.z - qz);
          e[v] = o.w;
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          bad |= (k + v >= k0 && k + v >= 1) && !(e[v] > r);
          r = fminf(r, d[v]);
        }
      }
      for (; k < kmax; ++k) {
        const float4 o = s_q[k];
        bad |= (k >= k0 && k >= 1) && !(o.w > r);
        r = fminf(r, dist2(o.x - qx, o.y - qy, o.z - qz));
      }
    }
    return bad;
  };
  const int quick = M < 16 ? M : 16;
  if (tid < quick) s_q[tid].w = e_of(tid);
  __syncthreads();
  if (test(0, quick)) s_bad = 1;
  __syncthreads();
  if (s_bad != 0) return;                       // the usual case for an arbitrary cloud
  float ek = 0.f;
  if (tid >= quick && tid < M) ek = e_of(tid);
  __syncthreads();
  if (tid >= quick && tid < M) s_q[tid].w = ek;
  __syncthreads();
  if (test(quick, M)) s_bad = 1;
  __syncthreads();
  if (s_bad == 0) {
    for (int t = tid; t < M; t += BS) idx[t] = t;
    if (tid == 0) flag[b] = 1;
  }
}

template <int BS, int PPT>
__global__ __launch_bounds__(BS) void fps_reg_kernel(int N, int M,
                                                     const float* __restrict__ xyz,
                                                     int* __restrict__ idx,
                                                     const int* __restrict__ skip) {
  constexpr int NW = BS / 64;
  // Claim the CU's whole vector register file (BS/256 waves per SIMD x the per-wave budget):
  // this latency-bound chain runs for milliseconds on one CU per scene, and any other kernel's
  // workgroup scheduled next to it becomes the straggler of that kernel.
  if constexpr (BS == 1024) asm volatile("" ::: "v127");
  else if constexpr (BS == 512) asm volatile("" ::: "v255");
  __shared__ float s_wmax[16];
  __shared__ __attribute__((aligned(16))) float s_res[4];  // {idx bits, x, y, z}
  __shared__ int s_idx[FPS_IDX_CHUNK];
  const int b = blockIdx.x;
  xyz += (size_t)b * N * 3;
  idx += (size_t)b * M;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // already-ordered input, verified by fps_ordered_check_k: nothing to do for this scene
  if (skip != nullptr && skip[b] != 0) return;

  // two points per 64-bit register pair: the distance math runs on the packed-fp32 pipe
  // (v_pk_add/mul/fma_f32, bit-identical to the scalar ops)
  static_assert(PPT % 2 == 0, "points are processed in pairs");
  constexpr int PP = PPT / 2;
  f2 px[PP], py[PP], pz[PP];
  float tmp[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int k = tid + p * BS;
    const bool ok = k < N;
    px[p / 2][p % 2] = ok ? xyz[3 * k + 0] : 0.f;
    py[p / 2][p % 2] = ok ? xyz[3 * k + 1] : 0.f;
    pz[p / 2][p % 2] = ok ? xyz[3 * k + 2] : 0.f;
    tmp[p] = ok ? 1e10f : -2.f;  // pad slots can never reach the maximum
  }
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (tid == 0) s_idx[0] = 0;

#ifdef DEMF_FPS_PROFILE
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int j = 1; j < M; ++j) {
    FPS_T(0)
    // ---- update running distances, keep only the lane's maximum VALUE
    float best = -1.f;
    const f2 X1 = {x1, x1}, Y1 = {y1, y1}, Z1 = {z1, z1};
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const f2 dx = px[i] - X1, dy = py[i] - Y1, dz = pz[i] - Z1;
      f2 d = dy * dy;                               // dist2(): fma(dz,dz,fma(dx,dx,dy*dy))
      d = __builtin_elementwise_fma(dx, dx, d);
      d = __builtin_elementwise_fma(dz, dz, d);
      tmp[2 * i] = raw_min(d[0], tmp[2 * i]);
      tmp[2 * i + 1] = raw_min(d[1], tmp[2 * i + 1]);
      best = raw_max3(best, tmp[2 * i], tmp[2 * i + 1]);
    }
    FPS_T(1)
    const float wmax = wave_max_dpp(best);
    FPS_T(2)

    // ---- which wave holds the block maximum (lowest wave on ties)
    float gmax;
    int wwin;
    if constexpr (NW == 1) {
      gmax = wmax;
      wwin = 0;
    } else {
      if (lane == 0) s_wmax[wave] = wmax;
      lds_barrier();
      const float v = s_wmax[lane & (NW - 1)];
      gmax = readlane_f(row16_max_dpp(v), 0);
      wwin = __builtin_ctzll(__ballot(v == gmax));
    }
    FPS_T(3)

    // ---- only the winning wave locates the slot and publishes the point
    if (wave == wwin) {
      float sx = 0.f, sy = 0.f, sz = 0.f;
      int sp = 0;
#pragma unroll
      for (int p = PPT - 1; p >= 0; --p) {  // descending: the lowest matching slot wins
        const bool c = tmp[p] == gmax;
        sx = c ? px[p / 2][p % 2] : sx;
        sy = c ? py[p / 2][p % 2] : sy;
        sz = c ? pz[p / 2][p % 2] : sz;
        sp = c ? p : sp;
      }
      const int wl = __builtin_ctzll(__ballot(best == gmax));  // lowest lane = lowest thread
      const int wi = wave * 64 + wl + readlane_i(sp, wl) * BS;
      const float wx = readlane_f(sx, wl), wy = readlane_f(sy, wl), wz = readlane_f(sz, wl);
      if constexpr (NW == 1) {
        x1 = wx;
        y1 = wy;
        z1 = wz;
        if (lane == 0) s_idx[j & (FPS_IDX_CHUNK - 1)] = wi;
      } else if (lane == 0) {
        s_res[0] = __builtin_bit_cast(float, wi);
        s_res[1] = wx;
        s_res[2] = wy;
        s_res[3] = wz;
        s_idx[j & (FPS_IDX_CHUNK - 1)] = wi;
      }
    }
    FPS_T(4)
    if constexpr (NW > 1) {
      lds_barrier();
      const float4 r = *reinterpret_cast<const float4*>(s_res);
      x1 = readlane_f(r.y, 0);
      y1 = readlane_f(r.z, 0);
      z1 = readlane_f(r.w, 0);
    }
    FPS_T(5)
    FPS_ACC(0, t0, t1) FPS_ACC(1, t1, t2) FPS_ACC(2, t2, t3) FPS_ACC(3, t3, t4) FPS_ACC(4, t4, t5)
    // ---- flush a full chunk of selected indices (coalesced)
    if (((j + 1) & (FPS_IDX_CHUNK - 1)) == 0) {
      if constexpr (NW == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else lds_barrier();
      const int base = j + 1 - FPS_IDX_CHUNK;
      for (int t = tid; t < FPS_IDX_CHUNK; t += BS) idx[base + t] = s_idx[t];
      if constexpr (NW > 1) lds_barrier();
    }
  }
#ifdef DEMF_FPS_PROFILE
  if (tid == 0 && b == 0)
    for (int i = 0; i < 8; ++i) g_fps_prof[i] = prof[i];
#endif
  // tail flush
  if constexpr (NW > 1) lds_barrier();
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int base = M & ~(FPS_IDX_CHUNK - 1);
  for (int t = tid; base + t < M; t += BS) idx[base + t] = s_idx[t];
}

// ---- exact spatially pruned FPS for large clouds -------------------------------------------------------
// A round of fps_reg_kernel costs ~3 200 cycles, of which ~1 700 are the SIMDs' VALU issue: every one of the
// 16 waves updates its 20 x 64 points although, once a few hundred samples exist, a new sample can only
// lower the running distance of points within the current maximum of it.  Here the scene's points are first
// binned into 16^3 cells in Hilbert order (fps_sort_k) and a wave owns a CONTIGUOUS run of the cell order, i.e. a
// compact block of space.  Per round a wave evaluates the canonical squared distance from the new sample to
// its block's bounding box, lb - with the same operations in the same order as dist2(), so by the monotonicity
// of IEEE rounding lb <= d(p) for every point p of the block - and if lb >= the block's cached maximum of the
// running distances, then min(d, tmp) == tmp for all of them: the wave skips the update and contributes its
// cached maximum.  Picks are bit-identical to the full update.  What the spatial order destroys is the upstream
// tie rule "first maximum in the thread, then lowest thread" = lowest (k mod 1024, k / 1024); every point
// therefore carries that key (15 bits, two per register) and ties are resolved by it explicitly: inside a wave
// by a min-key chain over the slots that hold the maximum, across waves by a second exchange among the waves
// whose maxima tie (usually one).  Two barriers per round, as before.
constexpr int FPS_CELLS = 4096;

// 12-bit Hilbert index of a cell (x, y, z in 0..15) - Skilling's axes-to-transpose transform.  A Morton
// (Z-order) run of equal point COUNT jumps across space wherever it crosses a high-level cell boundary
// (measured: 3 of 16 waves with a bounding box as large as the scene, updating in 55-80 % of the rounds);
// consecutive Hilbert cells are always face neighbours, so every run is one connected, compact region.
__device__ __forceinline__ unsigned hilbert12(unsigned x, unsigned y, unsigned z) {
  unsigned X[3] = {x, y, z};
#pragma unroll
  for (unsigned Q = 8; Q > 1; Q >>= 1) {
    const unsigned P = Q - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) X[0] ^= P;
      else { const unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  unsigned t = 0;
#pragma unroll
  for (unsigned Q = 8; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  unsigned idx = 0;
#pragma unroll
  for (int j = 3; j >= 0; --j)
#pragma unroll
    for (int i = 0; i < 3; ++i) idx = (idx << 1) | ((X[i] >> j) & 1u);
  return idx;
}

// perm[b][i] = original index of the i-th point of scene b in Hilbert-cell order (order inside a cell
// arbitrary: the sampling result does not depend on it)
__global__ __launch_bounds__(1024) void fps_sort_k(int N, const float* __restrict__ xyz, int* __restrict__ perm) {
  __shared__ int s_cnt[FPS_CELLS];
  __shared__ float s_red[6][16];
  __shared__ int s_part[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  xyz += (size_t)b * N * 3;
  perm += (size_t)b * N;
  float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
  for (int k = tid; k < N; k += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float v = xyz[3 * k + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float mn = -wave_allmax(-lo[a]), mx = wave_allmax(hi[a]);
    if (lane == 0) { s_red[a][wave] = mn; s_red[3 + a][wave] = mx; }
  }
  for (int i = tid; i < FPS_CELLS; i += 1024) s_cnt[i] = 0;
  __syncthreads();
  float inv[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float mn = s_red[a][0], mx = s_red[3 + a][0];
    for (int w = 1; w < 16; ++w) { mn = fminf(mn, s_red[a][w]); mx = fmaxf(mx, s_red[3 + a][w]); }
    lo[a] = mn;
    inv[a] = mx > mn ? 16.0f / (mx - mn) : 0.f;
  }
  auto cell_of = [&](int k) {
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int v = (int)((xyz[3 * k + a] - lo[a]) * inv[a]);
      q[a] = (unsigned)(v < 0 ? 0 : (v > 15 ? 15 : v));
    }
    return (int)hilbert12(q[0], q[1], q[2]);
  };
  for (int k = tid; k < N; k += 1024) atomicAdd(&s_cnt[cell_of(k)], 1);
  __syncthreads();
  // exclusive scan of the 4096 counters: 4 per thread, wave scan, 16 wave totals
  int c4[4], run = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { c4[i] = s_cnt[4 * tid + i]; run += c4[i]; }
  int inc = run;
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) s_part[wave] = inc;
  __syncthreads();
  int base = inc - run;
  for (int w = 0; w < wave; ++w) base += s_part[w];
#pragma unroll
  for (int i = 0; i < 4; ++i) { s_cnt[4 * tid + i] = base; base += c4[i]; }
  __syncthreads();
  for (int k = tid; k < N; k += 1024) perm[atomicAdd(&s_cnt[cell_of(k)], 1)] = k;
}

// unsigned minimum over the wave / over each 16-lane row, on DPP (a __shfl_xor butterfly is six LDS-crossbar
// round trips: ~700 cycles per reduction, measured as +1 200 cycles per round)
__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned row16_min_u32(unsigned v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}
// wave64 minimum, returned wave-uniform (lane 63 holds it after the two row broadcasts)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = row16_min_u32(v);
  asm volatile(
      "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

struct __attribute__((aligned(32))) FpsCand { float v; unsigned tk; float x, y, z; };

// Exact farthest point sampling with bounding-box pruning.  The points of a scene arrive in Hilbert-cell
// order (perm); wave w owns positions [w*64*PPT, (w+1)*64*PPT), two register slots (a PAIR, 128 consecutive
// positions of the curve = one compact region) per packed register.  Lane g of the wave keeps the bounding
// box of pair g, so ONE pass of the distance arithmetic tests all pairs of the wave:
//   * a new sample can lower a running distance of a pair only if its distance to the pair's box is below
//     the wave's maximum running distance cv.  The box distance is the same rounded operations as dist2()
//     on offsets no larger in magnitude than any point's, so it is <= every computed point distance
//     (rounding is monotone); a pair that is skipped would have changed nothing.
//   * the wave's candidate (cv, the lowest tie key that holds it, that point's coordinates) is cached.
//     Lane 63 carries the candidate point itself as a degenerate box, whose "box distance" IS the
//     candidate's distance to the new sample, bit for bit: while that stays >= cv the candidate keeps its
//     value, still holds the maximum (no distance ever grows) and is still the lowest key that holds it -
//     so the wave reduces and searches again only in the rounds that lower its own candidate.
// One barrier per round: waves publish their candidate (double-buffered by round parity, rewritten only
// for two rounds after a change), every wave reduces the 16.  Picks are bit-identical to the full update.
// NW waves per scene: 16 (1 024 threads, PPT <= 20 slots per lane) or 8 (512 threads, PPT <= 40: the per-round
// skeleton - box test, candidate exchange, reduce - is issued by two waves per SIMD instead of four).
template <int PPT, int NW = 16>
__global__ __launch_bounds__(64 * NW) void fps_prune_kernel(int N, int M, const float* __restrict__ xyz,
                                                         const int* __restrict__ perm, int* __restrict__ idx) {
  constexpr int BS = 64 * NW, PP = PPT / 2, SB = PPT > 32 ? 6 : 5;      // SB: bits of the slot number under the tie key
  static_assert(PPT % 2 == 0 && PP < 63, "pairs of slots, lane 63 is the candidate's");
  if constexpr (NW == 16) asm volatile("" ::: "v127");   // the CU's whole register file, as fps_reg_kernel
  __shared__ FpsCand s_cand[2][16];
  __shared__ int s_idx[FPS_IDX_CHUNK];
  const int b = blockIdx.x;
  xyz += (size_t)b * N * 3;
  perm += (size_t)b * N;
  idx += (size_t)b * M;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f2 px[PP], py[PP], pz[PP];
  float tmp[PPT];
  unsigned kp[PPT];                               // (tie key << 5) | slot
  float blx = 1e30f, bly = 1e30f, blz = 1e30f, bhx = -1e30f, bhy = -1e30f, bhz = -1e30f;
  // a valid point of the wave for its padding slots
  const int first = wave * 64 * PPT;
  const int kf = perm[first < N ? first : 0];
  const float fx = xyz[3 * kf], fy = xyz[3 * kf + 1], fz = xyz[3 * kf + 2];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    float l[3] = {1e30f, 1e30f, 1e30f}, h[3] = {-1e30f, -1e30f, -1e30f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = 2 * i + q;
      const int pos = first + p * 64 + lane;
      const bool ok = pos < N;
      const int k = ok ? perm[pos] : 0;
      const float x = ok ? xyz[3 * k] : fx, y = ok ? xyz[3 * k + 1] : fy, z = ok ? xyz[3 * k + 2] : fz;
      px[i][q] = x; py[i][q] = y; pz[i][q] = z;
      tmp[p] = ok ? 1e10f : -2.f;                 // pad slots can never reach the maximum
      const unsigned key = ok ? (((unsigned)k & 1023u) << 5) | ((unsigned)k >> 10) : 0x7FFFu;
      kp[p] = (key << SB) | (unsigned)p;
      if (ok) {
        l[0] = fminf(l[0], x); h[0] = fmaxf(h[0], x);
        l[1] = fminf(l[1], y); h[1] = fmaxf(h[1], y);
        l[2] = fminf(l[2], z); h[2] = fmaxf(h[2], z);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { l[a] = -wave_allmax(-l[a]); h[a] = wave_allmax(h[a]); }
    if (lane == i) { blx = l[0]; bly = l[1]; blz = l[2]; bhx = h[0]; bhy = h[1]; bhz = h[2]; }
  }
  if (lane == 63) { blx = bly = blz = bhx = bhy = bhz = 0.f; }     // any finite point: round 1 searches
  float cv = first < N ? 1e10f : -2.f;
  unsigned ctk = 0xFFFFFFFFu;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  int dirty = 2;
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (tid == 0) s_idx[0] = 0;

#ifdef DEMF_FPS_PROFILE
  long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int j = 1; j < M; ++j) {
#ifdef DEMF_FPS_PROFILE
    const long long pt0 = __builtin_readcyclecounter();
#endif
    // ---- every pair's box (and the candidate) against the new sample, one lane each
    const float ex = raw_max3(blx - x1, x1 - bhx, 0.f);
    const float ey = raw_max3(bly - y1, y1 - bhy, 0.f);
    const float ez = raw_max3(blz - z1, z1 - bhz, 0.f);
    const float lb = dist2(ex, ey, ez);
    const unsigned long long need = __ballot(!(lb >= cv));
    if (need) {                                           // (wave-uniform)
      const f2 X1 = {x1, x1}, Y1 = {y1, y1}, Z1 = {z1, z1};
#pragma unroll
      for (int i = 0; i < PP; ++i) {
        if ((need >> i) & 1ull) {
          const f2 dx = px[i] - X1, dy = py[i] - Y1, dz = pz[i] - Z1;
          f2 d = dy * dy;                               // dist2(): fma(dz,dz,fma(dx,dx,dy*dy))
          d = __builtin_elementwise_fma(dx, dx, d);
          d = __builtin_elementwise_fma(dz, dz, d);
          tmp[2 * i] = raw_min(d[0], tmp[2 * i]);
          tmp[2 * i + 1] = raw_min(d[1], tmp[2 * i + 1]);
        }
      }
#ifdef DEMF_FPS_PROFILE
      const long long pts = __builtin_readcyclecounter();
      pp[7] += pts - pt0;                                 // box test + update, rounds with an update
#endif
      if (need >> 63) {                                   // the candidate's own distance fell: search again
        float best = -1.f;
#pragma unroll
        for (int i = 0; i < PP; ++i) best = raw_max3(best, tmp[2 * i], tmp[2 * i + 1]);
        cv = wave_max_dpp(best);
        // lowest (tie key, slot) among the wave's points that hold the maximum
        unsigned bt = 0xFFFFFFFFu;
#pragma unroll
        for (int p = 0; p < PPT; ++p) bt = tmp[p] == cv ? umin(bt, kp[p]) : bt;
        const unsigned btmin = wave_min_u32(bt);
        const int wl = __builtin_ctzll(__ballot(bt == btmin));
        ctk = btmin >> SB;
        // its coordinates: a scalar branch on the (wave-uniform) slot, one v_readlane per coordinate
#define DEMF_FPS_PICK(P)                                                                       \
  case P:                                                                                      \
    if constexpr ((P) < PPT) {                                                                 \
      cx = readlane_f(px[(P) / 2][(P) % 2], wl);                                               \
      cy = readlane_f(py[(P) / 2][(P) % 2], wl);                                               \
      cz = readlane_f(pz[(P) / 2][(P) % 2], wl);                                               \
    }                                                                                          \
    break;
        switch ((int)(btmin & ((1u << SB) - 1u))) {
          DEMF_FPS_PICK(0) DEMF_FPS_PICK(1) DEMF_FPS_PICK(2) DEMF_FPS_PICK(3) DEMF_FPS_PICK(4) DEMF_FPS_PICK(5)
          DEMF_FPS_PICK(6) DEMF_FPS_PICK(7) DEMF_FPS_PICK(8) DEMF_FPS_PICK(9) DEMF_FPS_PICK(10) DEMF_FPS_PICK(11)
          DEMF_FPS_PICK(12) DEMF_FPS_PICK(13) DEMF_FPS_PICK(14) DEMF_FPS_PICK(15) DEMF_FPS_PICK(16)
          DEMF_FPS_PICK(17) DEMF_FPS_PICK(18) DEMF_FPS_PICK(19) DEMF_FPS_PICK(20) DEMF_FPS_PICK(21)
          DEMF_FPS_PICK(22) DEMF_FPS_PICK(23) DEMF_FPS_PICK(24) DEMF_FPS_PICK(25) DEMF_FPS_PICK(26) DEMF_FPS_PICK(27)
          DEMF_FPS_PICK(28) DEMF_FPS_PICK(29) DEMF_FPS_PICK(30) DEMF_FPS_PICK(31) DEMF_FPS_PICK(32) DEMF_FPS_PICK(33)
          DEMF_FPS_PICK(34) DEMF_FPS_PICK(35) DEMF_FPS_PICK(36) DEMF_FPS_PICK(37) DEMF_FPS_PICK(38) DEMF_FPS_PICK(39)
          default: break;
        }
#undef DEMF_FPS_PICK
        if (lane == 63) { blx = bhx = cx; bly = bhy = cy; blz = bhz = cz; }
        dirty = 2;
#ifdef DEMF_FPS_PROFILE
        pp[6] += __builtin_readcyclecounter() - pts;      // the search alone
#endif
      }
    }
    // ---- block maximum, lowest tie key among the waves that hold it
#ifdef DEMF_FPS_PROFILE
    const long long pt1 = __builtin_readcyclecounter();
    pp[0] += pt1 - pt0;
    pp[3] += need != 0;
    pp[4] += (need >> 63) != 0;
    pp[5] += __builtin_popcountll(need & ~(1ull << 63));
#endif
    FpsCand* slot = s_cand[j & 1];
    if (dirty) {
      if (lane == 0) slot[wave] = FpsCand{cv, ctk, cx, cy, cz};
      --dirty;
    }
    lds_barrier();
#ifdef DEMF_FPS_PROFILE
    const long long pt2 = __builtin_readcyclecounter();
    pp[1] += pt2 - pt1;
#endif
    {
      const FpsCand c = slot[lane & (NW - 1)];           // every 16-lane row: all NW waves (twice over at NW = 8)
      const float gmax = readlane_f(row16_max_dpp(c.v), 0);
      const unsigned long long held = __ballot(c.v == gmax) & ((1ull << NW) - 1ull);
      int wv = __builtin_ctzll(held);
      if (held & (held - 1)) {                           // several waves hold it: lowest tie key
        const unsigned key = c.v == gmax ? c.tk : 0xFFFFFFFFu;
        const unsigned tmin = (unsigned)__builtin_amdgcn_readfirstlane((int)row16_min_u32(key));
        wv = __builtin_ctzll(__ballot(key == tmin));
      }
      x1 = readlane_f(c.x, wv);
      y1 = readlane_f(c.y, wv);
      z1 = readlane_f(c.z, wv);
      if (tid == 0) {
        const unsigned tk = (unsigned)__builtin_amdgcn_readlane((int)c.tk, wv);
        s_idx[j & (FPS_IDX_CHUNK - 1)] = (int)(((tk & 31u) << 10) | (tk >> 5));
      }
    }
#ifdef DEMF_FPS_PROFILE
    pp[2] += __builtin_readcyclecounter() - pt2;
#endif
    if (((j + 1) & (FPS_IDX_CHUNK - 1)) == 0) {
      lds_barrier();
      const int base = j + 1 - FPS_IDX_CHUNK;
      for (int t = tid; t < FPS_IDX_CHUNK; t += BS) idx[base + t] = s_idx[t];
      lds_barrier();
    }
  }
#ifdef DEMF_FPS_PROFILE
  if (b == 0 && lane == 0 && wave < 16)
    for (int i = 0; i < 8; ++i) g_prune_prof[wave][i] = pp[i];
#endif
  lds_barrier();
  const int base = M & ~(FPS_IDX_CHUNK - 1);
  for (int t = tid; base + t < M; t += BS) idx[base + t] = s_idx[t];
}

// ---- fps_prune_kernel with the 20-slot re-search replaced by per-PAIR cached maxima (round 6) ----------------------
// In fps_prune_kernel the wave whose candidate was consumed - one per round, always on the round's critical path - runs
// the full search of its 20 slots: a max chain, a wave maximum, a 20-step key-select chain, a wave minimum, a slot
// switch: ~230 dependent instructions, 1 074 cycles, after a 534-cycle update (profiles/r05_fps_exchange_variants.log).
// Here lane i of the wave keeps, next to the bounding box of pair i, that pair's exact BEST point: (running distance,
// (key << 5 | slot), coordinates).  A new sample can change a pair's best only by lowering that very point (every other
// distance can only fall), which one extra vector pass over the pair lanes detects (the sample's distance to the cached
// point, bit for bit the update's own arithmetic).  Only THOSE pairs - the consumed candidate's pair, now and then a
// neighbour - are re-searched: 2 slots x 64 lanes each (one wave maximum, one wave minimum: ~35 instructions).  The
// wave's candidate is the best of its <= 16 pair lanes: one row reduction.  Same exchange, same barrier, same picks.
template <int PPT, int NW = 16>
__global__ __launch_bounds__(64 * NW) void fps_pair_kernel(int N, int M, const float* __restrict__ xyz,
                                                           const int* __restrict__ perm, int* __restrict__ idx) {
  constexpr int BS = 64 * NW, PP = PPT / 2, SB = PPT > 32 ? 6 : 5;
  static_assert(PPT % 2 == 0 && PP <= 32, "pairs of slots, pair lanes 0 .. PP-1");
  if constexpr (NW == 16) asm volatile("" ::: "v127");   // the CU's whole register file, as fps_reg_kernel
  __shared__ FpsCand s_cand[2][16];
  __shared__ int s_idx[FPS_IDX_CHUNK];
  const int b = blockIdx.x;
  xyz += (size_t)b * N * 3;
  perm += (size_t)b * N;
  idx += (size_t)b * M;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f2 px[PP], py[PP], pz[PP];
  float tmp[PPT];
  unsigned kp[PPT];                               // (tie key << 5) | slot
  float blx = 1e30f, bly = 1e30f, blz = 1e30f, bhx = -1e30f, bhy = -1e30f, bhz = -1e30f;
  float pbv = -2.f, pbx = 0.f, pby = 0.f, pbz = 0.f;          // lane i < PP: pair i's best point
  unsigned pbk = 0xFFFFFFFFu;
  const int first = wave * 64 * PPT;
  const int kf = perm[first < N ? first : 0];
  const float fx = xyz[3 * kf], fy = xyz[3 * kf + 1], fz = xyz[3 * kf + 2];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    float l[3] = {1e30f, 1e30f, 1e30f}, h[3] = {-1e30f, -1e30f, -1e30f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = 2 * i + q;
      const int pos = first + p * 64 + lane;
      const bool ok = pos < N;
      const int k = ok ? perm[pos] : 0;
      const float x = ok ? xyz[3 * k] : fx, y = ok ? xyz[3 * k + 1] : fy, z = ok ? xyz[3 * k + 2] : fz;
      px[i][q] = x; py[i][q] = y; pz[i][q] = z;
      tmp[p] = ok ? 1e10f : -2.f;                 // pad slots can never reach the maximum
      const unsigned key = ok ? (((unsigned)k & 1023u) << 5) | ((unsigned)k >> 10) : 0x7FFFu;
      kp[p] = (key << SB) | (unsigned)p;
      if (ok) {
        l[0] = fminf(l[0], x); h[0] = fmaxf(h[0], x);
        l[1] = fminf(l[1], y); h[1] = fmaxf(h[1], y);
        l[2] = fminf(l[2], z); h[2] = fmaxf(h[2], z);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { l[a] = -wave_allmax(-l[a]); h[a] = wave_allmax(h[a]); }
    if (lane == i) { blx = l[0]; bly = l[1]; blz = l[2]; bhx = h[0]; bhy = h[1]; bhz = h[2]; }
  }
  // pair i's exact best: maximum of its 128 running distances, lowest (key, slot) among the holders
  auto pair_best = [&](int i) {
    const float m = raw_max(tmp[2 * i], tmp[2 * i + 1]);
    const float mx = wave_max_dpp(m);
    unsigned k = tmp[2 * i] == mx ? kp[2 * i] : 0xFFFFFFFFu;
    k = tmp[2 * i + 1] == mx ? umin(k, kp[2 * i + 1]) : k;
    const unsigned kmin = wave_min_u32(k);
    const int wl = __builtin_ctzll(__ballot(k == kmin));
    const bool odd = kmin & 1u;                                  // slot 2i + 1
    const float x = readlane_f(odd ? px[i][1] : px[i][0], wl);
    const float y = readlane_f(odd ? py[i][1] : py[i][0], wl);
    const float z = readlane_f(odd ? pz[i][1] : pz[i][0], wl);
    if (lane == i) { pbv = mx; pbk = kmin; pbx = x; pby = y; pbz = z; }
  };
#pragma unroll
  for (int i = 0; i < PP; ++i) pair_best(i);
  float cv = first < N ? 1e10f : -2.f;
  unsigned ctk = 0xFFFFFFFFu;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  // the wave's candidate: best of the pair lanes (one row of 16)
  auto wave_best = [&]() {
    const float v = lane < PP ? pbv : -3.f;
    if constexpr (PP <= 16) cv = readlane_f(row16_max_dpp(v), 0);
    else cv = wave_max_dpp(v);
    const unsigned k = (lane < PP && v == cv) ? pbk : 0xFFFFFFFFu;
    unsigned kmin;
    if constexpr (PP <= 16) kmin = (unsigned)__builtin_amdgcn_readfirstlane((int)row16_min_u32(k));
    else kmin = wave_min_u32(k);
    const int wl = __builtin_ctzll(__ballot(k == kmin));
    ctk = kmin >> SB;
    cx = readlane_f(pbx, wl); cy = readlane_f(pby, wl); cz = readlane_f(pbz, wl);
  };
  wave_best();
  int dirty = 2;
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (tid == 0) s_idx[0] = 0;

#ifdef DEMF_FPS_PROFILE
  long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int j = 1; j < M; ++j) {
#ifdef DEMF_FPS_PROFILE
    const long long pt0 = __builtin_readcyclecounter();
#endif
    // ---- every pair's box against the new sample, one lane each
    const float ex = raw_max3(blx - x1, x1 - bhx, 0.f);
    const float ey = raw_max3(bly - y1, y1 - bhy, 0.f);
    const float ez = raw_max3(blz - z1, z1 - bhz, 0.f);
    const float lb = dist2(ex, ey, ez);
    const unsigned need = (unsigned)(__ballot(lane < PP && !(lb >= cv)));
    if (need) {                                           // (wave-uniform)
      // which of those pairs lose their cached best point to this sample?
      const float hd = dist2(pbx - x1, pby - y1, pbz - z1);
      const unsigned hit = (unsigned)(__ballot(lane < PP && !(hd >= pbv))) & need;
      const f2 X1 = {x1, x1}, Y1 = {y1, y1}, Z1 = {z1, z1};
#pragma unroll
      for (int i = 0; i < PP; ++i) {
        if ((need >> i) & 1u) {
          const f2 dx = px[i] - X1, dy = py[i] - Y1, dz = pz[i] - Z1;
          f2 d = dy * dy;                               // dist2(): fma(dz,dz,fma(dx,dx,dy*dy))
          d = __builtin_elementwise_fma(dx, dx, d);
          d = __builtin_elementwise_fma(dz, dz, d);
          tmp[2 * i] = raw_min(d[0], tmp[2 * i]);
          tmp[2 * i + 1] = raw_min(d[1], tmp[2 * i + 1]);
          if ((hit >> i) & 1u) pair_best(i);
        }
      }
      if (hit) {
        wave_best();
        dirty = 2;
      }
#ifdef DEMF_FPS_PROFILE
      pp[7] += __builtin_readcyclecounter() - pt0;          // rounds with an update: test + update + re-search
      pp[3] += 1;
      pp[4] += hit != 0;
      pp[5] += __builtin_popcount(need);
      pp[6] += __builtin_popcount(hit);
#endif
    }
#ifdef DEMF_FPS_PROFILE
    const long long pt1 = __builtin_readcyclecounter();
    pp[0] += pt1 - pt0;
#endif
    // ---- block maximum, lowest tie key among the waves that hold it
    FpsCand* slot = s_cand[j & 1];
    if (dirty) {
      if (lane == 0) slot[wave] = FpsCand{cv, ctk, cx, cy, cz};
      --dirty;
    }
    lds_barrier();
#ifdef DEMF_FPS_PROFILE
    const long long pt2 = __builtin_readcyclecounter();
    pp[1] += pt2 - pt1;
#endif
    {
      const FpsCand c = slot[lane & (NW - 1)];           // every 16-lane row: all NW waves
      const float gmax = readlane_f(row16_max_dpp(c.v), 0);
      const unsigned long long held = __ballot(c.v == gmax) & ((1ull << NW) - 1ull);
      int wv = __builtin_ctzll(held);
      if (held & (held - 1)) {                           // several waves hold it: lowest tie key
        const unsigned key = c.v == gmax ? c.tk : 0xFFFFFFFFu;
        const unsigned tmin = (unsigned)__builtin_amdgcn_readfirstlane((int)row16_min_u32(key));
        wv = __builtin_ctzll(__ballot(key == tmin));
      }
      x1 = readlane_f(c.x, wv);
      y1 = readlane_f(c.y, wv);
      z1 = readlane_f(c.z, wv);
      if (tid == 0) {
        const unsigned tk = (unsigned)__builtin_amdgcn_readlane((int)c.tk, wv);
        s_idx[j & (FPS_IDX_CHUNK - 1)] = (int)(((tk & 31u) << 10) | (tk >> 5));
      }
    }
#ifdef DEMF_FPS_PROFILE
    pp[2] += __builtin_readcyclecounter() - pt2;
#endif
    if (((j + 1) & (FPS_IDX_CHUNK - 1)) == 0) {
      lds_barrier();
      const int base = j + 1 - FPS_IDX_CHUNK;
      for (int t = tid; t < FPS_IDX_CHUNK; t += BS) idx[base + t] = s_idx[t];
      lds_barrier();
    }
  }
#ifdef DEMF_FPS_PROFILE
  if (b == 0 && lane == 0 && wave < 16)
    for (int i = 0; i < 8; ++i) g_prune_prof[wave][i] = pp[i];
#endif
  lds_barrier();
  const int base = M & ~(FPS_IDX_CHUNK - 1);
  for (int t = tid; base + t < M; t += BS) idx[base + t] = s_idx[t];
}

struct __attribute__((aligned(32))) FpsSlot {
  float v;
  int i;
  int pad[6];
};

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

// The same verification spread over the chip (round 5).  fps_ordered_check_k squeezes its N*M pair evaluations
// through ONE compute unit per scene: 162 us at 2 048 -> 1 024, 288 us for the four checks of a step - all of it on
// the serial pre-pass chain (and at B = 1 the chip is otherwise idle).  Here
//   fps_check_e_k    : E[k] = min_{i<k} d(q_i, q_k); a workgroup takes 64 consecutive k, its 4 waves split the i range;
//   fps_check_test_k : one thread per j walks k < min(j, M) with the running minimum r and tests E[k] > r; a workgroup
//                      per 256 points; the verdict of a scene is ONE packed atomic (arrivals | violations << 16), the
//                      last workgroup to arrive writes flag[b] and idx = 0..M-1 - no fence, no second variable.
// Same predicate, same dist2(), same comparisons as above: the outcome is identical.
// Scratch (ints, in the caller's (B,N) temp): flag[B] | ticket[B] | E[B*M] (floats); needs 2 + M <= N.
__global__ __launch_bounds__(256) void fps_check_e_k(int N, int M, const float* __restrict__ xyz, int* __restrict__ flag,
                                                     int* __restrict__ ticket, float* __restrict__ E) {
  __shared__ float s_min[4][64];
  __shared__ float4 s_p[FPS_PREFIX_MAX];          // the earlier samples, one broadcast 16-byte read per i
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  xyz += (size_t)b * N * 3;
  if (blockIdx.x == 0 && threadIdx.x == 0) { flag[b] = 0; ticket[b] = 0; }
  const int k = blockIdx.x * 64 + lane;
  const int kk = k < M ? k : M - 1;
  const float qx = xyz[3 * kk], qy = xyz[3 * kk + 1], qz = xyz[3 * kk + 2];
  // i ranges over [0, 64 * blockIdx.x + 63): the wave's quarter, every lane masks i >= its own k
  const int hi = min(M, (int)blockIdx.x * 64 + 64);
  for (int t = threadIdx.x; t < hi; t += 256) s_p[t] = make_float4(xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2], 0.f);
  __syncthreads();
  const int per = (hi + 3) / 4;
  const int i0 = wave * per, i1 = min(hi, i0 + per);
  float e = 1e10f;
  int i = i0;
  for (; i + 4 <= i1; i += 4) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 o = s_p[i + v];
      const float d = dist2(o.x - qx, o.y - qy, o.z - qz);
      e = i + v < k ? fminf(e, d) : e;
    }
  }
  for (; i < i1; ++i) {
    const float4 o = s_p[i];
    const float d = dist2(o.x - qx, o.y - qy, o.z - qz);
    e = i < k ? fminf(e, d) : e;
  }
  s_min[wave][lane] = e;
  __syncthreads();
  if (wave == 0 && k < M)
    E[(size_t)b * M + k] = fminf(fminf(s_min[0][lane], s_min[1][lane]), fminf(s_min[2][lane], s_min[3][lane]));
}

__global__ __launch_bounds__(256) void fps_check_test_k(int N, int M, const float* __restrict__ xyz,
                                                        const float* __restrict__ E, int* __restrict__ idx,
                                                        int* __restrict__ flag, int* __restrict__ ticket) {
  // eight lanes per point j: lane c owns the steps k of the c-th eighth of [0, kmax).  The running minimum
  // r_k = min_{i<k} d(q_i, q_j) of the sequential test is (minimum over the earlier lanes' whole ranges) min (the
  // lane's own running minimum): a first sweep takes every lane's range minimum, an exclusive prefix minimum across
  // the eight lanes gives each its starting value, a second sweep makes the comparisons.  Twice the distance
  // evaluations, an eighth of the dependent chain; min is exact, so every comparison sees the same r_k.
  // {x, y, z, E[k]} at slot k + k/64: the eight lanes of a point read ranges `per` entries apart, and for the points
  // behind the prefix per = M/8 = 128 entries = 2 KB - the same banks for all eight (an 8-way conflict on every read of
  // both sweeps, for half of the points).  One 16-byte pad per 64 entries turns that stride into 2 080 bytes: eight
  // different 4-bank groups.
  __shared__ float4 s_q[FPS_PREFIX_MAX + FPS_PREFIX_MAX / 64 + 1];
#define SQI(k_) ((k_) + ((k_) >> 6))
  __shared__ int s_bad, s_last;
  const int b = blockIdx.y, tid = threadIdx.x;
  xyz += (size_t)b * N * 3;
  idx += (size_t)b * M;
  E += (size_t)b * M;
  for (int t = tid; t < M; t += 256) s_q[SQI(t)] = make_float4(xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2], E[t]);
  if (tid == 0) s_bad = 0;
  __syncthreads();
  const int c = tid & 7;
  const int j = blockIdx.x * 32 + (tid >> 3);
  const int jj = j < N ? j : N - 1;
  const float qx = xyz[3 * jj], qy = xyz[3 * jj + 1], qz = xyz[3 * jj + 2];
  const int kmax = j < N ? (j < M ? j : M) : 0;
  const int per = (kmax + 7) >> 3;
  const int k0 = min(kmax, c * per), k1 = min(kmax, k0 + per);
  float mine = 1e10f;
  for (int k = k0; k < k1; ++k) {
    const float4 o = s_q[SQI(k)];
    mine = fminf(mine, dist2(o.x - qx, o.y - qy, o.z - qz));
  }
  // exclusive prefix minimum over the 8 lanes of the group (lane c: ranges 0 .. c-1)
  float r = 1e10f;
#pragma unroll
  for (int d = 1; d < 8; ++d) {
    const float other = __shfl(mine, ((tid & 63) & ~7) + ((c - d) & 7), 64);
    r = d <= c ? fminf(r, other) : r;
  }
  bool bad = false;
  for (int k = k0; k < k1; ++k) {
    const float4 o = s_q[SQI(k)];
    bad |= (k >= 1) && !(o.w > r);
    r = fminf(r, dist2(o.x - qx, o.y - qy, o.z - qz));
  }
  if (bad) s_bad = 1;
  __syncthreads();
  if (tid == 0) {
    const int add = 1 + (s_bad ? 0x10000 : 0);
    const int now = atomicAdd(ticket + b, add) + add;
    s_last = ((now & 0xffff) == (int)gridDim.x) ? (now >> 16 ? 1 : 2) : 0;      // 2: last, no violation anywhere
  }
  __syncthreads();
  if (s_last == 2) {
    for (int t = tid; t < M; t += 256) idx[t] = t;
    if (tid == 0) flag[b] = 1;
  }
}
#undef SQI

template <int BS, int PPT>
static void launch_reg(int B, int N, int M, const float* xyz, int* idx, const int* skip,
                       hipStream_t s) {
  hipLaunchKernelGGL((fps_reg_kernel<BS, PPT>), dim3(B), dim3(BS), 0, s, N, M, xyz, idx, skip);
}

}  // namespace demf

using namespace demf;

static int fps_impl(int B, int N, int M, const float* xyz, float* temp, long long temp_floats, int* idx,
                    demf_stream_t stream);

extern "C" int demf_fps_f32(int B, int N, int M, const float* xyz, float* temp, int* idx,
                            demf_stream_t stream) {
  // (the scratch size is implied by the shape class: see include/demf_hip.h; the ordered-input check gets the
  // B words the contract promises, i.e. its one-CU form)
  return fps_impl(B, N, M, xyz, temp, temp != nullptr ? (long long)B : 0, idx, stream);
}

extern "C" int demf_fps_ws_f32(int B, int N, int M, const float* xyz, float* temp, long long temp_floats, int* idx,
                               demf_stream_t stream) {
  DEMF_REQUIRE(temp == nullptr || temp_floats >= B, "fps_ws: scratch of %lld floats for %d scenes", temp_floats, B);
  return fps_impl(B, N, M, xyz, temp, temp != nullptr ? temp_floats : 0, idx, stream);
}

static int fps_impl(int B, int N, int M, const float* xyz, float* temp, long long temp_floats, int* idx,
                    demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && M >= 0, "fps: bad sizes B=%d N=%d M=%d", B, N, M);
  if (B == 0 || M == 0) return DEMF_OK;
  DEMF_REQUIRE(xyz && idx, "fps: null pointer");
  hipStream_t s = (hipStream_t)stream;
  int bs = 1;
  while (bs * 2 <= N && bs < 1024) bs *= 2;  // upstream opt_n_threads()
  const int ppt = cdiv(N, bs);
  // level l+1 of the backbone is handed the FPS order of level l: try the dependency-free check
  // first (needs B ints of scratch in `temp`); scenes it verifies are skipped by the main kernel
  const int* skip = nullptr;
  if (temp != nullptr && M >= 2 && M <= FPS_PREFIX_MAX && N <= 4 * M && bs >= 64 && ppt <= 24) {
    static const int split = [] { const char* v = getenv("DEMF_FPS_CHECK_SPLIT"); return v ? atoi(v) : 1; }();   // A/B
    if (split && temp_floats >= (long long)B * (M + 2) && cdiv(N, 32) < 0x8000) {
      int* flag = (int*)temp;
      int* ticket = flag + B;
      float* E = temp + 2 * (size_t)B;
      hipLaunchKernelGGL(fps_check_e_k, dim3(cdiv(M, 64), B), dim3(256), 0, s, N, M, xyz, flag, ticket, E);
      hipLaunchKernelGGL(fps_check_test_k, dim3(cdiv(N, 32), B), dim3(256), 0, s, N, M, xyz, E, idx, flag, ticket);
    } else {
      hipLaunchKernelGGL(fps_ordered_check_k, dim3(B), dim3(1024), 0, s, N, M, xyz, idx, (int*)temp);
    }
    skip = (const int*)temp;
  }
  // large clouds with the (B, N) scratch at hand: Hilbert-cell order + the exact box-pruned kernel
  static const int prune_on = [] { const char* v = getenv("DEMF_FPS_PRUNE"); return v ? atoi(v) : 1; }();
  if (prune_on && skip == nullptr && temp != nullptr && bs == 1024 && N >= 4096 && N <= 20 * 1024 && M >= 64 &&
      M <= N) {
    int* perm = (int*)temp;
    hipLaunchKernelGGL(fps_sort_k, dim3(B), dim3(1024), 0, s, N, xyz, perm);
    // per-pair cached maxima instead of the 20-slot re-search (round 6).  Measured on MI355X (tools/fps_micro.py):
    // 20 000 -> 2 048: 2 364 vs 2 420 cycles per round; 16 384 -> 2 048: 2 234 vs 2 184 (fewer slots: the search it
    // replaces is shorter) - so the default takes it above 16 slots per lane only.  DEMF_FPS_PAIR = 0 / 1 forces.
    static const int pair_env = [] { const char* v = getenv("DEMF_FPS_PAIR"); return v ? atoi(v) : -1; }();
    const bool pair_on = pair_env < 0 ? ppt > 16 : pair_env != 0;
    if (pair_on) {
#define PAIR(P) hipLaunchKernelGGL((fps_pair_kernel<P>), dim3(B), dim3(1024), 0, s, N, M, xyz, perm, idx)
#define PAIR8(P) hipLaunchKernelGGL((fps_pair_kernel<P, 8>), dim3(B), dim3(512), 0, s, N, M, xyz, perm, idx)
      static const int pwaves = [] { const char* v = getenv("DEMF_FPS_PAIR_WAVES"); return v ? atoi(v) : 16; }();   // A/B
      if (pwaves == 8 && ppt > 8) {
        if (ppt <= 12) PAIR8(24); else if (ppt <= 16) PAIR8(32); else PAIR8(40);
      } else if (ppt <= 4) PAIR(4); else if (ppt <= 8) PAIR(8); else if (ppt <= 12) PAIR(12);
      else if (ppt <= 16) PAIR(16); else PAIR(20);
#undef PAIR8
#undef PAIR
      return check_launch("fps_pair");
    }
#define PRUNE(P) hipLaunchKernelGGL((fps_prune_kernel<P>), dim3(B), dim3(1024), 0, s, N, M, xyz, perm, idx)
#define PRUNE8(P) hipLaunchKernelGGL((fps_prune_kernel<P, 8>), dim3(B), dim3(512), 0, s, N, M, xyz, perm, idx)
    // A/B switch, default 16: 8 waves x 40 slots is bit-identical but SLOWER (20 000 -> 2 048: 2 643 vs 2 058 us) -
    // the round's critical path is the wave that has to update and search again, and that wave's work doubles
    static const int waves = [] { const char* v = getenv("DEMF_FPS_WAVES"); return v ? atoi(v) : 16; }();
    if (waves == 8 && ppt > 8) {                            // 8 waves x up to 40 slots per lane
      if (ppt <= 12) PRUNE8(24); else if (ppt <= 16) PRUNE8(32); else PRUNE8(40);
    } else if (ppt <= 4) PRUNE(4); else if (ppt <= 8) PRUNE(8); else if (ppt <= 12) PRUNE(12);
    else if (ppt <= 16) PRUNE(16); else PRUNE(20);          // 24 slots + their keys spill
#undef PRUNE
#undef PRUNE8
    return check_launch("fps_prune");
  }
  bool done = true;
  if (bs == 1024) {
    if (ppt <= 2) launch_reg<1024, 2>(B, N, M, xyz, idx, skip, s);
    else if (ppt <= 4) launch_reg<1024, 4>(B, N, M, xyz, idx, skip, s);
    else if (ppt <= 6) launch_reg<1024, 6>(B, N, M, xyz, idx, skip, s);
    else if (ppt <= 8) launch_reg<1024, 8>(B, N, M, xyz, idx, skip, s);
    else if (ppt <= 12) launch_reg<1024, 12>(B, N, M, xyz, idx, skip, s);
    else if (ppt <= 16) launch_reg<1024, 16>(B, N, M, xyz, idx, skip, s);
    else if (ppt <= 20) launch_reg<1024, 20>(B, N, M, xyz, idx, skip, s);
    else if (ppt <= 24) launch_reg<1024, 24>(B, N, M, xyz, idx, skip, s);
    else done = false;
  } else if (bs == 512) {
    launch_reg<512, 2>(B, N, M, xyz, idx, skip, s);
  } else if (bs == 256) {
    launch_reg<256, 2>(B, N, M, xyz, idx, skip, s);
  } else if (bs == 128) {
    launch_reg<128, 2>(B, N, M, xyz, idx, skip, s);
  } else if (bs == 64) {
    launch_reg<64, 2>(B, N, M, xyz, idx, skip, s);
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
