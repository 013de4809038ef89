// First shared-MLP layer of a set-abstraction level WITHOUT the grouped tensor.
//
// Upstream (PointSAModule via build_sa_module, class_agnostic_vote_head.py:383,455 and the
// backbone) gathers ns neighbours per centre into rows [(xyz_j - centre)/radius (3) | feat_j (C)]
// and feeds them to a 1x1 convolution W (C1, 3+C).  The convolution is linear, so
//
//     y[m,s,:] = feat[idx[m,s]] . Wf^T + rel(m,s) . Wx^T = U[idx[m,s], :] + rel(m,s) . Wx^T
//
// with U = feat . Wf^T computed ONCE per source point (N rows) instead of once per neighbour
// (M*ns rows, 16 x more at SA2).  The forward kernel is then a row gather of U plus three fmas
// per element and the BN statistics of y; the backward kernel sums the (BN-backward transformed)
// gradient rows of every source point through the inverse neighbour lists (demf_invert_index) ->
// dU, from which the weight / feature gradients are two N-row GEMMs, and reduces the 3 x C1 xyz
// weight gradient on the way.  Both are HBM/L2 byte movers: y written once (fwd), dZ and y read
// once (bwd); the (M*ns, 3+C) grouped rows and their gradient never exist.
#include "common.h"
#include "bn_fin.h"

namespace demf {

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// A group of LPR = C1/4 lanes owns one output row at a time (4 columns per lane).  Indices and
// relative coordinates are fetched for LPR consecutive rows at once - lane t of the group owns
// row t of the chunk - and handed round with ds_bpermute, so the row loop has no dependent index
// load, one division per row instead of one per lane, and its U-row gathers are all in flight
// together.
template <int LPR>
__device__ __forceinline__ float grp_bcast(float v, int src) { return __shfl(v, src, LPR); }
template <int LPR>
__device__ __forceinline__ int grp_bcast(int v, int src) { return __shfl(v, src, LPR); }

// coordinate weights of 4 consecutive output channels c..c+3 for input column j: from the transposed
// (3, C1) copy (wld == 0) or in place from the layer's weight W (C1 x wld row-major, columns 0..2)
__device__ __forceinline__ float4 load_wx(const float* __restrict__ Wx, int wld, int C1, int c, int j) {
  if (wld == 0) return *reinterpret_cast<const float4*>(Wx + j * C1 + c);
  return make_float4(Wx[(size_t)c * wld + j], Wx[(size_t)(c + 1) * wld + j],
                     Wx[(size_t)(c + 2) * wld + j], Wx[(size_t)(c + 3) * wld + j]);
}

template <int LPR>
__global__ __launch_bounds__(512) void group_first_fwd_k(
    int N, int M, int ns, float div, const float* __restrict__ xyz,
    const float* __restrict__ center, const int* __restrict__ idx, const float* __restrict__ U,
    const float* __restrict__ Wx, int wld, float* __restrict__ Y, double* __restrict__ stats,
    long long rows, BnFin fin) {
  constexpr int C1 = LPR * 4;
  constexpr int GPB = 512 / LPR;
  constexpr int UNR = 8;
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const int c = sub * 4;
  const float4 w0 = load_wx(Wx, wld, C1, c, 0);
  const float4 w1 = load_wx(Wx, wld, C1, c, 1);
  const float4 w2 = load_wx(Wx, wld, C1, c, 2);
  float4 s = f4_zero(), q = f4_zero();
  const long long chunks = (rows + LPR - 1) / LPR;
  for (long long ch = (long long)blockIdx.x * GPB + grp; ch < chunks;
       ch += (long long)gridDim.x * GPB) {
    const long long row0 = ch * LPR;
    const int n = rows - row0 < LPR ? (int)(rows - row0) : LPR;
    // this lane's row of the chunk: source row in U and relative coordinates
    int my_src = 0;
    float mx = 0.f, my = 0.f, mz = 0.f;
    if (sub < n) {
      const long long row = row0 + sub;
      const long long bm = row / ns;
      const int b = (int)(bm / M);
      my_src = b * N + idx[row];
      const float* p = xyz + (size_t)my_src * 3;
      const float* o = center + bm * 3;
      mx = (p[0] - o[0]) / div;  // upstream: grouped_xyz /= max_radius
      my = (p[1] - o[1]) / div;
      mz = (p[2] - o[2]) / div;
    }
    for (int k0 = 0; k0 < n; k0 += UNR) {
      float4 u[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const int src = grp_bcast<LPR>(my_src, k0 + k);
        u[k] = f4_zero();
        if (k0 + k < n) u[k] = *reinterpret_cast<const float4*>(U + (size_t)src * C1 + c);
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const float rx = grp_bcast<LPR>(mx, k0 + k), ry = grp_bcast<LPR>(my, k0 + k),
                    rz = grp_bcast<LPR>(mz, k0 + k);
        if (k0 + k < n) {
          float4 y;
#define GF_Y(m) y.m = __builtin_fmaf(rz, w2.m, __builtin_fmaf(ry, w1.m, __builtin_fmaf(rx, w0.m, u[k].m)));
          GF_Y(x) GF_Y(y) GF_Y(z) GF_Y(w)
#undef GF_Y
          *reinterpret_cast<float4*>(Y + (row0 + k0 + k) * C1 + c) = y;
          s.x += y.x; s.y += y.y; s.z += y.z; s.w += y.w;
          q.x = __builtin_fmaf(y.x, y.x, q.x); q.y = __builtin_fmaf(y.y, y.y, q.y);
          q.z = __builtin_fmaf(y.z, y.z, q.z); q.w = __builtin_fmaf(y.w, y.w, q.w);
        }
      }
    }
  }
  if (stats == nullptr && fin.ticket == nullptr) return;
  __shared__ float4 red[2][GPB][LPR];
  __shared__ int s_last;
  red[0][grp][sub] = s;
  red[1][grp][sub] = q;
  __syncthreads();
  // 2*C1 column totals, one thread each (fp64 from here on).  With a counter set the launch's last workgroup does the
  // BatchNorm bookkeeping itself (no finalize launch) and the partial sums go to this workgroup's copy of a replicated
  // accumulator (csrc/bn_fin.h: 512 workgroups adding to the same 2*C1 doubles queue for ~10 us).
  double* dst = fin.ticket != nullptr ? repl_copy(fin.racc, C1, (int)blockIdx.x) : stats;
  for (int t = threadIdx.x; t < 2 * C1; t += 512) {
    const int which = t / C1, col = t - which * C1;
    const float* base = reinterpret_cast<const float*>(&red[which][0][0]) + col;
    double tot = 0.0;
#pragma unroll 4
    for (int g = 0; g < GPB; ++g) tot += (double)base[g * C1];
    atomicAdd(dst + which * C1 + col, tot);
  }
  if (fin.ticket == nullptr) return;
  sync_drained();
  if (threadIdx.x == 0) s_last = last_workgroup(fin.ticket, (int)gridDim.x, (int)blockIdx.x);
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0 && fin.nbt != nullptr) *fin.nbt += 1;
  for (int c = threadIdx.x; c < C1; c += 512) {
    const double s1 = repl_take(fin.racc, C1, c), s2 = repl_take(fin.racc, C1, C1 + c);
    bn_finalize_channel(c, C1, fin.count, s1, s2, fin.gamma, fin.beta, fin.eps, fin.momentum, fin.rmean, fin.rvar,
                        fin.ss, fin.mi, fin.conv_bias);
  }
}

// One group of LPR lanes per source point: sums dY over the rows that gathered the point.  The
// list entries (and the centre of each entry's row) are fetched LPR at a time, one per lane, and
// broadcast inside the group.
// An atomic add costs ~20 ns at its memory channel and the adds to one cache line queue up there: W workgroups adding
// to the same few lines end W * 20 ns after the first (tools/ubench/atomic_flush.cpp: 1024 workgroups x 384 doubles =
// 23 us, 256 = 7 us).  The partial sums of a launch therefore go to GF_REPL copies, workgroup w to copy w % GF_REPL.
constexpr int GF_REPL = 8;

template <int LPR>
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, LPR);
  return v;
}

// XG: the coordinates carry a gradient too (the vote aggregation: vote points are predicted):
// d rel = dY . Wx^T per row -> + to the source point (summed here, written once), - to the centre
// (3 atomics per row).
// RC: y is not read back but formed again from the point's row of U and the relative coordinates, with the forward
// kernel's own expression (same division, same fma chain: bit-identical), which halves the bytes of the launch.
template <int LPR, bool XG, bool RC>
__global__ __launch_bounds__(256) void group_first_bwd_k(
    int N, int M, int ns, int ns_shift, float inv_div, float div, const float* __restrict__ xyz,
    const float* __restrict__ center, const float* __restrict__ G, const float* __restrict__ Yl,
    const float* __restrict__ U,
    const float* __restrict__ vec, const int* __restrict__ off, const int* __restrict__ rows_,
    float* __restrict__ dU, float* __restrict__ dWx, int dwld, long long points,
    const float* __restrict__ Wx, int wld, float* __restrict__ dxyz, float* __restrict__ dcenter,
    double* __restrict__ wacc, int* __restrict__ ticket) {
  constexpr int C1 = LPR * 4;
  constexpr int GPB = 256 / LPR;
  constexpr int UNR = RC ? 16 : 8;
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const int c = sub * 4;
  const int E = M * ns;
  // dY = gi*dZ + (a*y + b), dZ = G where y*scale+shift > 0   (vec = demf_bn_bwd_vectors)
  const float4 sc = *reinterpret_cast<const float4*>(vec + c);
  const float4 sh = *reinterpret_cast<const float4*>(vec + C1 + c);
  const float4 gi = *reinterpret_cast<const float4*>(vec + 2 * C1 + c);
  const float4 va = *reinterpret_cast<const float4*>(vec + 3 * C1 + c);
  const float4 vb = *reinterpret_cast<const float4*>(vec + 4 * C1 + c);
  float4 wa0 = f4_zero(), wa1 = f4_zero(), wa2 = f4_zero();  // sum (p - q)_k * dY, scaled at the end
  float4 wx0 = f4_zero(), wx1 = f4_zero(), wx2 = f4_zero();
  if constexpr (XG || RC) {
    wx0 = load_wx(Wx, wld, C1, c, 0);
    wx1 = load_wx(Wx, wld, C1, c, 1);
    wx2 = load_wx(Wx, wld, C1, c, 2);
  }
  for (long long pt = (long long)blockIdx.x * GPB + grp; pt < points;
       pt += (long long)gridDim.x * GPB) {
    const int b = (int)(pt / N), j = (int)(pt - (long long)b * N);
    const int* o = off + (size_t)b * (N + 1) + j;
    const int e0 = o[0], e1 = o[1];
    const int* r = rows_ + (size_t)b * E;
    const size_t rbase = (size_t)b * E;
    const float* cb = center + (size_t)b * M * 3;
    const float px = xyz[pt * 3], py = xyz[pt * 3 + 1], pz = xyz[pt * 3 + 2];
    float4 acc = f4_zero();
    float gx = 0.f, gy = 0.f, gz = 0.f;
    float4 u = f4_zero();
    if constexpr (RC) {
      if (e1 > e0) u = *reinterpret_cast<const float4*>(U + pt * C1 + c);
    }
    for (int eb = e0; eb < e1; eb += LPR) {
      const int n = e1 - eb < LPR ? e1 - eb : LPR;
      int my_row = 0, my_m = 0;
      float mx = 0.f, my = 0.f, mz = 0.f;
      if (sub < n) {
        my_row = r[eb + sub];
        const int m = ns_shift >= 0 ? my_row >> ns_shift : my_row / ns;
        my_m = m;
        const float* q = cb + (size_t)m * 3;
        mx = px - q[0];
        my = py - q[1];
        mz = pz - q[2];
        if constexpr (RC) {   // the forward's relative coordinates (grouped_xyz / max_radius); dWx takes them as they are
          mx /= div;
          my /= div;
          mz /= div;
        }
      }
      for (int k0 = 0; k0 < n; k0 += UNR) {
        float4 g[UNR], y[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
          const int row = grp_bcast<LPR>(my_row, k0 + k);
          g[k] = f4_zero();
          y[k] = f4_zero();
          if (k0 + k < n) {
            g[k] = *reinterpret_cast<const float4*>(G + (rbase + row) * C1 + c);
            if constexpr (!RC) y[k] = *reinterpret_cast<const float4*>(Yl + (rbase + row) * C1 + c);
          }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
          const float rx = grp_bcast<LPR>(mx, k0 + k), ry = grp_bcast<LPR>(my, k0 + k),
                      rz = grp_bcast<LPR>(mz, k0 + k);
          const int mrow = XG ? grp_bcast<LPR>(my_m, k0 + k) : 0;
          if (k0 + k < n) {
            float4 d;
            if constexpr (RC) {
#define GF_Y(m) y[k].m = __builtin_fmaf(rz, wx2.m, __builtin_fmaf(ry, wx1.m, __builtin_fmaf(rx, wx0.m, u.m)));
              GF_Y(x) GF_Y(y) GF_Y(z) GF_Y(w)
#undef GF_Y
            }
#define GF_DY(m)                                                                       \
            {                                                                          \
              const float dz = __builtin_fmaf(y[k].m, sc.m, sh.m) > 0.f ? g[k].m : 0.f; \
              d.m = __builtin_fmaf(gi.m, dz, __builtin_fmaf(va.m, y[k].m, vb.m));      \
              acc.m += d.m;                                                            \
              wa0.m = __builtin_fmaf(rx, d.m, wa0.m);                                  \
              wa1.m = __builtin_fmaf(ry, d.m, wa1.m);                                  \
              wa2.m = __builtin_fmaf(rz, d.m, wa2.m);                                  \
            }
            GF_DY(x) GF_DY(y) GF_DY(z) GF_DY(w)
#undef GF_DY
            if constexpr (XG) {
              const float t0 = grp_sum<LPR>(d.x * wx0.x + d.y * wx0.y + d.z * wx0.z + d.w * wx0.w);
              const float t1 = grp_sum<LPR>(d.x * wx1.x + d.y * wx1.y + d.z * wx1.z + d.w * wx1.w);
              const float t2 = grp_sum<LPR>(d.x * wx2.x + d.y * wx2.y + d.z * wx2.z + d.w * wx2.w);
              gx += t0; gy += t1; gz += t2;
              if (sub == 0) {
                float* dc = dcenter + ((size_t)b * M + mrow) * 3;
                atomicAdd(dc, -t0 * inv_div);
                atomicAdd(dc + 1, -t1 * inv_div);
                atomicAdd(dc + 2, -t2 * inv_div);
              }
            }
          }
        }
      }
    }
    if constexpr (XG) {
      if (sub == 0) {
        dxyz[pt * 3] = gx * inv_div;
        dxyz[pt * 3 + 1] = gy * inv_div;
        dxyz[pt * 3 + 2] = gz * inv_div;
      }
    }
    *reinterpret_cast<float4*>(dU + pt * C1 + c) = acc;
  }
  __shared__ float4 red[3][GPB][LPR];
  red[0][grp][sub] = wa0;
  red[1][grp][sub] = wa1;
  red[2][grp][sub] = wa2;
  __syncthreads();
  for (int t = threadIdx.x; t < 3 * C1; t += 256) {
    const int which = t / C1, col = t - which * C1;
    const float* base = reinterpret_cast<const float*>(&red[which][0][0]) + col;
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < GPB; ++g) tot += base[g * C1];
    tot = RC ? tot : tot * inv_div;
    // In place in the layer's weight gradient (dwld > 0) the 3 x C1 targets lie dwld floats apart: one memory-side
    // atomic transaction per LANE and per workgroup, ~35 us of a launch that moves 16 MB.  With an accumulator the
    // workgroups add into 3*C1 CONTIGUOUS doubles (a few cache lines per wave) and the last one moves the totals.
    if (wacc != nullptr) atomicAdd(wacc + (blockIdx.x % GF_REPL) * 3 * C1 + t, (double)tot);
    else atomicAdd(dwld == 0 ? dWx + which * C1 + col : dWx + (size_t)col * dwld + which, tot);
  }
  if (wacc == nullptr) return;
  sync_drained();
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = last_workgroup(ticket, (int)gridDim.x, (int)blockIdx.x);
  __syncthreads();
  if (!s_last) return;
  for (int t = threadIdx.x; t < 3 * C1; t += 256) {
    const int which = t / C1, col = t - which * C1;
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < GF_REPL; ++r)
      v += __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(wacc + r * 3 * C1 + t), 0ull));
    float* dst = dwld == 0 ? dWx + which * C1 + col : dWx + (size_t)col * dwld + which;
    *dst += (float)v;       // (this launch is the only writer of the coordinate columns)
  }
}

}  // namespace demf

using namespace demf;

extern "C" int demf_group_first_fwd(int B, int N, int M, int ns, int C1, float radius,
                                    int normalize_xyz, const float* xyz, const float* center,
                                    const int* idx, const float* U, const float* Wx, int w_ld,
                                    float* Y, double* stats, const float* gamma, const float* beta, float eps,
                                    float momentum, float* running_mean, float* running_var,
                                    long long* num_batches_tracked, float* scale_shift, float* mean_invstd,
                                    const float* conv_bias, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && M >= 0 && ns >= 1 && (C1 == 64 || C1 == 128 || C1 == 256),
               "group_first_fwd: bad sizes B=%d N=%d M=%d ns=%d C1=%d", B, N, M, ns, C1);
  if (B == 0 || M == 0) return DEMF_OK;
  DEMF_REQUIRE(xyz && center && idx && U && Wx && Y, "group_first_fwd: null pointer");
  const long long rows = (long long)B * M * ns;
  const float div = normalize_xyz ? radius : 1.0f;
  hipStream_t s = (hipStream_t)stream;
  const int lpr = C1 / 4, gpb = 512 / lpr;
  const long long chunks = (rows + lpr - 1) / lpr;
  long long blocks = (chunks + gpb - 1) / gpb;
  // (one fp64 add per column per block; with the statistics in replicated accumulators the tail no longer argues for
  //  few blocks - resident step 256 / 512 / 1024 blocks: 4.44 / 4.42 / 4.39 ms)
  static const int fcap = getenv("DEMF_GF_FWD_BLOCKS") ? atoi(getenv("DEMF_GF_FWD_BLOCKS")) : 1024;
  if (blocks > fcap) blocks = fcap;
  const dim3 grid((unsigned)blocks);
  // train-mode BatchNorm bookkeeping (demf_bn_finalize's arguments): by the launch's last workgroup when a counter set
  // and an accumulator block are to be had, as a launch behind this one otherwise
  BnFin fin{};
  const bool want_fin = scale_shift != nullptr;
  if (want_fin) {
    DEMF_REQUIRE(stats && gamma && beta && mean_invstd, "group_first_fwd: the BatchNorm bookkeeping needs stats, gamma, "
                 "beta, scale_shift and mean_invstd");
    static const bool fin_off = getenv("DEMF_GF_FIN") && atoi(getenv("DEMF_GF_FIN")) == 0;   // A/B switch
    int* ticket = fin_off ? nullptr : sched_slot();
    double* racc = ticket ? accum_slot() : nullptr;
    if (ticket != nullptr && racc != nullptr)
      fin = BnFin{(double)rows, gamma, beta, conv_bias, eps, momentum, running_mean, running_var, num_batches_tracked,
                  scale_shift, mean_invstd, ticket, racc};
  }
#define GF_FWD(L)                                                                          \
  hipLaunchKernelGGL(group_first_fwd_k<L>, grid, dim3(512), 0, s, N, M, ns, div, xyz, center, \
                     idx, U, Wx, w_ld, Y, stats, rows, fin)
  if (lpr == 64) GF_FWD(64);
  else if (lpr == 32) GF_FWD(32);
  else GF_FWD(16);
#undef GF_FWD
  if (int e = check_launch("group_first_fwd")) return e;
  if (want_fin && fin.ticket == nullptr)
    return demf_bn_finalize(C1, rows, stats, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                            scale_shift, mean_invstd, conv_bias, stream);
  return DEMF_OK;
}

extern "C" int demf_group_first_bwd(int B, int N, int M, int ns, int C1, float radius,
                                    int normalize_xyz, const float* xyz, const float* center,
                                    const float* G, const float* Y, const float* U, const float* vec6,
                                    const int* inv_off, const int* inv_rows, float* dU,
                                    float* dWx, int dw_ld, const float* Wx, int w_ld, float* dxyz,
                                    float* dcenter, double* wacc, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && M >= 0 && ns >= 1 && (C1 == 64 || C1 == 128 || C1 == 256),
               "group_first_bwd: bad sizes B=%d N=%d M=%d ns=%d C1=%d", B, N, M, ns, C1);
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(xyz && center && G && (Y || U) && vec6 && inv_off && inv_rows && dU && dWx,
               "group_first_bwd: null pointer");
  DEMF_REQUIRE(U == nullptr || Wx != nullptr, "group_first_bwd: forming y again from U needs the layer's weight");
  static const bool rc_off = getenv("DEMF_GF_RECOMPUTE") && atoi(getenv("DEMF_GF_RECOMPUTE")) == 0;   // A/B switch
  const bool rc = U != nullptr && !(rc_off && Y != nullptr);
  const float div = normalize_xyz ? radius : 1.0f;
  const long long points = (long long)B * N;
  const float inv_div = normalize_xyz ? 1.0f / radius : 1.0f;
  int ns_shift = -1;
  if ((ns & (ns - 1)) == 0) { ns_shift = 0; while ((1 << ns_shift) < ns) ++ns_shift; }
  hipStream_t s = (hipStream_t)stream;
  const int lpr = C1 / 4, gpb = 256 / lpr;
  long long blocks = (points + gpb - 1) / gpb;
  static const bool wacc_off = getenv("DEMF_GF_WACC") && atoi(getenv("DEMF_GF_WACC")) == 0;   // A/B switch
  int* ticket = wacc && !wacc_off ? sched_slot() : nullptr;
  if (ticket == nullptr) wacc = nullptr;          // (no counter set: every workgroup adds to dWx itself)
  // (one group per source point and pass: more workgroups = more independent load chains in flight; the partial dWx of
  //  a workgroup go to one of GF_REPL copies of the accumulator, so the cap no longer trades against the atomic tail)
  static const int bcap = getenv("DEMF_GF_BWD_BLOCKS") ? atoi(getenv("DEMF_GF_BWD_BLOCKS")) : 0;
  const int cap = bcap > 0 ? bcap : (dxyz == nullptr && wacc != nullptr ? 2048 : 1024);
  if (blocks > cap) blocks = cap;
  const dim3 grid((unsigned)blocks);
  DEMF_REQUIRE((dxyz == nullptr) == (dcenter == nullptr) && (dxyz == nullptr || Wx != nullptr),
               "group_first_bwd: dxyz, dcenter (and Wx) must be given together");
#define GF_BWD_(L, XG_, RC_)                                                                \
  hipLaunchKernelGGL((group_first_bwd_k<L, XG_, RC_>), grid, dim3(256), 0, s, N, M, ns, ns_shift, inv_div, div, xyz,   \
                     center, G, Y, U, vec6, inv_off, inv_rows, dU, dWx, dw_ld, points, Wx, w_ld, dxyz, dcenter, wacc, ticket)
#define GF_BWD(L)                                                                          \
  do {                                                                                     \
    if (dxyz) { if (rc) GF_BWD_(L, true, true); else GF_BWD_(L, true, false); }            \
    else      { if (rc) GF_BWD_(L, false, true); else GF_BWD_(L, false, false); }          \
  } while (0)
  if (lpr == 64) GF_BWD(64);
  else if (lpr == 32) GF_BWD(32);
  else GF_BWD(16);
#undef GF_BWD_
#undef GF_BWD
  return check_launch("group_first_bwd");
}
