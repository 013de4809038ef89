// three_nn / three_interpolate for gfx950.
//
// Stand in for mmdet3d.ops three_nn / three_interpolate inside the two
// PointFPModules of the backbone (configs/demf/demf_votenet.py:56), reached via
// DeMFVoteNet.extract_pts_feat (demf/modeling/detectors/demfnet.py:151-152).
//
// three_nn: one lane per target point; the (<= a few hundred) source points are
// staged once per workgroup in LDS as SoA and read as wave-uniform broadcasts.
// Strict '<' insertion keeps the upstream "earliest source wins ties" order.
// three_interpolate: 3-tap weighted row gather; HBM/L2-bound.
#include "common.h"
#include <stdlib.h>

namespace demf {

constexpr int NN_TILE = 2048;

__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m,
                                                       const float* __restrict__ target,
                                                       const float* __restrict__ source,
                                                       float* __restrict__ dist2_out,
                                                       int* __restrict__ idx_out,
                                                       float* __restrict__ weight_out) {
  __shared__ float sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
  const int b = blockIdx.y;
  target += (size_t)b * n * 3;
  source += (size_t)b * m * 3;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = t < n;
  const float ux = ok ? target[3 * t + 0] : 0.f;
  const float uy = ok ? target[3 * t + 1] : 0.f;
  const float uz = ok ? target[3 * t + 2] : 0.f;
  const float INF = __builtin_inff();  // upstream initialises with 1e40f == +inf
  float b1 = INF, b2 = INF, b3 = INF;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int k0 = 0; k0 < m; k0 += NN_TILE) {
    const int cnt = min(NN_TILE, m - k0);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += blockDim.x) {
      sx[k] = source[3 * (k0 + k) + 0];
      sy[k] = source[3 * (k0 + k) + 1];
      sz[k] = source[3 * (k0 + k) + 2];
    }
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const float d = dist2(ux - sx[k], uy - sy[k], uz - sz[k]);
      const int kk = k0 + k;
      if (d < b1) {
        b3 = b2; i3 = i2;
        b2 = b1; i2 = i1;
        b1 = d;  i1 = kk;
      } else if (d < b2) {
        b3 = b2; i3 = i2;
        b2 = d;  i2 = kk;
      } else if (d < b3) {
        b3 = d;  i3 = kk;
      }
    }
  }
  if (ok) {
    float* dd = dist2_out + ((size_t)b * n + t) * 3;
    int* ii = idx_out + ((size_t)b * n + t) * 3;
    ii[0] = i1; ii[1] = i2; ii[2] = i3;
    if (weight_out == nullptr) {
      dd[0] = b1; dd[1] = b2; dd[2] = b3;
    } else {
      // PointFPModule.forward: dist = sqrt(dist2); r = 1 / (dist + 1e-8); weight = r / sum(r) -
      // the five element-wise launches of the reference's expression, in its operation order
      const float d1 = sqrtf(b1), d2 = sqrtf(b2), d3 = sqrtf(b3);
      const float r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f), r3 = 1.0f / (d3 + 1e-8f);
      const float sum = (r1 + r2) + r3;
      float* ww = weight_out + ((size_t)b * n + t) * 3;
      dd[0] = d1; dd[1] = d2; dd[2] = d3;
      ww[0] = r1 / sum; ww[1] = r2 / sum; ww[2] = r3 / sum;
    }
  }
}

// The same search with EIGHT lanes per target (round 5).  three_nn_kernel walks the m sources with one thread per
// target - 1 024 x 512 pairs per scene on four waves, a branchy dependent top-3 insertion per pair: 40-67 us per call
// on the serial pre-pass chain.  Here lane c of a group takes the sources k = c, c + 8, ... (consecutive LDS words
// across the group, the 8 groups of a wave broadcast), keeps its own three smallest, and the group selects the three
// smallest of its 24 candidates by the key (distance bits << 32 | index): squared distances are >= +0, so their
// bit patterns order like the values, and the index in the low word reproduces the sequential scan's tie rule
// (strict <: the earlier source wins).  Bit-identical results; needs m >= 64.
__global__ __launch_bounds__(256) void three_nn_split_kernel(int n, int m,
                                                             const float* __restrict__ target,
                                                             const float* __restrict__ source,
                                                             float* __restrict__ dist2_out,
                                                             int* __restrict__ idx_out,
                                                             float* __restrict__ weight_out) {
  __shared__ float sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
  const int b = blockIdx.y;
  target += (size_t)b * n * 3;
  source += (size_t)b * m * 3;
  const int c = threadIdx.x & 7;
  const int t = blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool ok = t < n;
  const int tt = ok ? t : n - 1;
  const float ux = target[3 * tt + 0], uy = target[3 * tt + 1], uz = target[3 * tt + 2];
  const unsigned long long KINF = ((unsigned long long)0x7f800000u << 32);     // +inf, index 0: the scan's initial state
  unsigned long long k1 = KINF, k2 = KINF, k3 = KINF;
  for (int k0 = 0; k0 < m; k0 += NN_TILE) {
    const int cnt = min(NN_TILE, m - k0);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += blockDim.x) {
      sx[k] = source[3 * (k0 + k) + 0];
      sy[k] = source[3 * (k0 + k) + 1];
      sz[k] = source[3 * (k0 + k) + 2];
    }
    __syncthreads();
    for (int k = c; k < cnt; k += 8) {
      const float d = dist2(ux - sx[k], uy - sy[k], uz - sz[k]);
      const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(k0 + k);
      // (d < b) on values == (key < kb) on keys for finite d >= 0 with increasing indices; NaN (sign clear) sorts last
      if (d == d) {
        if (key < k1) { k3 = k2; k2 = k1; k1 = key; }
        else if (key < k2) { k3 = k2; k2 = key; }
        else if (key < k3) { k3 = key; }
      }
    }
  }
  // the group's three smallest keys: three rounds of (minimum over the 8 heads, the owner pops)
  unsigned long long out[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    unsigned long long mn = k1;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const unsigned long long other = __shfl_xor(mn, o);
      mn = other < mn ? other : mn;
    }
    out[r] = mn;
    if (k1 == mn && mn != KINF) { k1 = k2; k2 = k3; k3 = KINF; }
  }
  if (ok && c == 0) {
    const float b1 = __uint_as_float((unsigned)(out[0] >> 32)), b2 = __uint_as_float((unsigned)(out[1] >> 32)),
                b3 = __uint_as_float((unsigned)(out[2] >> 32));
    float* dd = dist2_out + ((size_t)b * n + t) * 3;
    int* ii = idx_out + ((size_t)b * n + t) * 3;
    ii[0] = (int)(unsigned)out[0]; ii[1] = (int)(unsigned)out[1]; ii[2] = (int)(unsigned)out[2];
    if (weight_out == nullptr) {
      dd[0] = b1; dd[1] = b2; dd[2] = b3;
    } else {
      const float d1 = sqrtf(b1), d2 = sqrtf(b2), d3 = sqrtf(b3);
      const float r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f), r3 = 1.0f / (d3 + 1e-8f);
      const float sum = (r1 + r2) + r3;
      float* ww = weight_out + ((size_t)b * n + t) * 3;
      dd[0] = d1; dd[1] = d2; dd[2] = d3;
      ww[0] = r1 / sum; ww[1] = r2 / sum; ww[2] = r3 / sum;
    }
  }
}

// out = fma(w2, f2, fma(w0, f0, w1*f1)) : pinned contraction, see DESIGN.md
__device__ __forceinline__ float interp3(float w0, float f0, float w1, float f1, float w2,
                                         float f2) {
  return __builtin_fmaf(w2, f2, __builtin_fmaf(w0, f0, w1 * f1));
}

template <int CT>
__global__ __launch_bounds__(256) void interp_cm_fwd(int C, int m, int n,
                                                     const float* __restrict__ feat,
                                                     const int* __restrict__ idx,
                                                     const float* __restrict__ w,
                                                     float* __restrict__ out) {
  const int b = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = blockIdx.y * CT;
  if (t >= n) return;
  const int* ii = idx + ((size_t)b * n + t) * 3;
  const float* ww = w + ((size_t)b * n + t) * 3;
  const int i0 = ii[0], i1 = ii[1], i2 = ii[2];
  const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) {
    const int c = c0 + cc;
    if (c < C) {
      const float* f = feat + ((size_t)b * C + c) * m;
      out[((size_t)b * C + c) * n + t] = interp3(w0, f[i0], w1, f[i1], w2, f[i2]);
    }
  }
}

template <int CT>
__global__ __launch_bounds__(256) void interp_cm_bwd(int C, int n, int m,
                                                     const float* __restrict__ gout,
                                                     const int* __restrict__ idx,
                                                     const float* __restrict__ w,
                                                     float* __restrict__ gfeat) {
  const int b = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = blockIdx.y * CT;
  if (t >= n) return;
  const int* ii = idx + ((size_t)b * n + t) * 3;
  const float* ww = w + ((size_t)b * n + t) * 3;
  const int i0 = ii[0], i1 = ii[1], i2 = ii[2];
  const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) {
    const int c = c0 + cc;
    if (c < C) {
      const float g = gout[((size_t)b * C + c) * n + t];
      float* f = gfeat + ((size_t)b * C + c) * m;
      atomicAdd(f + i0, g * w0);
      atomicAdd(f + i1, g * w1);
      atomicAdd(f + i2, g * w2);
    }
  }
}

// point-major: one wave per target row
__global__ __launch_bounds__(256) void interp_cl_fwd(int m, int n, int C, int ldo, int col0,
                                                     const float* __restrict__ feat,
                                                     const int* __restrict__ idx,
                                                     const float* __restrict__ w,
                                                     float* __restrict__ out,
                                                     long long rows,
                                                     const float* __restrict__ skip, int Cs) {
  // skip (rows, Cs), optional: the target level's own features, copied behind the interpolated
  // columns - the torch.cat of PointFPModule.forward written by the same launch
  const int lane = threadIdx.x & 63;
  long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
  for (; row < rows; row += stride) {
    const int b = (int)(row / n);
    const int* ii = idx + row * 3;
    const float* ww = w + row * 3;
    const float* f0 = feat + ((size_t)b * m + ii[0]) * C;
    const float* f1 = feat + ((size_t)b * m + ii[1]) * C;
    const float* f2 = feat + ((size_t)b * m + ii[2]) * C;
    const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
    float* o = out + row * ldo + col0;
    for (int c = lane; c < C; c += 64) o[c] = interp3(w0, f0[c], w1, f1[c], w2, f2[c]);
    if (skip != nullptr) {
      const float* sk = skip + row * Cs;
      for (int c = lane; c < Cs; c += 64) o[C + c] = sk[c];
    }
  }
}

__global__ __launch_bounds__(256) void interp_cl_bwd(int m, int n, int C, int ldo, int col0,
                                                     const float* __restrict__ gout,
                                                     const int* __restrict__ idx,
                                                     const float* __restrict__ w,
                                                     float* __restrict__ gfeat,
                                                     long long rows) {
  const int lane = threadIdx.x & 63;
  long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
  for (; row < rows; row += stride) {
    const int b = (int)(row / n);
    const int* ii = idx + row * 3;
    const float* ww = w + row * 3;
    float* f0 = gfeat + ((size_t)b * m + ii[0]) * C;
    float* f1 = gfeat + ((size_t)b * m + ii[1]) * C;
    float* f2 = gfeat + ((size_t)b * m + ii[2]) * C;
    const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
    const float* g = gout + row * ldo + col0;
    for (int c = lane; c < C; c += 64) {
      const float v = g[c];
      atomicAdd(f0 + c, v * w0);
      atomicAdd(f1 + c, v * w1);
      atomicAdd(f2 + c, v * w2);
    }
  }
}

static inline int grid_rows(long long rows) {
  long long g = (rows + 3) / 4;
  const long long cap = 256LL * 16;
  return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

}  // namespace demf

using namespace demf;

static bool three_nn_split(int m) {
  static const int on = [] { const char* v = getenv("DEMF_THREE_NN_SPLIT"); return v ? atoi(v) : 1; }();   // A/B switch
  return on && m >= 64;
}

extern "C" int demf_three_nn_f32(int B, int n, int m, const float* target,
                                 const float* source, float* dist2, int* idx,
                                 demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && n >= 0 && m >= 1, "three_nn: bad sizes B=%d n=%d m=%d", B, n, m);
  if (B == 0 || n == 0) return DEMF_OK;
  DEMF_REQUIRE(target && source && dist2 && idx, "three_nn: null pointer");
  if (three_nn_split(m))
    hipLaunchKernelGGL(three_nn_split_kernel, dim3(cdiv(n, 32), B), dim3(256), 0,
                       (hipStream_t)stream, n, m, target, source, dist2, idx, (float*)nullptr);
  else
    hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 256), B), dim3(256), 0,
                       (hipStream_t)stream, n, m, target, source, dist2, idx, (float*)nullptr);
  return check_launch("three_nn");
}

extern "C" int demf_three_nn_weights_f32(int B, int n, int m, const float* target, const float* source,
                                         float* dist, int* idx, float* weight, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && n >= 0 && m >= 1, "three_nn_weights: bad sizes B=%d n=%d m=%d", B, n, m);
  if (B == 0 || n == 0) return DEMF_OK;
  DEMF_REQUIRE(target && source && dist && idx && weight, "three_nn_weights: null pointer");
  if (three_nn_split(m))
    hipLaunchKernelGGL(three_nn_split_kernel, dim3(cdiv(n, 32), B), dim3(256), 0,
                       (hipStream_t)stream, n, m, target, source, dist, idx, weight);
  else
    hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 256), B), dim3(256), 0,
                       (hipStream_t)stream, n, m, target, source, dist, idx, weight);
  return check_launch("three_nn_weights");
}

extern "C" int demf_three_interpolate_fwd(int B, int C, int m, int n, const float* features,
                                          const int* idx, const float* weight, float* out,
                                          demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && C >= 0 && m >= 1 && n >= 0, "three_interpolate: bad sizes");
  if (B == 0 || C == 0 || n == 0) return DEMF_OK;
  DEMF_REQUIRE(features && idx && weight && out, "three_interpolate: null pointer");
  hipLaunchKernelGGL((interp_cm_fwd<8>), dim3(cdiv(n, 256), cdiv(C, 8), B), dim3(256), 0,
                     (hipStream_t)stream, C, m, n, features, idx, weight, out);
  return check_launch("three_interpolate_fwd");
}

extern "C" int demf_three_interpolate_bwd(int B, int C, int n, int m, const float* grad_out,
                                          const int* idx, const float* weight,
                                          float* grad_features, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && C >= 0 && m >= 1 && n >= 0, "three_interpolate: bad sizes");
  if (B == 0 || C == 0 || n == 0) return DEMF_OK;
  DEMF_REQUIRE(grad_out && idx && weight && grad_features, "three_interpolate: null pointer");
  hipLaunchKernelGGL((interp_cm_bwd<8>), dim3(cdiv(n, 256), cdiv(C, 8), B), dim3(256), 0,
                     (hipStream_t)stream, C, n, m, grad_out, idx, weight, grad_features);
  return check_launch("three_interpolate_bwd");
}

extern "C" int demf_three_interpolate_cl_fwd(int B, int m, int n, int C, int ldo, int col0,
                                             const float* feat, const int* idx,
                                             const float* weight, float* out,
                                             demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && m >= 1 && n >= 0 && C >= 0 && col0 >= 0 && col0 + C <= ldo,
               "three_interpolate_cl: bad sizes");
  if (B == 0 || C == 0 || n == 0) return DEMF_OK;
  DEMF_REQUIRE(feat && idx && weight && out, "three_interpolate_cl: null pointer");
  const long long rows = (long long)B * n;
  hipLaunchKernelGGL(interp_cl_fwd, dim3(grid_rows(rows)), dim3(256), 0, (hipStream_t)stream,
                     m, n, C, ldo, col0, feat, idx, weight, out, rows, (const float*)nullptr, 0);
  return check_launch("three_interpolate_cl_fwd");
}

extern "C" int demf_three_interpolate_cat_cl_fwd(int B, int m, int n, int C, int Cs, const float* feat,
                                                 const int* idx, const float* weight, const float* skip,
                                                 float* out, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && m >= 1 && n >= 0 && C >= 1 && Cs >= 1, "three_interpolate_cat_cl: bad sizes");
  if (B == 0 || n == 0) return DEMF_OK;
  DEMF_REQUIRE(feat && idx && weight && skip && out, "three_interpolate_cat_cl: null pointer");
  const long long rows = (long long)B * n;
  hipLaunchKernelGGL(interp_cl_fwd, dim3(grid_rows(rows)), dim3(256), 0, (hipStream_t)stream,
                     m, n, C, C + Cs, 0, feat, idx, weight, out, rows, skip, Cs);
  return check_launch("three_interpolate_cat_cl_fwd");
}

extern "C" int demf_three_interpolate_cl_bwd(int B, int m, int n, int C, int ldo, int col0,
                                             const float* grad_out, const int* idx,
                                             const float* weight, float* grad_feat,
                                             demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && m >= 1 && n >= 0 && C >= 0 && col0 >= 0 && col0 + C <= ldo,
               "three_interpolate_cl: bad sizes");
  if (B == 0 || C == 0 || n == 0) return DEMF_OK;
  DEMF_REQUIRE(grad_out && idx && weight && grad_feat, "three_interpolate_cl: null pointer");
  const long long rows = (long long)B * n;
  hipLaunchKernelGGL(interp_cl_bwd, dim3(grid_rows(rows)), dim3(256), 0, (hipStream_t)stream,
                     m, n, C, ldo, col0, grad_out, idx, weight, grad_feat, rows);
  return check_launch("three_interpolate_cl_bwd");
}
