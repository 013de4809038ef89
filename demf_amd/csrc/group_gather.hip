// Neighbour gather / scatter kernels for gfx950.
//
// (1) The operator-compatible, channel-major forms (features (B,C,N)) that stand in
//     for mmdet3d.ops grouping_operation / gather_points - reached from
//     QueryAndGroup / PointSAModule (class_agnostic_vote_head.py:383,455).
// (2) Point-major ("channels-last", features (B,N,C)) forms used by the MI355X
//     modules: a neighbour is one contiguous C*4-byte row, so the gather is a
//     coalesced float4 row copy instead of C scattered 4-byte reads, and the output
//     (B,M,ns,ld) is directly the A operand of the shared-MLP GEMM.
// All of these are pure HBM/L2 byte movers: the roofline is bytes written + bytes
// gathered over 8 TB/s.
#include "common.h"
#include <stdlib.h>

namespace demf {

// ---------------- channel-major (operator ABI) -----------------------------
// out[b,c,j] = feat[b,c,idx[b,j]]  with j over M*ns; CT channels per thread so the
// index is read once per CT outputs.
template <int CT>
__global__ __launch_bounds__(256) void group_cm_fwd(int C, int N, int J,
                                                    const float* __restrict__ feat,
                                                    const int* __restrict__ idx,
                                                    float* __restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = blockIdx.y * CT;
  if (j >= J) return;
  const int i = idx[(size_t)b * J + j];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) {
    const int c = c0 + cc;
    if (c < C) out[((size_t)b * C + c) * J + j] = feat[((size_t)b * C + c) * N + i];
  }
}

template <int CT>
__global__ __launch_bounds__(256) void group_cm_bwd(int C, int N, int J,
                                                    const float* __restrict__ gout,
                                                    const int* __restrict__ idx,
                                                    float* __restrict__ gfeat) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = blockIdx.y * CT;
  if (j >= J) return;
  const int i = idx[(size_t)b * J + j];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) {
    const int c = c0 + cc;
    if (c < C)
      atomicAdd(&gfeat[((size_t)b * C + c) * N + i], gout[((size_t)b * C + c) * J + j]);
  }
}

// ---------------- point-major fused QueryAndGroup --------------------------
// One wave per output row (b,m,s).  Row layout: [feat(C) | rel_xyz(3) | 0-pad].
template <bool VEC4>
__global__ __launch_bounds__(256) void group_concat_cl_fwd_k(
    int N, int M, int ns, int C, int ldo, int xyz_col, int feat_col, float radius,
    const float* __restrict__ xyz, const float* __restrict__ center,
    const float* __restrict__ feat, const int* __restrict__ idx, float* __restrict__ out,
    long long rows) {
  const int lane = threadIdx.x & 63;
  long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
  for (; row < rows; row += stride) {
    const long long bm = row / ns;            // b*M + m
    const int b = (int)(bm / M);
    const int i = idx[row];
    float* o = out + row * ldo;
    if (C > 0) {
      const float* f = feat + ((size_t)b * N + i) * C;
      if constexpr (VEC4) {
        for (int c = lane * 4; c < C; c += 256)
          *reinterpret_cast<float4*>(o + feat_col + c) =
              *reinterpret_cast<const float4*>(f + c);
      } else {
        for (int c = lane; c < C; c += 64) o[feat_col + c] = f[c];
      }
    }
    if (lane < 3) {
      const float p = xyz[((size_t)b * N + i) * 3 + lane];
      const float q = center[bm * 3 + lane];
      o[xyz_col + lane] = (p - q) / radius;  // upstream: grouped_xyz /= max_radius
    }
    // zero every column not covered above
    for (int c = lane; c < ldo; c += 64) {
      const bool in_feat = (c >= feat_col && c < feat_col + C);
      const bool in_xyz = (c >= xyz_col && c < xyz_col + 3);
      if (!in_feat && !in_xyz) o[c] = 0.f;
    }
  }
}

// Thin rows (ld <= 8: the first SA level carries one feature channel): one THREAD per output row
// instead of one wave - 1 M rows of 16 bytes are otherwise 1 M mostly idle waves.
__global__ __launch_bounds__(256) void group_concat_cl_fwd_thin_k(
    int N, int M, int ns, int C, int ldo, int xyz_col, int feat_col, float radius,
    const float* __restrict__ xyz, const float* __restrict__ center,
    const float* __restrict__ feat, const int* __restrict__ idx, float* __restrict__ out,
    long long rows) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += stride) {
    const long long bm = row / ns;
    const int b = (int)(bm / M);
    const int i = idx[row];
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.f;
    const float* f = feat + ((size_t)b * N + i) * C;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c >= feat_col && c < feat_col + C) v[c] = f[c - feat_col];
    const float* pp = xyz + ((size_t)b * N + i) * 3;
    const float* qq = center + bm * 3;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c >= xyz_col && c < xyz_col + 3) v[c] = (pp[c - xyz_col] - qq[c - xyz_col]) / radius;
    float* o = out + row * ldo;
    if (ldo == 4) {
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else if (ldo == 8) {
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < ldo) o[c] = v[c];
    }
  }
}

template <bool VEC4>
__global__ __launch_bounds__(256) void group_concat_cl_bwd_k(
    int N, int M, int ns, int C, int ldo, int xyz_col, int feat_col, float inv_r,
    const float* __restrict__ gout, const int* __restrict__ idx, float* __restrict__ gfeat,
    float* __restrict__ gxyz, float* __restrict__ gcenter, long long rows) {
  const int lane = threadIdx.x & 63;
  long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
  for (; row < rows; row += stride) {
    const long long bm = row / ns;
    const int b = (int)(bm / M);
    const int i = idx[row];
    if (gfeat != nullptr && C > 0) {
      const float* g = gout + row * ldo + feat_col;
      float* f = gfeat + ((size_t)b * N + i) * C;
      if constexpr (VEC4) {
        for (int c = lane * 4; c < C; c += 256) {
          const float4 v = *reinterpret_cast<const float4*>(g + c);
          atomicAdd(f + c + 0, v.x);
          atomicAdd(f + c + 1, v.y);
          atomicAdd(f + c + 2, v.z);
          atomicAdd(f + c + 3, v.w);
        }
      } else {
        for (int c = lane; c < C; c += 64) atomicAdd(f + c, g[c]);
      }
    }
    // rel = (xyz[idx] - center) * inv_r : d/dxyz = +g*inv_r, d/dcenter = -g*inv_r
    if (gxyz != nullptr && lane < 3) {
      const float g = gout[row * ldo + xyz_col + lane] * inv_r;
      atomicAdd(gxyz + ((size_t)b * N + i) * 3 + lane, g);
      atomicAdd(gcenter + bm * 3 + lane, -g);
    }
  }
}

// ---- inverse neighbour lists --------------------------------------------------------------
// idx (B, E) with values in [0, N) (E = M*ns entries of a ball query) -> CSR by source point:
// off (B, N+1), rows (B, E) = the entry positions e (= m*ns+s) that reference point j, ascending
// within a list.  One block per scene; histogram, scan and cursors live in LDS (N <= 16384).
// Depends on coordinates only, so the training step computes it in the pre-pass.
constexpr int INV_MAX_N = 16384;
__global__ __launch_bounds__(1024) void invert_index_k(int N, int E, const int* __restrict__ idx,
                                                       int* __restrict__ off,
                                                       int* __restrict__ rows, int lds_rows) {
  asm volatile("" ::: "v127");            // one WG per scene for ~100s of us: keep the CU to itself
  extern __shared__ int s_inv[];          // cnt[N] | start[N+1]
  int* cnt = s_inv;
  int* start = s_inv + N;
  __shared__ int s_part[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  idx += (size_t)b * E;
  off += (size_t)b * (N + 1);
  rows += (size_t)b * E;
  for (int j = tid; j < N; j += 1024) cnt[j] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += 1024) atomicAdd(&cnt[idx[e]], 1);
  __syncthreads();
  // exclusive scan of cnt: each thread owns a contiguous chunk
  const int chunk = (N + 1023) / 1024;
  const int j0 = tid * chunk, j1 = j0 + chunk < N ? j0 + chunk : N;
  int local = 0;
  for (int j = j0; j < j1; ++j) local += cnt[j];
  s_part[tid] = local;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = tid >= d ? s_part[tid - d] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - local;
  for (int j = j0; j < j1; ++j) {
    start[j] = run;
    off[j] = run;
    run += cnt[j];
  }
  if (tid == 1023) { start[N] = E; off[N] = E; }
  __syncthreads();
  for (int j = tid; j < N; j += 1024) cnt[j] = 0;      // reuse as fill cursors
  __syncthreads();
  if (lds_rows) {
    // the unsorted lists stay in LDS; every entry then counts the smaller entries of its own list
    // (its rank) and is written straight to its final place: no dependent chain, no global sort
    int* s_rows = s_inv + 2 * N + 1;
    for (int e = tid; e < E; e += 1024) {
      const int j = idx[e];
      s_rows[start[j] + atomicAdd(&cnt[j], 1)] = e;
    }
    __syncthreads();
    for (int e = tid; e < E; e += 1024) {
      const int j = idx[e];
      const int s = start[j], n = cnt[j];
      int rank = 0;
      for (int i = 0; i < n; ++i) rank += s_rows[s + i] < e ? 1 : 0;
      rows[s + rank] = e;
    }
    return;
  }
  for (int e = tid; e < E; e += 1024) {
    const int j = idx[e];
    rows[start[j] + atomicAdd(&cnt[j], 1)] = e;
  }
  __syncthreads();
  // make every list ascending (the fill order above depends on wave scheduling)
  for (int j = tid; j < N; j += 1024) {
    int* l = rows + start[j];
    const int n = cnt[j];
    for (int a = 1; a < n; ++a) {
      const int v = l[a];
      int c = a - 1;
      while (c >= 0 && l[c] > v) { l[c + 1] = l[c]; --c; }
      l[c + 1] = v;
    }
  }
}

// grad_feat (B,N,C) = sum over the entries that gathered point j of grad_out rows - the
// atomic-free transpose of group_concat_cl_fwd's feature copy.  LPR lanes (C/4) own one
// destination row; 4 source rows in flight; every destination row is written exactly once.
template <int LPR>
__global__ __launch_bounds__(256) void group_concat_cl_bwd_gather_k(
    int N, int E, int C, int ldo, int feat_col, const float* __restrict__ gout,
    const int* __restrict__ off, const int* __restrict__ rows, float* __restrict__ gfeat,
    long long points) {
  const int sub = threadIdx.x % LPR;
  long long pt = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
  const long long stride = (long long)gridDim.x * blockDim.x / LPR;
  for (; pt < points; pt += stride) {
    const int b = (int)(pt / N), j = (int)(pt - (long long)b * N);
    const int* o = off + (size_t)b * (N + 1) + j;
    const int e0 = o[0], e1 = o[1];
    const int* r = rows + (size_t)b * E;
    const float* g = gout + (size_t)b * E * ldo + feat_col;
    for (int c = sub * 4; c < C; c += LPR * 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int e = e0;
      for (; e + 3 < e1; e += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = *reinterpret_cast<const float4*>(g + (size_t)r[e + u] * ldo + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        }
      }
      for (; e < e1; ++e) {
        const float4 v = *reinterpret_cast<const float4*>(g + (size_t)r[e] * ldo + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<float4*>(gfeat + ((size_t)b * N + j) * C + c) = acc;
    }
  }
}

// rows gather (B,N,C)[idx (B,M)] -> (B,M,C): G lanes per row, float4 when aligned
__global__ __launch_bounds__(256) void gather_rows_cl_fwd_k(int N, int M, int C,
                                                            const float* __restrict__ feat,
                                                            const int* __restrict__ idx,
                                                            float* __restrict__ out,
                                                            long long rows) {
  const int lane = threadIdx.x & 63;
  long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
  for (; row < rows; row += stride) {
    const int b = (int)(row / M);
    const int i = idx[row];
    const float* f = feat + ((size_t)b * N + i) * C;
    float* o = out + row * C;
    for (int c = lane; c < C; c += 64) o[c] = f[c];
  }
}

__global__ __launch_bounds__(256) void gather_rows_cl_bwd_k(int N, int M, int C,
                                                            const float* __restrict__ gout,
                                                            const int* __restrict__ idx,
                                                            float* __restrict__ gfeat,
                                                            long long rows) {
  const int lane = threadIdx.x & 63;
  long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
  for (; row < rows; row += stride) {
    const int b = (int)(row / M);
    const int i = idx[row];
    const float* g = gout + row * C;
    float* f = gfeat + ((size_t)b * N + i) * C;
    for (int c = lane; c < C; c += 64) atomicAdd(f + c, g[c]);
  }
}

// ---------------- max over the ns neighbours --------------------------------
// x (R,ns,C) -> out (R,C), arg (R,C).  One thread per (r,c); consecutive threads walk
// consecutive channels so every read is coalesced.  First maximum wins.
__global__ __launch_bounds__(256) void maxpool_ns_fwd_k(long long RC, int ns, int C,
                                                        const float* __restrict__ x,
                                                        float* __restrict__ out,
                                                        int* __restrict__ arg) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; t < RC; t += stride) {
    const long long r = t / C;
    const int c = (int)(t - r * C);
    const float* p = x + r * ns * C + c;
    float best = p[0];
    int bi = 0;
    for (int s = 1; s < ns; ++s) {
      const float v = p[(size_t)s * C];
      if (v > best) {
        best = v;
        bi = s;
      }
    }
    out[t] = best;
    arg[t] = bi;
  }
}

__global__ __launch_bounds__(256) void maxpool_ns_bwd_k(long long RC, int ns, int C,
                                                        const float* __restrict__ gout,
                                                        const int* __restrict__ arg,
                                                        float* __restrict__ gx) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; t < RC; t += stride) {
    const long long r = t / C;
    const int c = (int)(t - r * C);
    float* p = gx + r * ns * C + c;
    const float g = gout[t];
    const int a = arg[t];
    for (int s = 0; s < ns; ++s) p[(size_t)s * C] = (s == a) ? g : 0.f;
  }
}

static inline int grid_for_rows(long long rows, int rows_per_block) {
  long long g = (rows + rows_per_block - 1) / rows_per_block;
  const long long cap = 256LL * 16;
  return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

// column sums of a row-major (R, N) matrix (row stride ld): the bias gradient of a linear layer.
// One block = a chunk of rows; a thread owns VEC consecutive columns and keeps 4 rows in flight;
// row lanes are folded through LDS, then one atomic per column per block.
template <int VEC>
__global__ __launch_bounds__(256) void colsum_k(int R, int N, int ld, int rows_per_block,
                                                const float* __restrict__ x,
                                                float* __restrict__ out) {
  __shared__ float red[256 * VEC];
  const int cg = (N + VEC - 1) / VEC;                 // column groups
  const int cgp = cg < 256 ? cg : 256;                // column groups per pass
  const int rows_par = 256 / cgp;
  const int c_in = threadIdx.x % cgp, r_in = threadIdx.x / cgp;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  for (int cb = 0; cb < cg; cb += cgp) {
    const int c = (cb + c_in) * VEC;
    float acc[VEC] = {};
    if (r_in < rows_par && c < N) {
      int r = r0 + r_in;
      if constexpr (VEC == 4) {
        for (; r + 3 * rows_par < r1; r += 4 * rows_par) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            v[u] = *reinterpret_cast<const float4*>(x + (size_t)(r + u * rows_par) * ld + c);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc[0] += v[u].x; acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w;
          }
        }
        for (; r < r1; r += rows_par) {
          const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ld + c);
          acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        }
      } else {
        for (; r < r1; r += rows_par) acc[0] += x[(size_t)r * ld + c];
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) red[threadIdx.x * VEC + v] = acc[v];
    __syncthreads();
    if (threadIdx.x < cgp && (cb + threadIdx.x) * VEC < N) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float t = 0.f;
        for (int j = 0; j < rows_par; ++j) t += red[(j * cgp + threadIdx.x) * VEC + v];
        if ((cb + threadIdx.x) * VEC + v < N) atomicAdd(out + (cb + threadIdx.x) * VEC + v, t);
      }
    }
    __syncthreads();
  }
}

// (B,C,HW) channel-major map -> rows [row0, row0+HW) of the channels-last token buffer (B,S,C):
// the flatten(2).transpose(1,2) + concat of prepare_decoder_inputs (class_agnostic_vote_head.py:
// 570-591) as a tiled transpose through LDS (both sides coalesced).
__global__ __launch_bounds__(256) void nchw_to_tokens_k(int C, int HW, int S, int row0,
                                                        const float* __restrict__ src,
                                                        const unsigned char* __restrict__ mask,
                                                        float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  src += (size_t)b * C * HW;
  dst += ((size_t)b * S + row0) * C;
  if (mask != nullptr) mask += (size_t)b * S + row0;
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, p = p0 + tx;
    tile[j][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int p = p0 + j, c = c0 + tx;
    if (p < HW && c < C)
      dst[(size_t)p * C + c] = (mask != nullptr && mask[p]) ? 0.f : tile[tx][j];   // padding token
  }
}

// All levels of the pyramid in one launch, 64 pixels x 64 channels per workgroup: 256-byte segments on
// the channel-major side, whole 256-byte row pieces on the token side, float4 both ways.
struct TokLevels {
  const float* src[8];
  int hw[8], row0[8], tile0[9];
  int n;
};
template <bool OB>     // OB: bf16 token rows (2-byte elements, round to nearest even) - C % 4 == 0
__global__ __launch_bounds__(256) void pyramid_to_tokens_k(int C, int S, TokLevels lv,
                                                           const unsigned char* __restrict__ mask,
                                                           void* __restrict__ dst_) {
  __shared__ float tile[64][65];
  const int bx = blockIdx.x, b = blockIdx.z, t = threadIdx.x;
  int l = 0;
  while (l + 1 < lv.n && bx >= lv.tile0[l + 1]) ++l;
  const int HW = lv.hw[l], p0 = (bx - lv.tile0[l]) * 64, c0 = blockIdx.y * 64;
  const float* __restrict__ src = lv.src[l] + (size_t)b * C * HW;
  const size_t drow = ((size_t)b * S + lv.row0[l]) * C;
  float* __restrict__ d = reinterpret_cast<float*>(dst_) + drow;
  unsigned short* __restrict__ d16 = reinterpret_cast<unsigned short*>(dst_) + drow;
  const unsigned char* m = mask != nullptr ? mask + (size_t)b * S + lv.row0[l] : nullptr;
  const int q = t & 15, r = t >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cl = r + 16 * i, c = c0 + cl, p = p0 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
      const float* g = src + (size_t)c * HW + p;
      if (p + 3 < HW) {
        v = *reinterpret_cast<const float4*>(g);
      } else {
        if (p < HW) v.x = g[0];
        if (p + 1 < HW) v.y = g[1];
        if (p + 2 < HW) v.z = g[2];
      }
    }
    tile[cl][4 * q] = v.x; tile[cl][4 * q + 1] = v.y; tile[cl][4 * q + 2] = v.z; tile[cl][4 * q + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = r + 16 * i, pp = p0 + pl, c = c0 + 4 * q;
    if (pp >= HW || c >= C) continue;
    const bool z = m != nullptr && m[pp];                       // padding token
    float4 v = make_float4(tile[4 * q][pl], tile[4 * q + 1][pl], tile[4 * q + 2][pl], tile[4 * q + 3][pl]);
    if (z) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (OB) {
      const __bf16 h[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      *reinterpret_cast<uint2*>(d16 + (size_t)pp * C + c) = *reinterpret_cast<const uint2*>(h);
      continue;
    }
    float* o = d + (size_t)pp * C + c;
    if (c + 3 < C) {
      *reinterpret_cast<float4*>(o) = v;
    } else {
      o[0] = v.x;
      if (c + 1 < C) o[1] = v.y;
      if (c + 2 < C) o[2] = v.z;
    }
  }
}

// ---- the same inverse lists spread over the chip (round 5) ------------------------------------------------------
// invert_index_k is ONE workgroup per scene, and at SA1 (131 072 entries over 20 000 points) the lists do not fit
// LDS next to the histogram, so it fills them through global cursors and insertion-sorts every list in global
// memory: 102 us on the serial pre-pass chain.  With E ints of workspace per scene:
//   inv_hist_k   counts[j] += 1                               one thread per entry, atomics into off[j + 1]
//   inv_scan_k   off[j + 1] <- start[j] (exclusive scan, in place)   one workgroup per scene
//   inv_fill_k   ws[atomicAdd(off[j + 1], 1)] = e             one thread per entry; afterwards off[j + 1] = start[j + 1],
//                                                             i.e. the final offsets without a second cursor array
//   inv_rank_k   rows[start + #{e' in list: e' < e}] = e      one thread per (unsorted) list position: ascending
//                                                             lists whatever order the atomics filled them in
__global__ __launch_bounds__(256) void inv_zero_k(long long n, int* __restrict__ a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = 0;
}
__global__ __launch_bounds__(256) void inv_hist_k(int N, int E, const int* __restrict__ idx, int* __restrict__ off) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e < E) atomicAdd(off + (size_t)b * (N + 1) + idx[(size_t)b * E + e] + 1, 1);
}
__global__ __launch_bounds__(1024) void inv_scan_k(int N, int* __restrict__ off) {
  __shared__ int s_part[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  int* a = off + (size_t)b * (N + 1) + 1;          // counts of lists 0 .. N-1
  const int chunk = (N + 1023) / 1024;
  const int j0 = tid * chunk, j1 = j0 + chunk < N ? j0 + chunk : N;
  int local = 0;
  for (int j = j0; j < j1; ++j) local += a[j];
  s_part[tid] = local;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = tid >= d ? s_part[tid - d] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - local;
  for (int j = j0; j < j1; ++j) {
    const int c = a[j];
    a[j] = run;
    run += c;
  }
  if (tid == 0) a[-1] = 0;
}
__global__ __launch_bounds__(256) void inv_fill_k(int N, int E, const int* __restrict__ idx, int* __restrict__ off,
                                                  int* __restrict__ ws) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int j = idx[(size_t)b * E + e];
  ws[(size_t)b * E + atomicAdd(off + (size_t)b * (N + 1) + j + 1, 1)] = e;
}
__global__ __launch_bounds__(256) void inv_rank_k(int N, int E, const int* __restrict__ idx, const int* __restrict__ off,
                                                  const int* __restrict__ ws, int* __restrict__ rows) {
  const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  if (q >= E) return;
  ws += (size_t)b * E;
  const int e = ws[q];
  const int j = idx[(size_t)b * E + e];
  const int* o = off + (size_t)b * (N + 1);
  const int s = o[j], n = o[j + 1] - s;            // (after inv_fill_k: off[j] = start[j], off[j + 1] = its end)
  int rank = 0;
  for (int i = 0; i < n; ++i) rank += ws[s + i] < e ? 1 : 0;
  rows[(size_t)b * E + s + rank] = e;
}

// sa_indices of the backbone (PointNet2SASSG.forward, mmdet3d pointnet2_sa_ssg.py: indices into the INPUT cloud
// of every level's samples): out_0 = arange(N), out_l[b][i] = out_{l-1}[b][idx_l[b][i]].  One workgroup per
// scene walks the levels (their sizes shrink 20 000 -> 2 048 -> ... -> 256), one launch instead of
// arange + repeat + 4 x (int64 conversion + torch.gather).
struct IndexChain {
  const int* idx[8];
  long long* out[9];
  int m[9];              // m[0] = N, m[l] = samples of level l
  int n;
};
__global__ __launch_bounds__(1024) void index_chain_k(IndexChain c) {
  const int b = blockIdx.x;
  long long* o0 = c.out[0] + (size_t)b * c.m[0];
  for (int i = threadIdx.x; i < c.m[0]; i += blockDim.x) o0[i] = i;
  for (int l = 1; l <= c.n; ++l) {
    __syncthreads();
    const long long* prev = c.out[l - 1] + (size_t)b * c.m[l - 1];
    const int* id = c.idx[l - 1] + (size_t)b * c.m[l];
    long long* o = c.out[l] + (size_t)b * c.m[l];
    for (int i = threadIdx.x; i < c.m[l]; i += blockDim.x) o[i] = prev[id[i]];
  }
}

// points (rows, 3 + C) -> xyz (rows, 3) and the feature columns (rows, C): the two strided copies of the
// pre-pass in one launch
__global__ __launch_bounds__(256) void split_points_k(long long rows, int C, const float* __restrict__ pts,
                                                      float* __restrict__ xyz, float* __restrict__ feat) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* p = pts + r * (3 + C);
  xyz[3 * r] = p[0]; xyz[3 * r + 1] = p[1]; xyz[3 * r + 2] = p[2];
  for (int c = 0; c < C; ++c) feat[r * C + c] = p[3 + c];
}

}  // namespace demf


using namespace demf;

static int pyramid_to_tokens_impl(int B, int C, int S, int nlev, const float* const* srcs, const int* hws,
                                  const unsigned char* mask, void* dst, bool bf16, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && C >= 1 && nlev >= 1 && nlev <= 8 && srcs && hws, "pyramid_to_tokens: bad arguments");
  DEMF_REQUIRE(!bf16 || C % 4 == 0, "pyramid_to_tokens_bf16: C = %d is not a multiple of 4", C);
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(dst, "pyramid_to_tokens: null pointer");
  TokLevels lv{};
  int row = 0, tiles = 0;
  for (int l = 0; l < nlev; ++l) {
    DEMF_REQUIRE(srcs[l] && hws[l] >= 1, "pyramid_to_tokens: level %d", l);
    lv.src[l] = srcs[l]; lv.hw[l] = hws[l]; lv.row0[l] = row; lv.tile0[l] = tiles;
    row += hws[l];
    tiles += (hws[l] + 63) / 64;
  }
  lv.tile0[nlev] = tiles;
  lv.n = nlev;
  DEMF_REQUIRE(row == S, "pyramid_to_tokens: the levels hold %d tokens, S = %d", row, S);
  const dim3 grid(tiles, (C + 63) / 64, B);
  if (bf16) hipLaunchKernelGGL(pyramid_to_tokens_k<true>, grid, dim3(256), 0, (hipStream_t)stream, C, S, lv, mask, dst);
  else hipLaunchKernelGGL(pyramid_to_tokens_k<false>, grid, dim3(256), 0, (hipStream_t)stream, C, S, lv, mask, dst);
  return check_launch("pyramid_to_tokens");
}

extern "C" int demf_pyramid_to_tokens(int B, int C, int S, int nlev, const float* const* srcs, const int* hws,
                                      const unsigned char* mask, float* dst, demf_stream_t stream) {
  return pyramid_to_tokens_impl(B, C, S, nlev, srcs, hws, mask, dst, false, stream);
}

extern "C" int demf_pyramid_to_tokens_bf16(int B, int C, int S, int nlev, const float* const* srcs, const int* hws,
                                           const unsigned char* mask, uint16_t* dst, demf_stream_t stream) {
  return pyramid_to_tokens_impl(B, C, S, nlev, srcs, hws, mask, dst, true, stream);
}

extern "C" int demf_group_points_fwd(int B, int C, int N, int M, int ns,
                                     const float* features, const int* idx, float* out,
                                     demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && ns >= 1, "group_points: bad sizes");
  if (B == 0 || C == 0 || M == 0) return DEMF_OK;
  DEMF_REQUIRE(features && idx && out, "group_points: null pointer");
  const int J = M * ns;
  dim3 grid(cdiv(J, 256), cdiv(C, 8), B);
  hipLaunchKernelGGL((group_cm_fwd<8>), grid, dim3(256), 0, (hipStream_t)stream, C, N, J,
                     features, idx, out);
  return check_launch("group_points_fwd");
}

extern "C" int demf_group_points_bwd(int B, int C, int N, int M, int ns,
                                     const float* grad_out, const int* idx,
                                     float* grad_features, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && ns >= 1, "group_points: bad sizes");
  if (B == 0 || C == 0 || M == 0) return DEMF_OK;
  DEMF_REQUIRE(grad_out && idx && grad_features, "group_points: null pointer");
  const int J = M * ns;
  dim3 grid(cdiv(J, 256), cdiv(C, 8), B);
  hipLaunchKernelGGL((group_cm_bwd<8>), grid, dim3(256), 0, (hipStream_t)stream, C, N, J,
                     grad_out, idx, grad_features);
  return check_launch("group_points_bwd");
}

extern "C" int demf_gather_points_fwd(int B, int C, int N, int M, const float* features,
                                      const int* idx, float* out, demf_stream_t stream) {
  return demf_group_points_fwd(B, C, N, M, 1, features, idx, out, stream);
}

extern "C" int demf_gather_points_bwd(int B, int C, int N, int M, const float* grad_out,
                                      const int* idx, float* grad_features,
                                      demf_stream_t stream) {
  return demf_group_points_bwd(B, C, N, M, 1, grad_out, idx, grad_features, stream);
}

extern "C" int demf_group_concat_cl_fwd(int B, int N, int M, int ns, int C, int ldo,
                                        int xyz_col, int feat_col, float radius,
                                        int normalize_xyz, const float* xyz,
                                        const float* center, const float* feat,
                                        const int* idx, float* out, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && M >= 0 && ns >= 1 && C >= 0, "group_concat: bad sizes");
  DEMF_REQUIRE(xyz_col >= 0 && xyz_col + 3 <= ldo && feat_col >= 0 && feat_col + C <= ldo &&
                   (C == 0 || xyz_col + 3 <= feat_col || feat_col + C <= xyz_col),
               "group_concat: column ranges xyz=%d feat=%d C=%d ld=%d", xyz_col, feat_col, C,
               ldo);
  if (B == 0 || M == 0) return DEMF_OK;
  DEMF_REQUIRE(xyz && center && idx && out && (C == 0 || feat), "group_concat: null pointer");
  const long long rows = (long long)B * M * ns;
  const float inv_r = normalize_xyz ? radius : 1.0f;  // divisor
  const bool vec4 = (C % 4 == 0) && (ldo % 4 == 0) && (feat_col % 4 == 0) &&
                    (((uintptr_t)feat | (uintptr_t)out) % 16 == 0);
  if (ldo <= 8 && (ldo % 4 != 0 || ((uintptr_t)out) % 16 == 0)) {
    hipLaunchKernelGGL(group_concat_cl_fwd_thin_k, dim3(grid_for_rows(rows, 256)), dim3(256), 0,
                       (hipStream_t)stream, N, M, ns, C, ldo, xyz_col, feat_col, inv_r, xyz,
                       center, feat, idx, out, rows);
    return check_launch("group_concat_cl_fwd");
  }
  const int grid = grid_for_rows(rows, 4);
  if (vec4)
    hipLaunchKernelGGL((group_concat_cl_fwd_k<true>), dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, N, M, ns, C, ldo, xyz_col, feat_col, inv_r, xyz,
                       center, feat, idx, out, rows);
  else
    hipLaunchKernelGGL((group_concat_cl_fwd_k<false>), dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, N, M, ns, C, ldo, xyz_col, feat_col, inv_r, xyz,
                       center, feat, idx, out, rows);
  return check_launch("group_concat_cl_fwd");
}

extern "C" int demf_group_concat_cl_bwd(int B, int N, int M, int ns, int C, int ldo,
                                        int xyz_col, int feat_col, float radius,
                                        int normalize_xyz, const float* grad_out,
                                        const int* idx, float* grad_feat, float* grad_xyz,
                                        float* grad_center, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && M >= 0 && ns >= 1 && C >= 0 && feat_col >= 0 &&
                   feat_col + C <= ldo && xyz_col >= 0 && xyz_col + 3 <= ldo,
               "group_concat_bwd: bad sizes");
  if (B == 0 || M == 0) return DEMF_OK;
  DEMF_REQUIRE(grad_out && idx, "group_concat_bwd: null pointer");
  DEMF_REQUIRE((grad_xyz == nullptr) == (grad_center == nullptr),
               "group_concat_bwd: grad_xyz and grad_center must be given together");
  if ((grad_feat == nullptr || C == 0) && grad_xyz == nullptr) return DEMF_OK;
  const long long rows = (long long)B * M * ns;
  const float inv_r = normalize_xyz ? 1.0f / radius : 1.0f;
  const bool vec4 = (C % 4 == 0) && (ldo % 4 == 0) && (feat_col % 4 == 0) &&
                    (((uintptr_t)grad_out) % 16 == 0);
  const int grid = grid_for_rows(rows, 4);
  if (vec4)
    hipLaunchKernelGGL((group_concat_cl_bwd_k<true>), dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, N, M, ns, C, ldo, xyz_col, feat_col, inv_r,
                       grad_out, idx, grad_feat, grad_xyz, grad_center, rows);
  else
    hipLaunchKernelGGL((group_concat_cl_bwd_k<false>), dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, N, M, ns, C, ldo, xyz_col, feat_col, inv_r,
                       grad_out, idx, grad_feat, grad_xyz, grad_center, rows);
  return check_launch("group_concat_cl_bwd");
}

extern "C" int demf_gather_rows_cl_fwd(int B, int N, int M, int C, const float* feat,
                                       const int* idx, float* out, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && M >= 0 && C >= 0, "gather_rows: bad sizes");
  if (B == 0 || M == 0 || C == 0) return DEMF_OK;
  DEMF_REQUIRE(feat && idx && out, "gather_rows: null pointer");
  const long long rows = (long long)B * M;
  hipLaunchKernelGGL(gather_rows_cl_fwd_k, dim3(grid_for_rows(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, N, M, C, feat, idx, out, rows);
  return check_launch("gather_rows_cl_fwd");
}

extern "C" int demf_gather_rows_cl_bwd(int B, int N, int M, int C, const float* grad_out,
                                       const int* idx, float* grad_feat,
                                       demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && M >= 0 && C >= 0, "gather_rows: bad sizes");
  if (B == 0 || M == 0 || C == 0) return DEMF_OK;
  DEMF_REQUIRE(grad_out && idx && grad_feat, "gather_rows: null pointer");
  const long long rows = (long long)B * M;
  hipLaunchKernelGGL(gather_rows_cl_bwd_k, dim3(grid_for_rows(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, N, M, C, grad_out, idx, grad_feat, rows);
  return check_launch("gather_rows_cl_bwd");
}

extern "C" int demf_maxpool_ns_fwd(int R, int ns, int C, const float* x, float* out,
                                   int* arg, demf_stream_t stream) {
  DEMF_REQUIRE(R >= 0 && ns >= 1 && C >= 0, "maxpool_ns: bad sizes");
  if (R == 0 || C == 0) return DEMF_OK;
  DEMF_REQUIRE(x && out && arg, "maxpool_ns: null pointer");
  const long long RC = (long long)R * C;
  hipLaunchKernelGGL(maxpool_ns_fwd_k, dim3(grid_for_rows(RC, 256)), dim3(256), 0,
                     (hipStream_t)stream, RC, ns, C, x, out, arg);
  return check_launch("maxpool_ns_fwd");
}

extern "C" int demf_maxpool_ns_bwd(int R, int ns, int C, const float* grad_out,
                                   const int* arg, float* grad_x, demf_stream_t stream) {
  DEMF_REQUIRE(R >= 0 && ns >= 1 && C >= 0, "maxpool_ns: bad sizes");
  if (R == 0 || C == 0) return DEMF_OK;
  DEMF_REQUIRE(grad_out && arg && grad_x, "maxpool_ns: null pointer");
  const long long RC = (long long)R * C;
  hipLaunchKernelGGL(maxpool_ns_bwd_k, dim3(grid_for_rows(RC, 256)), dim3(256), 0,
                     (hipStream_t)stream, RC, ns, C, grad_out, arg, grad_x);
  return check_launch("maxpool_ns_bwd");
}

extern "C" int demf_colsum_f32(int R, int N, int ld, const float* x, float* out,
                               demf_stream_t stream) {
  if (R <= 0 || N <= 0) return DEMF_OK;
  DEMF_REQUIRE(x && out, "colsum: null pointer");
  DEMF_REQUIRE(ld >= N, "colsum: ld=%d < N=%d", ld, N);
  // at most ~256 blocks: each ends in N same-address atomics, which serialise in L2
  // and at least DEMF_COLSUM_ROWS rows per block (2 k-row layers: 32 blocks, atomic depth 32)
  int rpb = (R + 255) / 256;
  static const int min_rows = [] { const char* v = getenv("DEMF_COLSUM_ROWS"); return v ? atoi(v) : 64; }();
  rpb = rpb < min_rows ? min_rows : rpb;
  const dim3 grid((R + rpb - 1) / rpb);
  if (N % 4 == 0 && ld % 4 == 0 && ((size_t)x & 15) == 0)
    hipLaunchKernelGGL(colsum_k<4>, grid, dim3(256), 0, (hipStream_t)stream, R, N, ld, rpb, x, out);
  else
    hipLaunchKernelGGL(colsum_k<1>, grid, dim3(256), 0, (hipStream_t)stream, R, N, ld, rpb, x, out);
  return check_launch("colsum");
}

extern "C" int demf_invert_index(int B, int N, int E, const int* idx, int* off, int* rows,
                                 demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && E >= 0, "invert_index: bad sizes B=%d N=%d E=%d", B, N, E);
  if (N > INV_MAX_N) {
    set_error("invert_index: N=%d exceeds %d (LDS-resident histogram)", N, INV_MAX_N);
    return DEMF_EUNSUPPORTED;
  }
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(idx && off && rows, "invert_index: null pointer");
  // E more ints of LDS hold the unsorted lists when they fit next to the histogram (<= 150 KB)
  size_t lds = sizeof(int) * (2 * (size_t)N + 1);
  int lds_rows = 0;
  if (lds + sizeof(int) * (size_t)E <= 150 * 1024) {
    lds += sizeof(int) * (size_t)E;
    lds_rows = 1;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)invert_index_k, hipFuncAttributeMaxDynamicSharedMemorySize,
                              150 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        lds -= sizeof(int) * (size_t)E;
        lds_rows = 0;
      } else {
        attr_set = true;
      }
    }
  }
  hipLaunchKernelGGL(invert_index_k, dim3(B), dim3(1024), lds, (hipStream_t)stream, N, E, idx, off,
                     rows, lds_rows);
  return check_launch("invert_index");
}

extern "C" int demf_invert_index_ws(int B, int N, int E, const int* idx, int* off, int* rows, int* workspace,
                                    demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && E >= 0, "invert_index_ws: bad sizes");
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(idx && off && rows, "invert_index_ws: null pointer");
  static const int split = [] { const char* v = getenv("DEMF_INVERT_SPLIT"); return v ? atoi(v) : 1; }();   // A/B switch
  // The one-workgroup-per-scene form keeps its lists in LDS while they fit (150 KB: SA2's 32 768 entries over 2 048
  // points just do).  Alone on the GPU the five-launch split form is the faster one from ~28 k entries on, but the
  // inversion belongs to the coordinate pre-pass, which the training loop runs on a side stream UNDER the previous
  // step: there B busy CUs cost the step less than five whole-GPU launches (-0.065 ms per step at B = 8;
  // DEMF_INVERT_SPLIT=2 restores the 32 768-entry threshold).
  const bool fits = sizeof(int) * (2 * (size_t)N + 1 + (size_t)E) <= 150 * 1024;
  if (!split || workspace == nullptr || (split == 2 ? E < 32768 : fits))
    return demf_invert_index(B, N, E, idx, off, rows, stream);
  hipStream_t s = (hipStream_t)stream;
  // (a kernel, not hipMemsetAsync: inside the step's hipGraph a memset node was not ordered with the kernels around
  // it - the one-batch graph of bench.py --no-prefetch faulted on the stale counts)
  const long long nz = (long long)B * (N + 1);
  hipLaunchKernelGGL(inv_zero_k, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, s, nz, off);
  const dim3 ge(cdiv(E, 256), B);
  hipLaunchKernelGGL(inv_hist_k, ge, dim3(256), 0, s, N, E, idx, off);
  hipLaunchKernelGGL(inv_scan_k, dim3(B), dim3(1024), 0, s, N, off);
  hipLaunchKernelGGL(inv_fill_k, ge, dim3(256), 0, s, N, E, idx, off, workspace);
  hipLaunchKernelGGL(inv_rank_k, ge, dim3(256), 0, s, N, E, idx, off, workspace, rows);
  return check_launch("invert_index_ws");
}

extern "C" int demf_group_concat_cl_bwd_gather(int B, int N, int E, int C, int ldo, int feat_col,
                                               const float* grad_out, const int* off,
                                               const int* rows, float* grad_feat,
                                               demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && E >= 0 && C >= 4 && C % 4 == 0 && ldo % 4 == 0 &&
                   feat_col % 4 == 0 && feat_col + C <= ldo,
               "group_concat_bwd_gather: bad sizes C=%d ldo=%d feat_col=%d", C, ldo, feat_col);
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(grad_out && off && rows && grad_feat, "group_concat_bwd_gather: null pointer");
  DEMF_REQUIRE((((uintptr_t)grad_out | (uintptr_t)grad_feat) % 16) == 0,
               "group_concat_bwd_gather: buffers must be 16-byte aligned");
  const long long points = (long long)B * N;
  const int lpr = C >= 256 ? 64 : (C >= 128 ? 32 : 16);
  const long long threads = points * lpr;
  long long blocks = (threads + 255) / 256;
  if (blocks > 256LL * 16) blocks = 256LL * 16;
  const dim3 grid((unsigned)blocks);
  hipStream_t s = (hipStream_t)stream;
  if (lpr == 64)
    hipLaunchKernelGGL(group_concat_cl_bwd_gather_k<64>, grid, dim3(256), 0, s, N, E, C, ldo,
                       feat_col, grad_out, off, rows, grad_feat, points);
  else if (lpr == 32)
    hipLaunchKernelGGL(group_concat_cl_bwd_gather_k<32>, grid, dim3(256), 0, s, N, E, C, ldo,
                       feat_col, grad_out, off, rows, grad_feat, points);
  else
    hipLaunchKernelGGL(group_concat_cl_bwd_gather_k<16>, grid, dim3(256), 0, s, N, E, C, ldo,
                       feat_col, grad_out, off, rows, grad_feat, points);
  return check_launch("group_concat_cl_bwd_gather");
}

extern "C" int demf_nchw_to_tokens(int B, int C, int HW, int S, int row0, const float* src,
                                   const unsigned char* mask, float* dst, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && C >= 1 && HW >= 1 && row0 >= 0 && row0 + HW <= S,
               "nchw_to_tokens: bad sizes C=%d HW=%d S=%d row0=%d", C, HW, S, row0);
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(src && dst, "nchw_to_tokens: null pointer");
  hipLaunchKernelGGL(nchw_to_tokens_k, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0,
                     (hipStream_t)stream, C, HW, S, row0, src, mask, dst);
  return check_launch("nchw_to_tokens");
}

extern "C" int demf_sa_index_chain(int B, int N, int nlev, const int* const* idx, const int* samples,
                                   int64_t* const* out, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 1 && nlev >= 0 && nlev <= 8 && out && (nlev == 0 || (idx && samples)),
               "sa_index_chain: bad arguments");
  if (B == 0) return DEMF_OK;
  IndexChain c{};
  c.n = nlev; c.m[0] = N;
  DEMF_REQUIRE(out[0], "sa_index_chain: null pointer");
  c.out[0] = (long long*)out[0];
  for (int l = 1; l <= nlev; ++l) {
    DEMF_REQUIRE(idx[l - 1] && out[l] && samples[l - 1] >= 1, "sa_index_chain: level %d", l);
    c.idx[l - 1] = idx[l - 1]; c.out[l] = (long long*)out[l]; c.m[l] = samples[l - 1];
  }
  hipLaunchKernelGGL(index_chain_k, dim3(B), dim3(1024), 0, (hipStream_t)stream, c);
  return check_launch("sa_index_chain");
}

extern "C" int demf_split_points(long long rows, int C, const float* points, float* xyz, float* feat,
                                 demf_stream_t stream) {
  DEMF_REQUIRE(rows >= 0 && C >= 0, "split_points: bad sizes");
  if (rows == 0) return DEMF_OK;
  DEMF_REQUIRE(points && xyz && (C == 0 || feat), "split_points: null pointer");
  hipLaunchKernelGGL(split_points_k, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, C,
                     points, xyz, feat);
  return check_launch("split_points");
}
