// Test-time post-processing of DeMFVoteHead.get_bboxes (class_agnostic_vote_head.py:714-754) and the
// VoteHead.multiclass_nms_single it inherits from mmdet3d 0.18.1: per decoded box the number of
// scene points inside it and its axis-aligned corner extent, then class-aware greedy NMS on the
// extents.  Arithmetic mirrors the upstream torch code term by term (build: -ffp-contract=off).
#include <hip/hip_runtime.h>

#include "common.h"

namespace demf {

// One block per (box, scene).  box7 = (cx, cy, cz_gravity, dx, dy, dz, yaw) as decode() emits it;
// out_box = upstream's bottom-centre form, extent = min/max over the 8 rotated corners,
// count = points with |z - cz| <= dz/2, |x'| < dx/2, |y'| < dy/2 in the box frame.
__global__ __launch_bounds__(256) void box_extent_count_k(int N, int pstride, int K,
                                                          const float* __restrict__ points,
                                                          const float* __restrict__ box7,
                                                          const float* __restrict__ cosy,
                                                          const float* __restrict__ siny,
                                                          float* __restrict__ out_box,
                                                          float* __restrict__ extent,
                                                          int* __restrict__ count) {
  const int k = blockIdx.x, b = blockIdx.y;
  const size_t o = (size_t)b * K + k;
  const float* t = box7 + o * 7;
  const float dx = t[3], dy = t[4], dz = t[5];
  // DepthInstance3DBoxes(origin=(0.5,0.5,0.5)): tensor[:, :3] += dims * ((0.5,0.5,0) - origin)
  const float x0 = t[0] + dx * 0.f, y0 = t[1] + dy * 0.f, zb = t[2] + dz * -0.5f;
  const float c = cosy[o], s = siny[o];
  if (threadIdx.x == 0) {
    float* ob = out_box + o * 7;
    ob[0] = x0; ob[1] = y0; ob[2] = zb; ob[3] = dx; ob[4] = dy; ob[5] = dz; ob[6] = t[6];
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int q = 0; q < 8; ++q) {
      const float ux = dx * ((q & 4) ? 0.5f : -0.5f);
      const float uy = dy * ((q & 2) ? 0.5f : -0.5f);
      const float uz = dz * ((q & 1) ? 1.f : 0.f);
      // rotation_3d_in_axis(axis=2): x' = x cos + y sin, y' = -x sin + y cos
      const float p[3] = {(ux * c + uy * s) + x0, ((-ux) * s + uy * c) + y0, uz + zb};
      for (int a = 0; a < 3; ++a) {
        mn[a] = fminf(mn[a], p[a]);
        mx[a] = fmaxf(mx[a], p[a]);
      }
    }
    float* e = extent + o * 6;
    e[0] = mn[0]; e[1] = mn[1]; e[2] = mn[2]; e[3] = mx[0]; e[4] = mx[1]; e[5] = mx[2];
  }
  // membership frame: rotation by -yaw (cos(-y) = cos y, sin(-y) = -sin y), gravity centre
  const float gz = zb + dz * 0.5f;
  const float hx = dx / 2.f, hy = dy / 2.f, hz = dz / 2.f;
  const float cn = c, sn = -s;
  int local = 0;
  const float* pb = points + (size_t)b * N * pstride;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float r0 = pb[(size_t)n * pstride] - x0, r1 = pb[(size_t)n * pstride + 1] - y0;
    const float r2 = pb[(size_t)n * pstride + 2] - gz;
    const float lx = r0 * cn + r1 * sn;
    const float ly = (-r0) * sn + r1 * cn;
    local += (fabsf(r2) <= hz && fabsf(lx) < hx && fabsf(ly) < hy) ? 1 : 0;
  }
  __shared__ int s_cnt[4];
  for (int d = 32; d; d >>= 1) local += __shfl_xor(local, d);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) count[o] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// aligned_3d_nms, one block per scene, K <= 1024 boxes: candidates (valid != 0) are visited in
// descending score order; a visited box that is still alive is kept and removes every alive
// candidate of the same class with IoU > thr.
__global__ __launch_bounds__(1024) void aligned_nms_k(int K, float thr,
                                                      const float* __restrict__ extent,
                                                      const float* __restrict__ score,
                                                      const long long* __restrict__ cls,
                                                      const unsigned char* __restrict__ valid,
                                                      unsigned char* __restrict__ keep) {
  __shared__ float s_key[1024];
  __shared__ int s_ord[1024];
  __shared__ unsigned char s_alive[1024];
  __shared__ float s_cur[8];
  const int b = blockIdx.x, t = threadIdx.x;
  extent += (size_t)b * K * 6;
  score += (size_t)b * K;
  cls += (size_t)b * K;
  valid += (size_t)b * K;
  keep += (size_t)b * K;
  const bool mine = t < K && valid[t];
  s_key[t] = mine ? score[t] : -__builtin_inff();
  s_ord[t] = t;
  s_alive[t] = mine ? 1 : 0;
  if (t < K) keep[t] = 0;
  __syncthreads();
  // bitonic sort of (key, index), descending by key; ties: higher index first (argsort + "last")
  for (int size = 2; size <= 1024; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int p = t ^ stride;
      if (p > t) {
        const bool desc = (t & size) == 0;
        const float a = s_key[t], c = s_key[p];
        const int ia = s_ord[t], ic = s_ord[p];
        const bool a_first = a > c || (a == c && ia > ic);      // a should come before c
        if (desc ? !a_first : a_first) {
          s_key[t] = c; s_key[p] = a;
          s_ord[t] = ic; s_ord[p] = ia;
        }
      }
      __syncthreads();
    }
  float e[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float area = 0.f;
  long long my_cls = -1;
  if (t < K) {
    for (int a = 0; a < 6; ++a) e[a] = extent[t * 6 + a];
    area = (e[3] - e[0]) * (e[4] - e[1]) * (e[5] - e[2]);
    my_cls = cls[t];
  }
  for (int r = 0; r < K; ++r) {
    const int i = s_ord[r];
    if (s_key[r] == -__builtin_inff()) break;             // only invalid boxes remain (uniform)
    const bool alive_i = s_alive[i] != 0;
    __syncthreads();
    if (!alive_i) continue;                               // uniform: same LDS value for all threads
    if (t == i) {
      keep[i] = 1;
      s_alive[i] = 0;
      for (int a = 0; a < 6; ++a) s_cur[a] = e[a];
      s_cur[6] = area;
      s_cur[7] = __builtin_bit_cast(float, (int)my_cls);
    }
    __syncthreads();
    if (t < K && t != i && s_alive[t]) {
      const float l = fmaxf(0.f, fminf(s_cur[3], e[3]) - fmaxf(s_cur[0], e[0]));
      const float w = fmaxf(0.f, fminf(s_cur[4], e[4]) - fmaxf(s_cur[1], e[1]));
      const float h = fmaxf(0.f, fminf(s_cur[5], e[5]) - fmaxf(s_cur[2], e[2]));
      const float inter = l * w * h;
      float iou = inter / (s_cur[6] + area - inter);
      iou = iou * ((int)my_cls == __builtin_bit_cast(int, s_cur[7]) ? 1.f : 0.f);
      if (!(iou <= thr)) s_alive[t] = 0;                 // upstream keeps `iou <= thresh`
    }
    __syncthreads();
  }
}

}  // namespace demf

using namespace demf;

extern "C" int demf_box_extent_count(int B, int N, int point_stride, int K, const float* points,
                                     const float* boxes7, const float* cos_yaw, const float* sin_yaw,
                                     float* boxes_bottom, float* extent6, int* count,
                                     demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 0 && point_stride >= 3 && K >= 0, "box_extent_count: bad sizes");
  if (B * K == 0) return DEMF_OK;
  DEMF_REQUIRE(points && boxes7 && cos_yaw && sin_yaw && boxes_bottom && extent6 && count,
               "box_extent_count: null pointer");
  hipLaunchKernelGGL(box_extent_count_k, dim3(K, B), dim3(256), 0, (hipStream_t)stream, N,
                     point_stride, K, points, boxes7, cos_yaw, sin_yaw, boxes_bottom, extent6, count);
  return check_launch("box_extent_count");
}

extern "C" int demf_aligned_nms(int B, int K, float iou_thr, const float* extent6,
                                const float* scores, const int64_t* classes,
                                const unsigned char* valid, unsigned char* keep,
                                demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && K >= 0, "aligned_nms: bad sizes");
  if (K > 1024) {
    set_error("aligned_nms: K=%d boxes per scene exceeds 1024", K);
    return DEMF_EUNSUPPORTED;
  }
  if (B * K == 0) return DEMF_OK;
  DEMF_REQUIRE(extent6 && scores && classes && valid && keep, "aligned_nms: null pointer");
  hipLaunchKernelGGL(aligned_nms_k, dim3(B), dim3(1024), 0, (hipStream_t)stream, K, iou_thr,
                     extent6, scores, (const long long*)classes, valid, keep);
  return check_launch("aligned_nms");
}
