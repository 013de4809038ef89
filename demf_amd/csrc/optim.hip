// Flat AdamW step with the gradient-clip coefficient folded in.
//
// The reference trains with torch.optim.AdamW(lr=0.008, weight_decay=0.01) through mmcv's
// OptimizerHook(grad_clip=dict(max_norm=10, norm_type=2)) and a paramwise lr_mult of 0.05 on
// 'decoder' parameters (configs/_base_/schedules/schedule_3x.py:6-7, configs/demf/demf_votenet.py:16-24).
// Here parameters, gradients and both moments are flat fp32 buffers, so one launch per parameter
// group replaces the 22 multi-tensor launches of the foreach implementation.  HBM-bound:
// 7 floats of traffic per parameter (read p,g,m,v; write p,m,v).
#include <hip/hip_runtime.h>

#include "common.h"

namespace demf {

struct AdamWArgs {
  long long n;
  float* p;
  const float* g;
  float* m;
  float* v;
  const float* norm;  // device scalar: ||g||_2 over ALL groups, or null (no clipping)
  float max_norm, grad_scale;
  float lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2_sqrt;
};

__global__ __launch_bounds__(256) void adamw_flat_k(AdamWArgs a) {
  float coef = a.grad_scale;
  if (a.norm) {
    // torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (total_norm + 1e-6), max=1)
    const float c = a.max_norm / (*a.norm * a.grad_scale + 1e-6f);
    coef *= c < 1.f ? c : 1.f;
  }
  const float decay = 1.f - a.lr * a.weight_decay;
  const float step = a.lr / a.bias_c1;
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < a.n; i += stride) {
    if (i + 4 <= a.n && (((size_t)(a.p + i) | (size_t)(a.g + i) | (size_t)(a.m + i) |
                          (size_t)(a.v + i)) & 15) == 0) {
      float4 p = *reinterpret_cast<float4*>(a.p + i);
      const float4 g4 = *reinterpret_cast<const float4*>(a.g + i);
      float4 m = *reinterpret_cast<float4*>(a.m + i);
      float4 v = *reinterpret_cast<float4*>(a.v + i);
      float* pp = &p.x; const float* gg = &g4.x; float* mm = &m.x; float* vv = &v.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = gg[k] * coef;
        pp[k] *= decay;
        mm[k] = a.beta1 * mm[k] + (1.f - a.beta1) * g;
        vv[k] = a.beta2 * vv[k] + (1.f - a.beta2) * g * g;
        pp[k] -= step * (mm[k] / (sqrtf(vv[k]) / a.bias_c2_sqrt + a.eps));
      }
      *reinterpret_cast<float4*>(a.p + i) = p;
      *reinterpret_cast<float4*>(a.m + i) = m;
      *reinterpret_cast<float4*>(a.v + i) = v;
    } else {
      for (long long j = i; j < a.n && j < i + 4; ++j) {
        const float g = a.g[j] * coef;
        float p = a.p[j] * decay;
        const float m = a.beta1 * a.m[j] + (1.f - a.beta1) * g;
        const float v = a.beta2 * a.v[j] + (1.f - a.beta2) * g * g;
        p -= step * (m / (sqrtf(v) / a.bias_c2_sqrt + a.eps));
        a.p[j] = p; a.m[j] = m; a.v[j] = v;
      }
    }
  }
}

// n segments copied (src != null) or zero-filled (src == null) in ONE launch: blockIdx.y = segment,
// blockIdx.x strides over its 4-byte words.  table (device, 3 x n int64): src | dst | words.
__global__ __launch_bounds__(256) void multi_copy_k(int n, const long long* __restrict__ table) {
  const int seg = blockIdx.y;
  const unsigned* src = reinterpret_cast<const unsigned*>(table[seg]);
  unsigned* dst = reinterpret_cast<unsigned*>(table[n + seg]);
  const long long words = table[2 * n + seg];
  const bool vec = ((table[seg] | table[n + seg]) & 15) == 0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const long long q = words >> 2;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (long long j = i; j < q; j += stride) d4[j] = src != nullptr ? s4[j] : make_uint4(0, 0, 0, 0);
    for (long long j = 4 * q + i; j < words; j += stride) dst[j] = src != nullptr ? src[j] : 0u;
  } else {
    for (long long j = i; j < words; j += stride) dst[j] = src != nullptr ? src[j] : 0u;
  }
}

}  // namespace demf

using namespace demf;

extern "C" int demf_multi_copy(int n, const void* table, int blocks_per_segment, demf_stream_t stream) {
  DEMF_REQUIRE(n >= 0 && blocks_per_segment >= 1, "multi_copy: bad arguments");
  if (n == 0) return DEMF_OK;
  DEMF_REQUIRE(table != nullptr, "multi_copy: null table");
  hipLaunchKernelGGL(multi_copy_k, dim3(blocks_per_segment, n), dim3(256), 0, (hipStream_t)stream, n,
                     (const long long*)table);
  return check_launch("multi_copy");
}

extern "C" int demf_adamw_f32(long long n, float* param, const float* grad, float* exp_avg,
                              float* exp_avg_sq, const float* grad_norm, float max_norm,
                              float grad_scale, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, demf_stream_t stream) {
  if (n <= 0) return DEMF_OK;
  DEMF_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adamw: null pointer");
  DEMF_REQUIRE(step >= 1, "adamw: step=%d must be >= 1", step);
  AdamWArgs a;
  a.n = n; a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.norm = grad_norm;
  a.max_norm = max_norm; a.grad_scale = grad_scale;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  a.bias_c1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bias_c2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  long long blocks = (n + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_flat_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("adamw");
}
