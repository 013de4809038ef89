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


// ---- optimizer step with DEVICE-resident state -------------------------------------------------
// The step count, the learning-rate factor and the squared gradient norm live on the device, so the
// whole update (norm + clip + AdamW) is capturable in the step's hipGraph: no host argument changes
// between replays.  64 bytes:
//   double sumsq        accumulated by multi_copy_sumsq_k / sumsq_k, cleared by the AdamW launch
//   long long t         completed optimizer steps
//   unsigned ticket     last-workgroup detection of the AdamW launch (atomics only, no fence)
//   float lr_factor     multiplies every group's base learning rate (step schedule)
struct OptState {
  double sumsq;
  long long t;
  unsigned ticket;
  float lr_factor;
};

__device__ __forceinline__ void block_add_sumsq(float acc, OptState* st) {
  __shared__ double part[4];
  double d = (double)acc;
  for (int off = 32; off; off >>= 1) d += __shfl_xor(d, off);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) part[w] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = part[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) s += part[i];
    if (s != 0.0) atomicAdd(&st->sumsq, s);
  }
}

// multi_copy_k + the sum of squares of everything copied (fp32 words), added to st->sumsq: the
// gradient pack of the captured step takes the clip norm on the way.
__global__ __launch_bounds__(256) void multi_copy_sumsq_k(int n, const long long* __restrict__ table,
                                                          OptState* st) {
  const int seg = blockIdx.y;
  const float* src = reinterpret_cast<const float*>(table[seg]);
  float* dst = reinterpret_cast<float*>(table[n + seg]);
  const long long words = table[2 * n + seg];
  const bool vec = ((table[seg] | table[n + seg]) & 15) == 0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  if (src == nullptr) {
    for (long long j = i; j < words; j += stride) dst[j] = 0.f;
  } else if (vec) {
    const long long q = words >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (long long j = i; j < q; j += stride) {
      const float4 x = s4[j];
      acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      d4[j] = x;
    }
    for (long long j = 4 * q + i; j < words; j += stride) {
      const float x = src[j];
      acc += x * x;
      dst[j] = x;
    }
  } else {
    for (long long j = i; j < words; j += stride) {
      const float x = src[j];
      acc += x * x;
      dst[j] = x;
    }
  }
  block_add_sumsq(acc, st);
}

__global__ __launch_bounds__(256) void sumsq_k(long long n, const float* __restrict__ x, OptState* st) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += x[i] * x[i];
  block_add_sumsq(acc, st);
}

struct AdamWSeg {
  long long start, n;
  float lr, weight_decay;
  unsigned block0;   // first workgroup of the segment
};

struct AdamWStateArgs {
  int nseg;
  AdamWSeg seg[4];
  float* p;
  const float* g;
  float* m;
  float* v;
  OptState* st;
  float max_norm, grad_scale, beta1, beta2, eps;
};

// Every parameter group in ONE launch.  t = st->t + 1; clip coefficient from st->sumsq; the last
// workgroup to finish (ticket) publishes t, clears sumsq and the ticket for the next step.
__global__ __launch_bounds__(256) void adamw_state_k(AdamWStateArgs a) {
  __shared__ float sh[4];
  OptState* st = a.st;
  int s = 0;
  for (int k = 1; k < a.nseg; ++k)
    if (blockIdx.x >= a.seg[k].block0) s = k;
  const AdamWSeg sg = a.seg[s];
  if (threadIdx.x == 0) {
    const long long t = st->t + 1;
    const float lr = sg.lr * st->lr_factor;
    float coef = a.grad_scale;
    if (a.max_norm > 0.f) {
      const float norm = (float)sqrt(st->sumsq);
      const float c = a.max_norm / (norm * a.grad_scale + 1e-6f);
      coef *= c < 1.f ? c : 1.f;
    }
    sh[0] = coef;
    sh[1] = 1.f - lr * sg.weight_decay;
    sh[2] = lr / (float)(1.0 - pow((double)a.beta1, (double)t));
    sh[3] = (float)sqrt(1.0 - pow((double)a.beta2, (double)t));
  }
  __syncthreads();
  const float coef = sh[0], decay = sh[1], step = sh[2], bias_c2_sqrt = sh[3];
  const unsigned nblk = (s + 1 < a.nseg ? a.seg[s + 1].block0 : gridDim.x) - sg.block0;
  const long long stride = (long long)nblk * blockDim.x * 4;
  float* P = a.p + sg.start;
  const float* G = a.g + sg.start;
  float* M = a.m + sg.start;
  float* V = a.v + sg.start;
  const bool vec = (((size_t)P | (size_t)G | (size_t)M | (size_t)V) & 15) == 0;
  for (long long i = ((long long)(blockIdx.x - sg.block0) * blockDim.x + threadIdx.x) * 4; i < sg.n; i += stride) {
    if (vec && i + 4 <= sg.n) {
      float4 p = *reinterpret_cast<float4*>(P + i);
      const float4 g4 = *reinterpret_cast<const float4*>(G + i);
      float4 m = *reinterpret_cast<float4*>(M + i);
      float4 v = *reinterpret_cast<float4*>(V + i);
      float* pp = &p.x; const float* gg = &g4.x; float* mm = &m.x; float* vv = &v.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = gg[k] * coef;
        pp[k] *= decay;
        mm[k] = a.beta1 * mm[k] + (1.f - a.beta1) * g;
        vv[k] = a.beta2 * vv[k] + (1.f - a.beta2) * g * g;
        pp[k] -= step * (mm[k] / (sqrtf(vv[k]) / bias_c2_sqrt + a.eps));
      }
      *reinterpret_cast<float4*>(P + i) = p;
      *reinterpret_cast<float4*>(M + i) = m;
      *reinterpret_cast<float4*>(V + i) = v;
    } else {
      for (long long j = i; j < sg.n && j < i + 4; ++j) {
        const float g = G[j] * coef;
        float p = P[j] * decay;
        const float m = a.beta1 * M[j] + (1.f - a.beta1) * g;
        const float v = a.beta2 * V[j] + (1.f - a.beta2) * g * g;
        p -= step * (m / (sqrtf(v) / bias_c2_sqrt + a.eps));
        P[j] = p; M[j] = m; V[j] = v;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // every workgroup read t / sumsq before it took its ticket: the last one may overwrite them
    const unsigned ticket = atomicAdd(&st->ticket, 1u);
    if (ticket == gridDim.x - 1) {
      st->t = st->t + 1;
      st->sumsq = 0.0;
      st->ticket = 0u;
    }
  }
}


// x[0..n) = 0 (float4 body).  The step's zero arena is cleared by this launch: a hipMemsetAsync node is not
// reliably ordered with the kernel nodes of a captured graph on ROCm 7.2.
__global__ __launch_bounds__(256) void zero_f32_k(long long n, float* __restrict__ x) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if ((((size_t)x) & 15) == 0) {
    const long long q = n >> 2;
    float4* x4 = reinterpret_cast<float4*>(x);
    for (long long i = i0; i < q; i += stride) x4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = 4 * q + i0; i < n; i += stride) x[i] = 0.f;
  } else {
    for (long long i = i0; i < n; i += stride) x[i] = 0.f;
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

extern "C" int demf_multi_copy_sumsq(int n, const void* table, int blocks_per_segment, void* opt_state,
                                     demf_stream_t stream) {
  DEMF_REQUIRE(n >= 0 && blocks_per_segment >= 1, "multi_copy_sumsq: bad arguments");
  if (n == 0) return DEMF_OK;
  DEMF_REQUIRE(table != nullptr && opt_state != nullptr, "multi_copy_sumsq: null pointer");
  hipLaunchKernelGGL(multi_copy_sumsq_k, dim3(blocks_per_segment, n), dim3(256), 0, (hipStream_t)stream, n,
                     (const long long*)table, (OptState*)opt_state);
  return check_launch("multi_copy_sumsq");
}

extern "C" int demf_sumsq_f32(long long n, const float* x, void* opt_state, demf_stream_t stream) {
  if (n <= 0) return DEMF_OK;
  DEMF_REQUIRE(x && opt_state, "sumsq: null pointer");
  long long blocks = (n + 2047) / 2048;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sumsq_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, x,
                     (OptState*)opt_state);
  return check_launch("sumsq");
}

extern "C" int demf_adamw_state_f32(int nseg, const long long* seg_start, const long long* seg_n,
                                    const float* seg_lr, const float* seg_weight_decay, float* param,
                                    const float* grad, float* exp_avg, float* exp_avg_sq, void* opt_state,
                                    float max_norm, float grad_scale, float beta1, float beta2, float eps,
                                    demf_stream_t stream) {
  DEMF_REQUIRE(nseg >= 1 && nseg <= 4, "adamw_state: %d parameter groups (1..4 supported)", nseg);
  DEMF_REQUIRE(seg_start && seg_n && seg_lr && seg_weight_decay, "adamw_state: null segment table");
  DEMF_REQUIRE(param && grad && exp_avg && exp_avg_sq && opt_state, "adamw_state: null pointer");
  AdamWStateArgs a;
  a.nseg = nseg;
  unsigned blocks = 0;
  for (int s = 0; s < nseg; ++s) {
    DEMF_REQUIRE(seg_n[s] > 0 && seg_start[s] >= 0, "adamw_state: empty parameter group %d", s);
    a.seg[s].start = seg_start[s];
    a.seg[s].n = seg_n[s];
    a.seg[s].lr = seg_lr[s];
    a.seg[s].weight_decay = seg_weight_decay[s];
    a.seg[s].block0 = blocks;
    long long b = (seg_n[s] + 1023) / 1024;
    if (b > 1024) b = 1024;
    blocks += (unsigned)b;
  }
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.st = (OptState*)opt_state;
  a.max_norm = max_norm; a.grad_scale = grad_scale; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  hipLaunchKernelGGL(adamw_state_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("adamw_state");
}

extern "C" int demf_zero_f32(long long n, float* x, demf_stream_t stream) {
  if (n <= 0) return DEMF_OK;
  DEMF_REQUIRE(x != nullptr, "zero_f32: null pointer");
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_f32_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, x);
  return check_launch("zero_f32");
}
