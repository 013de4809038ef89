// Fused DeMF head losses for gfx950.
//
// DeMFVoteHead._loss (demf/modeling/heads/class_agnostic_vote_head.py:622-712) evaluates, per
// decode layer, seven reductions over the B*256 proposals (objectness CE, direction CE,
// direction-residual / size / centre SmoothL1, semantic CE, axis-aligned IoU) plus the vote
// loss over the B*1024 seeds (VoteModule.get_loss, called at :641-644) - upstream that is
// ~100 small kernels forward and more backward, all launch-latency bound.  Here each is ONE
// kernel forward and ONE backward over the raw prediction rows of the conv head:
//   cls row (12) = [objectness 2 | semantic 10]
//   reg row (30) = [centre offset 3 | size 3 | dir class 12 | dir residual (normalised) 12]
// (the slicing of DeMFClassAgnosticBBoxCoder.split_pred, coder.py:196-240, folded in).
// Loss definitions restate mmdet CrossEntropyLoss / SmoothL1Loss and mmdet3d
// AxisAlignedIoULoss with reduction='sum' (configs/demf/demf_votenet.py:116-141).
#include "common.h"

namespace demf {

struct HeadLossCfg {
  int R, ncls, nreg, nbins, nsem;          // rows, 12, 30, 12, 10
  float cw0, cw1;                          // objectness class weights
  float w_obj, w_dircls, w_dirres, w_size, w_center, w_sem, w_iou;
  float beta_dirres, beta_size, beta_center;
};

// order of the 7 outputs: objectness, dir_class, dir_res, size_res, center, semantic, iou
constexpr int NLOSS = 7;

__device__ __forceinline__ float smooth_l1(float d, float beta) {
  const float a = fabsf(d);
  return a < beta ? 0.5f * a * a / beta : a - 0.5f * beta;
}
__device__ __forceinline__ float smooth_l1_grad(float d, float beta) {
  const float a = fabsf(d);
  return a < beta ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
}

// log-softmax cross entropy over n logits at p; returns loss, optionally writes softmax-onehot
template <int MAXN>
__device__ __forceinline__ float ce(const float* p, int n, int t, float* grad /*nullable*/, float gscale) {
  float m = p[0];
  for (int i = 1; i < n; ++i) m = fmaxf(m, p[i]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += __expf(p[i] - m);
  const float lse = m + __logf(s);
  if (grad)
    for (int i = 0; i < n; ++i) grad[i] += gscale * (__expf(p[i] - lse) - (i == t ? 1.f : 0.f));
  return lse - p[t];
}

struct IoU {
  float iou, overlap, uni;
  float lo[3], hi[3];          // intersection bounds
  bool pos[3];                 // intersection extent > 0 per axis
  bool clamped;                // union was clamped to eps
};

__device__ __forceinline__ IoU aa_iou(const float* c, const float* s, const float* ct, const float* st) {
  IoU r;
  float a1 = 1.f, a2 = 1.f, ov = 1.f;
  for (int k = 0; k < 3; ++k) {
    const float l1 = c[k] - s[k] * 0.5f, h1 = c[k] + s[k] * 0.5f;
    const float l2 = ct[k] - st[k] * 0.5f, h2 = ct[k] + st[k] * 0.5f;
    a1 *= (h1 - l1);
    a2 *= (h2 - l2);
    r.lo[k] = fmaxf(l1, l2);
    r.hi[k] = fminf(h1, h2);
    const float w = r.hi[k] - r.lo[k];
    r.pos[k] = w > 0.f;
    ov *= fmaxf(w, 0.f);
  }
  r.overlap = ov;
  const float u = a1 + a2 - ov;
  r.clamped = u < 1e-6f;
  r.uni = r.clamped ? 1e-6f : u;
  r.iou = ov / r.uni;
  return r;
}

// one thread per proposal row; block-reduce the 7 partial sums; fp32 atomics into out[7]
__global__ __launch_bounds__(256) void head_loss_fwd_k(
    HeadLossCfg c, const float* __restrict__ cls, const float* __restrict__ reg,
    const float* __restrict__ base, const float* __restrict__ center_t,
    const float* __restrict__ size_t_, const long long* __restrict__ dircls_t,
    const float* __restrict__ dirres_t, const long long* __restrict__ sem_t,
    const long long* __restrict__ obj_t, const float* __restrict__ obj_w,
    const float* __restrict__ box_w, float* __restrict__ out) {
  __shared__ float red[NLOSS][256];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  float l[NLOSS] = {0, 0, 0, 0, 0, 0, 0};
  if (r < c.R) {
    const float* pc = cls + (size_t)r * c.ncls;
    const float* pr = reg + (size_t)r * c.nreg;
    const float bw = box_w[r];
    const int ot = (int)obj_t[r];
    l[0] = c.w_obj * obj_w[r] * (ot ? c.cw1 : c.cw0) * ce<2>(pc, 2, ot, nullptr, 0.f);
    if (bw != 0.f) {   // every box term carries the weight objectness_target / n_pos
      const int dt = (int)dircls_t[r];
      float cen[3], siz[3];
      for (int k = 0; k < 3; ++k) {
        cen[k] = base[r * 3 + k] + pr[k];
        siz[k] = pr[3 + k];
      }
      l[1] = c.w_dircls * bw * ce<12>(pr + 6, c.nbins, dt, nullptr, 0.f);
      l[2] = c.w_dirres * bw * smooth_l1(pr[6 + c.nbins + dt] - dirres_t[r], c.beta_dirres);
      for (int k = 0; k < 3; ++k) {
        l[3] += c.w_size * bw * smooth_l1(siz[k] - size_t_[r * 3 + k], c.beta_size);
        l[4] += c.w_center * bw * smooth_l1(cen[k] - center_t[r * 3 + k], c.beta_center);
      }
      l[5] = c.w_sem * bw * ce<10>(pc + 2, c.nsem, (int)sem_t[r], nullptr, 0.f);
      const IoU u = aa_iou(cen, siz, center_t + r * 3, size_t_ + r * 3);
      l[6] = c.w_iou * bw * (1.f - u.iou);
    }
  }
#pragma unroll
  for (int i = 0; i < NLOSS; ++i) red[i][threadIdx.x] = l[i];
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if (threadIdx.x < sft)
#pragma unroll
      for (int i = 0; i < NLOSS; ++i) red[i][threadIdx.x] += red[i][threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x < NLOSS) atomicAdd(out + threadIdx.x, red[threadIdx.x][0]);
}

// gradients wrt cls (R,12), reg (R,30), base (R,3) given the 7 upstream scalars gout[7]
__global__ __launch_bounds__(256) void head_loss_bwd_k(
    HeadLossCfg c, const float* __restrict__ cls, const float* __restrict__ reg,
    const float* __restrict__ base, const float* __restrict__ center_t,
    const float* __restrict__ size_t_, const long long* __restrict__ dircls_t,
    const float* __restrict__ dirres_t, const long long* __restrict__ sem_t,
    const long long* __restrict__ obj_t, const float* __restrict__ obj_w,
    const float* __restrict__ box_w, const float* __restrict__ gout,
    float* __restrict__ gcls, float* __restrict__ greg, float* __restrict__ gbase) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= c.R) return;
  const float* pc = cls + (size_t)r * c.ncls;
  const float* pr = reg + (size_t)r * c.nreg;
  float gc[12], gr[30];
  for (int i = 0; i < c.ncls; ++i) gc[i] = 0.f;
  for (int i = 0; i < c.nreg; ++i) gr[i] = 0.f;
  const float bw = box_w[r];
  const int ot = (int)obj_t[r];
  ce<2>(pc, 2, ot, gc, gout[0] * c.w_obj * obj_w[r] * (ot ? c.cw1 : c.cw0));
  if (bw != 0.f) {
    const int dt = (int)dircls_t[r];
    float cen[3], siz[3];
    for (int k = 0; k < 3; ++k) {
      cen[k] = base[r * 3 + k] + pr[k];
      siz[k] = pr[3 + k];
    }
    ce<12>(pr + 6, c.nbins, dt, gr + 6, gout[1] * c.w_dircls * bw);
    gr[6 + c.nbins + dt] += gout[2] * c.w_dirres * bw *
                            smooth_l1_grad(pr[6 + c.nbins + dt] - dirres_t[r], c.beta_dirres);
    for (int k = 0; k < 3; ++k) {
      gr[3 + k] += gout[3] * c.w_size * bw * smooth_l1_grad(siz[k] - size_t_[r * 3 + k], c.beta_size);
      gr[k] += gout[4] * c.w_center * bw * smooth_l1_grad(cen[k] - center_t[r * 3 + k], c.beta_center);
    }
    ce<10>(pc + 2, c.nsem, (int)sem_t[r], gc + 2, gout[5] * c.w_sem * bw);
    // IoU: loss = 1 - ov/u ; d(ov)/d(.) via the intersection extents, d(a1)/d(size)
    const IoU u = aa_iou(cen, siz, center_t + r * 3, size_t_ + r * 3);
    const float gl = -gout[6] * c.w_iou * bw;       // d total / d iou
    // d iou = (d ov * u - ov * d u) / u^2, d u = d a1 - d ov (unless clamped: d u = 0)
    const float inv_u = 1.f / u.uni;
    const float k_ov = gl * (inv_u + (u.clamped ? 0.f : u.overlap * inv_u * inv_u));
    const float k_a1 = u.clamped ? 0.f : -gl * u.overlap * inv_u * inv_u;
    float ext[3], sz[3];
    for (int k = 0; k < 3; ++k) {
      ext[k] = fmaxf(u.hi[k] - u.lo[k], 0.f);
      sz[k] = siz[k];
    }
    for (int k = 0; k < 3; ++k) {
      const float other_ov = ext[(k + 1) % 3] * ext[(k + 2) % 3];
      const float other_a1 = sz[(k + 1) % 3] * sz[(k + 2) % 3];
      float d_hi_c = 0.f, d_hi_s = 0.f, d_lo_c = 0.f, d_lo_s = 0.f;
      if (u.pos[k]) {
        const float l1 = cen[k] - siz[k] * 0.5f, h1 = cen[k] + siz[k] * 0.5f;
        const float l2 = center_t[r * 3 + k] - size_t_[r * 3 + k] * 0.5f;
        const float h2 = center_t[r * 3 + k] + size_t_[r * 3 + k] * 0.5f;
        // torch.min/max backward: on ties the gradient is split evenly between both operands
        const float wh = h1 < h2 ? 1.f : (h1 == h2 ? 0.5f : 0.f);
        const float wl = l1 > l2 ? 1.f : (l1 == l2 ? 0.5f : 0.f);
        d_hi_c = wh; d_hi_s = 0.5f * wh;
        d_lo_c = wl; d_lo_s = -0.5f * wl;
      }
      const float d_ext_c = d_hi_c - d_lo_c, d_ext_s = d_hi_s - d_lo_s;
      const float g_c = k_ov * other_ov * d_ext_c;
      const float g_s = k_ov * other_ov * d_ext_s + k_a1 * other_a1;
      gr[k] += g_c;
      gr[3 + k] += g_s;
    }
  }
  for (int i = 0; i < c.ncls; ++i) gcls[(size_t)r * c.ncls + i] = gc[i];
  for (int i = 0; i < c.nreg; ++i) greg[(size_t)r * c.nreg + i] = gr[i];
  for (int k = 0; k < 3; ++k) gbase[r * 3 + k] = gr[k];   // centre = base + offset
}

// ---- vote loss: sum over seeds of min_j L1(vote - (gt_vote_j + seed)) * mask/(sum mask) * w --
// COUNT (forward only): the number of seeds on a positive point - the denominator, torch.gather(mask, 1,
// seed_indices).sum() in the reference's VoteModule.get_loss - is counted here by every workgroup
// (B*S int64 loads from L2, an exact integer) instead of by three library launches; workgroup 0
// stores it to mask_sum_out for the backward.
template <bool COUNT>
__global__ __launch_bounds__(256) void vote_loss_k(
    int B, int S, int N, int G3 /*3*gt_per_seed*/, float wdst, const float* __restrict__ seed,
    const float* __restrict__ vote, const long long* __restrict__ seed_idx,
    const long long* __restrict__ tmask, const float* __restrict__ vtgt,
    const float* __restrict__ mask_sum, const float* __restrict__ gout /*null: forward*/,
    float* __restrict__ out, float* __restrict__ gvote, float* __restrict__ mask_sum_out) {
  __shared__ float red[256];
  __shared__ long long cnt[256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float msum;
  if constexpr (COUNT) {
    long long c = 0;
    const int total = B * S;
    int j = threadIdx.x;
    for (; j + 7 * 256 < total; j += 8 * 256) {          // 8 independent index -> mask chains in flight
      long long k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) k[u] = seed_idx[j + 256 * u];
#pragma unroll
      for (int u = 0; u < 8; ++u) c += tmask[(size_t)((j + 256 * u) / S) * N + k[u]];
    }
    for (; j < total; j += 256) c += tmask[(size_t)(j / S) * N + seed_idx[j]];
    cnt[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) cnt[threadIdx.x] += cnt[threadIdx.x + s];
      __syncthreads();
    }
    msum = (float)cnt[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) mask_sum_out[0] = msum;
  } else {
    msum = mask_sum[0];
  }
  float l = 0.f;
  if (i < B * S) {
    const int b = i / S;
    const long long k = seed_idx[i];
    const float w = (float)tmask[(size_t)b * N + k] / (msum + 1e-6f) * wdst;
    const float* t = vtgt + ((size_t)b * N + k) * G3;
    float best = 3.0e38f;
    int bj = 0;
    for (int j = 0; j < G3 / 3; ++j) {
      float d = 0.f;
      for (int c = 0; c < 3; ++c) d += fabsf(vote[i * 3 + c] - (t[j * 3 + c] + seed[i * 3 + c])) * w;
      if (d < best) {
        best = d;
        bj = j;
      }
    }
    l = best;
    if (gout) {
      for (int c = 0; c < 3; ++c) {
        const float d = vote[i * 3 + c] - (t[bj * 3 + c] + seed[i * 3 + c]);
        gvote[i * 3 + c] = gout[0] * w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
      }
    }
  }
  if (!gout) {
    red[threadIdx.x] = l;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, red[0]);
  }
}


// ---- vote targets (get_targets_single :828-858, batched) ---------------------------------------
// One thread per point: membership in the (<= 64) ground-truth boxes of its scene, votes towards the
// gravity centres of the first / second / last containing box.  Arithmetic follows the torch
// restatement operation by operation (mul and add kept separate: the library is built with
// -ffp-contract=off), so the result is bit-identical to it; cos/sin of -yaw are passed in.
__global__ __launch_bounds__(256) void vote_targets_k(int N, int pstride, int G,
                                                      const float* __restrict__ points,
                                                      const float* __restrict__ gt,
                                                      const float* __restrict__ cs,
                                                      const float* __restrict__ sn,
                                                      const unsigned char* __restrict__ valid,
                                                      float* __restrict__ vt,
                                                      long long* __restrict__ mask) {
  __shared__ float s_box[64 * 8];   // cx, cy, cz(gravity), hx, hy, hz, cos, sin
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < G; i += blockDim.x) {
    const float* g = gt + ((size_t)b * G + i) * 7;
    float* o = s_box + i * 8;
    o[0] = g[0]; o[1] = g[1];
    o[2] = g[2] + g[5] * 0.5f;
    o[3] = g[3] * 0.5f; o[4] = g[4] * 0.5f; o[5] = g[5] * 0.5f;
    o[6] = cs[(size_t)b * G + i];
    o[7] = valid[(size_t)b * G + i] ? sn[(size_t)b * G + i] : __builtin_nanf("");  // NaN = invalid
  }
  __syncthreads();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* p = points + ((size_t)b * N + n) * pstride;
  const float px = p[0], py = p[1], pz = p[2];
  int total = 0;
  float v1[3] = {0.f, 0.f, 0.f}, v2[3] = {0.f, 0.f, 0.f}, vl[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < G; ++i) {
    const float* o = s_box + i * 8;
    if (o[7] != o[7]) continue;                       // padded slot
    const float r0 = px - o[0], r1 = py - o[1], r2 = pz - o[2];
    const float lx = r0 * o[6] + r1 * o[7];
    const float ly = (-r0) * o[7] + r1 * o[6];
    const bool in = (fabsf(r2) <= o[5]) && (fabsf(lx) < o[3]) && (fabsf(ly) < o[4]);
    if (in) {
      ++total;
      if (total == 1) { v1[0] = -r0; v1[1] = -r1; v1[2] = -r2; }
      if (total == 2) { v2[0] = -r0; v2[1] = -r1; v2[2] = -r2; }
      vl[0] = -r0; vl[1] = -r1; vl[2] = -r2;
    }
  }
  float* o = vt + ((size_t)b * N + n) * 9;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o[c] = total > 0 ? v1[c] : 0.f;
    o[3 + c] = total > 0 ? (total >= 2 ? v2[c] : v1[c]) : 0.f;
    o[6 + c] = total > 0 ? (total >= 3 ? vl[c] : v1[c]) : 0.f;
  }
  mask[(size_t)b * N + n] = total > 0 ? 1 : 0;
}


// ---- proposal targets (get_targets_single :877-934, batched) -----------------------------------
// One thread per proposal: nearest valid ground-truth centre (first minimum), the gathered box
// targets, the canonical-frame distance targets and the objectness label / mask.  Per-box
// quantities that involve transcendental functions (direction class / residual, cos / sin of -yaw)
// are computed by the caller; the arithmetic here mirrors the torch restatement term by term.
struct PropTargetArgs {
  int Q, G, with_rot;
  float pos_thr, neg_thr, res_scale;   // res_scale = pi / num_dir_bins
  const float *agg, *gt, *cs, *sn, *dir_res;
  const long long *dir_cls, *lab;
  const unsigned char* valid;
  float *center_t, *size_t_, *dir_res_t, *dir_t, *dist_t, *obj_mask;
  long long *dir_cls_t, *mask_t, *obj_t;
};

__global__ __launch_bounds__(256) void proposal_targets_k(PropTargetArgs a) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.Q) return;
  const size_t r = (size_t)b * a.Q + q;
  const float px = a.agg[r * 3], py = a.agg[r * 3 + 1], pz = a.agg[r * 3 + 2];
  float best = __builtin_inff();
  int bi = 0;
  for (int g = 0; g < a.G; ++g) {
    const size_t o = (size_t)b * a.G + g;
    if (!a.valid[o]) continue;
    const float* t = a.gt + o * 7;
    const float dx = px - t[0], dy = py - t[1], dz = pz - (t[2] + t[5] * 0.5f);
    const float d = (dx * dx + dy * dy) + dz * dz;
    if (d < best) { best = d; bi = g; }
  }
  const size_t o = (size_t)b * a.G + bi;
  const float* t = a.gt + o * 7;
  const float cx = t[0], cy = t[1], cz = t[2] + t[5] * 0.5f;
  const float euc = sqrtf(best + 1e-6f);
  a.center_t[r * 3] = cx; a.center_t[r * 3 + 1] = cy; a.center_t[r * 3 + 2] = cz;
  a.size_t_[r * 3] = t[3]; a.size_t_[r * 3 + 1] = t[4]; a.size_t_[r * 3 + 2] = t[5];
  a.dir_cls_t[r] = a.dir_cls[o];
  a.dir_res_t[r] = a.dir_res[o] / a.res_scale;
  a.dir_t[r] = t[6];
  a.mask_t[r] = a.lab[o];
  float x = px - cx, y = py - cy;
  const float z = pz - cz;
  if (a.with_rot) {
    const float c = a.cs[o], s = a.sn[o];
    const float xr = x * c + y * s;
    const float yr = (-x) * s + y * c;
    x = xr; y = yr;
  }
  const float hx = t[3] / 2.0f, hy = t[4] / 2.0f, hz = t[5] / 2.0f;
  float* dt = a.dist_t + r * 6;
  dt[0] = hx - x; dt[1] = hy - y; dt[2] = hz - z;
  dt[3] = hx + x; dt[4] = hy + y; dt[5] = hz + z;
  const bool inside = dt[0] >= 0.f && dt[1] >= 0.f && dt[2] >= 0.f && dt[3] >= 0.f &&
                      dt[4] >= 0.f && dt[5] >= 0.f;
  a.obj_t[r] = (euc < a.pos_thr && inside) ? 1 : 0;
  a.obj_mask[r] = (euc < a.pos_thr || euc > a.neg_thr) ? 1.f : 0.f;
}

// ---- ground-truth preparation: everything the two target kernels need that depends on the GT boxes
// alone, in one launch instead of ~20 element-wise ones on (B, G) tensors:
//   cos / sin(-yaw); angle2class(yaw) (PartialBinBasedBBoxCoder: class = bin of the angle shifted
//   by half a bin, residual = offset from the bin centre), in torch's fp32 remainder / floor-divide
//   semantics; valid = label >= 0, label clamped at 0; gravity centre (x, y, z + dz/2).
__device__ __forceinline__ float torch_remainder(float a, float b) {
  float m = fmodf(a, b);
  if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
  return m;
}
__device__ __forceinline__ float torch_floor_divide(float a, float b) {   // ATen div_floor_floating
  float m = fmodf(a, b);
  float d = (a - m) / b;
  if (m != 0.f && ((b < 0.f) != (m < 0.f))) d -= 1.f;
  if (d != 0.f) {
    float f = floorf(d);
    if (d - f > 0.5f) f += 1.f;
    return f;
  }
  return copysignf(0.f, a / b);
}

__global__ void gt_prep_k(int n, int nbins, const float* __restrict__ gt,
                          const long long* __restrict__ labels, float* __restrict__ cs,
                          float* __restrict__ sn, long long* __restrict__ dir_cls,
                          float* __restrict__ dir_res, unsigned char* __restrict__ valid,
                          long long* __restrict__ lab, float* __restrict__ center) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* t = gt + (size_t)i * 7;
  const float yaw = t[6];
  cs[i] = cosf(-yaw);
  sn[i] = sinf(-yaw);
  const float two_pi = (float)(2.0 * 3.14159265358979323846);
  const float per = (float)(2.0 * 3.14159265358979323846 / (double)nbins);
  const float half = (float)(2.0 * 3.14159265358979323846 / (double)nbins / 2.0);
  const float ang = torch_remainder(yaw, two_pi);
  const float shifted = torch_remainder(ang + half, two_pi);
  const float c = torch_floor_divide(shifted, per);
  dir_cls[i] = (long long)c;
  dir_res[i] = shifted - (c * per + half);
  const long long l = labels[i];
  valid[i] = l >= 0 ? 1 : 0;
  lab[i] = l < 0 ? 0 : l;
  center[(size_t)i * 3] = t[0];
  center[(size_t)i * 3 + 1] = t[1];
  center[(size_t)i * 3 + 2] = t[2] + t[5] * 0.5f;
}

// Per-scene ground-truth lists -> static-shape (B, G, 7) boxes + (B, G) labels (-1 on padding slots; an
// empty scene gets the reference's single all-zero fake box with label 0,
// class_agnostic_vote_head.py:766-773).  The per-scene device pointers and box counts travel BY VALUE in
// the kernel arguments (B <= 32), so padding a new batch costs one launch and no host -> device copy: a
// pageable upload on the step's stream makes the host wait for the step in flight.
constexpr int PAD_GT_MAX_B = 32;
struct PadGtArgs {
  const float* boxes[PAD_GT_MAX_B];          // (count_b, box_dim) rows, box_dim >= 7
  const long long* labels[PAD_GT_MAX_B];     // (count_b)
  int count[PAD_GT_MAX_B];
  int box_dim[PAD_GT_MAX_B];
  int B, G;
};
__global__ void pad_gt_k(PadGtArgs a, float* __restrict__ gt, long long* __restrict__ lab,
                         unsigned char* __restrict__ valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.B * a.G) return;
  const int b = i / a.G, g = i - b * a.G;
  const int n = a.count[b];
  float* o = gt + (size_t)i * 7;
  if (g < n) {
    const float* t = a.boxes[b] + (size_t)g * a.box_dim[b];
#pragma unroll
    for (int c = 0; c < 7; ++c) o[c] = t[c];
    lab[i] = a.labels[b][g];
  } else {
#pragma unroll
    for (int c = 0; c < 7; ++c) o[c] = 0.f;
    lab[i] = (n == 0 && g == 0) ? 0 : -1;
  }
  if (valid != nullptr) valid[i] = g < (n > 1 ? n : 1) ? 1 : 0;
}

// position-embedding input of a decoder layer: [base + reg[0:3] | reg[3:6] | 0 0] per proposal row
__global__ void query_pos_rows_k(int R, int nreg, const float* __restrict__ reg, const float* __restrict__ base,
                                 float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* g = reg + (size_t)r * nreg;
  const float* b = base + (size_t)r * 3;
  float4* o = reinterpret_cast<float4*>(out + (size_t)r * 8);
  o[0] = make_float4(b[0] + g[0], b[1] + g[1], b[2] + g[2], g[3]);
  o[1] = make_float4(g[4], g[5], 0.f, 0.f);
}

// objectness_weights = masks / (sum(masks) + 1e-6); box_loss_weights = obj / (sum(obj) + 1e-6)
// (class_agnostic_vote_head.py:797-816), R <= a few thousand proposals: one workgroup.
__global__ __launch_bounds__(1024) void target_weights_k(int R, const float* __restrict__ obj_mask,
                                                         const long long* __restrict__ obj_t,
                                                         float* __restrict__ obj_w,
                                                         float* __restrict__ box_w) {
  __shared__ float s_m[16], s_o[16];
  float m = 0.f, o = 0.f;
  for (int i = threadIdx.x; i < R; i += 1024) { m += obj_mask[i]; o += (float)obj_t[i]; }
  for (int d = 32; d >= 1; d >>= 1) { m += __shfl_xor(m, d); o += __shfl_xor(o, d); }
  if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_o[threadIdx.x >> 6] = o; }
  __syncthreads();
  m = 0.f; o = 0.f;
  for (int w = 0; w < 16; ++w) { m += s_m[w]; o += s_o[w]; }
  const float im = m + 1e-6f, io = o + 1e-6f;
  for (int i = threadIdx.x; i < R; i += 1024) {
    obj_w[i] = obj_mask[i] / im;
    box_w[i] = (float)obj_t[i] / io;
  }
}

// mean of n (<= 4) seven-vectors of per-decode-layer losses and the grand total (+ the vote loss), one thread
struct LossTotalArgs { const float* v[4]; };
__global__ void loss_total_k(int n, LossTotalArgs a, const float* __restrict__ vote, float* __restrict__ out8) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float tot = 0.f;
  for (int i = 0; i < 7; ++i) {
    float m = a.v[0][i];
    for (int j = 1; j < n; ++j) m = m + a.v[j][i];      // the order of the reference's Python sum()
    m = m / (float)n;
    out8[i] = m;
    tot = i == 0 ? m : tot + m;                           // torch's 7-element sum: sequential
  }
  out8[7] = tot + (vote ? vote[0] : 0.f);
}
// gradients: every seven-vector gets (g_mean7[i] + g_total) / n, the vote loss g_total
__global__ void loss_total_bwd_k(int n, const float* __restrict__ g8, float* __restrict__ gv, float* __restrict__ gvote) {
  const int t = threadIdx.x;
  if (t < 7) {
    const float g = (g8[t] + g8[7]) / (float)n;
    for (int j = 0; j < n; ++j) gv[7 * j + t] = g;
  }
  if (t == 7 && gvote) gvote[0] = g8[7];
}

}  // namespace demf


using namespace demf;

static HeadLossCfg make_cfg(int R, const float* hp) {
  HeadLossCfg c;
  c.R = R; c.ncls = 12; c.nreg = 30; c.nbins = 12; c.nsem = 10;
  c.cw0 = hp[0]; c.cw1 = hp[1];
  c.w_obj = hp[2]; c.w_dircls = hp[3]; c.w_dirres = hp[4]; c.w_size = hp[5]; c.w_center = hp[6];
  c.w_sem = hp[7]; c.w_iou = hp[8];
  c.beta_dirres = hp[9]; c.beta_size = hp[10]; c.beta_center = hp[11];
  return c;
}

extern "C" int demf_head_loss_fwd(int R, int num_dir_bins, int num_classes, const float* hyper12,
                                  const float* cls, const float* reg, const float* base_xyz,
                                  const float* center_t, const float* size_t_,
                                  const int64_t* dir_class_t, const float* dir_res_t,
                                  const int64_t* sem_t, const int64_t* obj_t, const float* obj_w,
                                  const float* box_w, float* out7, demf_stream_t stream) {
  DEMF_REQUIRE(R >= 0 && num_dir_bins == 12 && num_classes == 10,
               "head_loss: only the reference layout (12 direction bins, 10 classes) is built");
  if (R == 0) return DEMF_OK;
  DEMF_REQUIRE(hyper12 && cls && reg && base_xyz && center_t && size_t_ && dir_class_t && dir_res_t &&
                   sem_t && obj_t && obj_w && box_w && out7, "head_loss_fwd: null pointer");
  hipLaunchKernelGGL(head_loss_fwd_k, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream,
                     make_cfg(R, hyper12), cls, reg, base_xyz, center_t, size_t_,
                     (const long long*)dir_class_t, dir_res_t, (const long long*)sem_t,
                     (const long long*)obj_t, obj_w, box_w, out7);
  return check_launch("head_loss_fwd");
}

extern "C" int demf_head_loss_bwd(int R, int num_dir_bins, int num_classes, const float* hyper12,
                                  const float* cls, const float* reg, const float* base_xyz,
                                  const float* center_t, const float* size_t_,
                                  const int64_t* dir_class_t, const float* dir_res_t,
                                  const int64_t* sem_t, const int64_t* obj_t, const float* obj_w,
                                  const float* box_w, const float* grad_out7, float* grad_cls,
                                  float* grad_reg, float* grad_base, demf_stream_t stream) {
  DEMF_REQUIRE(R >= 0 && num_dir_bins == 12 && num_classes == 10,
               "head_loss: only the reference layout (12 direction bins, 10 classes) is built");
  if (R == 0) return DEMF_OK;
  DEMF_REQUIRE(hyper12 && cls && reg && base_xyz && center_t && size_t_ && dir_class_t && dir_res_t &&
                   sem_t && obj_t && obj_w && box_w && grad_out7 && grad_cls && grad_reg && grad_base,
               "head_loss_bwd: null pointer");
  hipLaunchKernelGGL(head_loss_bwd_k, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream,
                     make_cfg(R, hyper12), cls, reg, base_xyz, center_t, size_t_,
                     (const long long*)dir_class_t, dir_res_t, (const long long*)sem_t,
                     (const long long*)obj_t, obj_w, box_w, grad_out7, grad_cls, grad_reg, grad_base);
  return check_launch("head_loss_bwd");
}

extern "C" int demf_vote_loss(int B, int S, int N, int gt_per_seed, float dst_weight,
                              const float* seed_points, const float* vote_points,
                              const int64_t* seed_indices, const int64_t* vote_target_masks,
                              const float* vote_targets, const float* mask_sum,
                              const float* grad_out, float* out, float* grad_vote,
                              demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && S >= 0 && N >= 1 && gt_per_seed >= 1, "vote_loss: bad sizes");
  if (B * S == 0) return DEMF_OK;
  DEMF_REQUIRE(seed_points && vote_points && seed_indices && vote_target_masks && vote_targets &&
                   mask_sum && (grad_out ? grad_vote != nullptr : out != nullptr),
               "vote_loss: null pointer");
  hipLaunchKernelGGL(vote_loss_k<false>, dim3(cdiv(B * S, 256)), dim3(256), 0, (hipStream_t)stream, B, S, N,
                     3 * gt_per_seed, dst_weight, seed_points, vote_points,
                     (const long long*)seed_indices, (const long long*)vote_target_masks,
                     vote_targets, mask_sum, grad_out, out, grad_vote, (float*)nullptr);
  return check_launch("vote_loss");
}

extern "C" int demf_vote_loss_fwd(int B, int S, int N, int gt_per_seed, float dst_weight,
                                  const float* seed_points, const float* vote_points,
                                  const int64_t* seed_indices, const int64_t* vote_target_masks,
                                  const float* vote_targets, float* mask_sum_out, float* out,
                                  demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && S >= 0 && N >= 1 && gt_per_seed >= 1, "vote_loss_fwd: bad sizes");
  DEMF_REQUIRE(mask_sum_out && out, "vote_loss_fwd: null pointer");
  if (B * S == 0) return DEMF_OK;
  DEMF_REQUIRE(seed_points && vote_points && seed_indices && vote_target_masks && vote_targets,
               "vote_loss_fwd: null pointer");
  hipLaunchKernelGGL(vote_loss_k<true>, dim3(cdiv(B * S, 256)), dim3(256), 0, (hipStream_t)stream, B, S, N,
                     3 * gt_per_seed, dst_weight, seed_points, vote_points,
                     (const long long*)seed_indices, (const long long*)vote_target_masks,
                     vote_targets, (const float*)nullptr, (const float*)nullptr, out, (float*)nullptr,
                     mask_sum_out);
  return check_launch("vote_loss_fwd");
}

extern "C" int demf_vote_targets(int B, int N, int point_stride, int G, const float* points,
                                 const float* gt_boxes, const float* cos_neg_yaw,
                                 const float* sin_neg_yaw, const unsigned char* valid,
                                 float* vote_targets, int64_t* vote_target_masks,
                                 demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && N >= 0 && point_stride >= 3 && G >= 1 && G <= 64,
               "vote_targets: bad sizes B=%d N=%d stride=%d G=%d (G <= 64)", B, N, point_stride, G);
  if (B * N == 0) return DEMF_OK;
  DEMF_REQUIRE(points && gt_boxes && cos_neg_yaw && sin_neg_yaw && valid && vote_targets &&
                   vote_target_masks, "vote_targets: null pointer");
  hipLaunchKernelGGL(vote_targets_k, dim3(cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, N,
                     point_stride, G, points, gt_boxes, cos_neg_yaw, sin_neg_yaw, valid,
                     vote_targets, (long long*)vote_target_masks);
  return check_launch("vote_targets");
}

extern "C" int demf_proposal_targets(int B, int Q, int G, int with_rot, float pos_thr, float neg_thr,
                                     float res_scale, const float* aggregated_points,
                                     const float* gt_boxes, const float* cos_neg_yaw,
                                     const float* sin_neg_yaw, const int64_t* gt_dir_class,
                                     const float* gt_dir_res, const int64_t* gt_labels,
                                     const unsigned char* valid, float* center_targets,
                                     float* size_targets, int64_t* dir_class_targets,
                                     float* dir_res_targets, float* dir_targets,
                                     int64_t* mask_targets, float* distance_targets,
                                     int64_t* objectness_targets, float* objectness_masks,
                                     demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && Q >= 0 && G >= 1 && res_scale > 0.f, "proposal_targets: bad sizes");
  if (B * Q == 0) return DEMF_OK;
  DEMF_REQUIRE(aggregated_points && gt_boxes && cos_neg_yaw && sin_neg_yaw && gt_dir_class &&
                   gt_dir_res && gt_labels && valid && center_targets && size_targets &&
                   dir_class_targets && dir_res_targets && dir_targets && mask_targets &&
                   distance_targets && objectness_targets && objectness_masks,
               "proposal_targets: null pointer");
  PropTargetArgs a;
  a.Q = Q; a.G = G; a.with_rot = with_rot; a.pos_thr = pos_thr; a.neg_thr = neg_thr;
  a.res_scale = res_scale; a.agg = aggregated_points; a.gt = gt_boxes; a.cs = cos_neg_yaw;
  a.sn = sin_neg_yaw; a.dir_res = gt_dir_res; a.dir_cls = (const long long*)gt_dir_class;
  a.lab = (const long long*)gt_labels; a.valid = valid; a.center_t = center_targets;
  a.size_t_ = size_targets; a.dir_res_t = dir_res_targets; a.dir_t = dir_targets;
  a.dist_t = distance_targets; a.obj_mask = objectness_masks;
  a.dir_cls_t = (long long*)dir_class_targets; a.mask_t = (long long*)mask_targets;
  a.obj_t = (long long*)objectness_targets;
  hipLaunchKernelGGL(proposal_targets_k, dim3(cdiv(Q, 256), B), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("proposal_targets");
}

extern "C" int demf_gt_prep(int B, int G, int num_dir_bins, const float* gt_boxes,
                            const int64_t* labels_padded, float* cos_neg_yaw, float* sin_neg_yaw,
                            int64_t* gt_dir_class, float* gt_dir_res, unsigned char* valid,
                            int64_t* labels_clamped, float* gravity_center, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && G >= 1 && num_dir_bins >= 1, "gt_prep: bad sizes B=%d G=%d", B, G);
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(gt_boxes && labels_padded && cos_neg_yaw && sin_neg_yaw && gt_dir_class && gt_dir_res &&
                   valid && labels_clamped && gravity_center, "gt_prep: null pointer");
  const int n = B * G;
  hipLaunchKernelGGL(gt_prep_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, num_dir_bins,
                     gt_boxes, (const long long*)labels_padded, cos_neg_yaw, sin_neg_yaw,
                     (long long*)gt_dir_class, gt_dir_res, valid, (long long*)labels_clamped,
                     gravity_center);
  return check_launch("gt_prep");
}

extern "C" int demf_pad_gt(int B, int G, const int* counts, const int* box_dims, const void* const* boxes,
                           const void* const* labels, float* gt_padded, int64_t* labels_padded,
                           unsigned char* valid, demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && B <= PAD_GT_MAX_B && G >= 1, "pad_gt: bad sizes B=%d (max %d) G=%d", B, PAD_GT_MAX_B, G);
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(counts && box_dims && boxes && labels && gt_padded && labels_padded, "pad_gt: null pointer");
  PadGtArgs a{};
  a.B = B; a.G = G;
  for (int b = 0; b < B; ++b) {
    DEMF_REQUIRE(counts[b] >= 0 && counts[b] <= G, "pad_gt: scene %d has %d boxes, the padded form holds %d", b, counts[b], G);
    DEMF_REQUIRE(counts[b] == 0 || (boxes[b] && labels[b] && box_dims[b] >= 7), "pad_gt: scene %d: null list / box_dim < 7", b);
    a.boxes[b] = (const float*)boxes[b];
    a.labels[b] = (const long long*)labels[b];
    a.count[b] = counts[b];
    a.box_dim[b] = box_dims[b];
  }
  const int n = B * G;
  hipLaunchKernelGGL(pad_gt_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, gt_padded,
                     (long long*)labels_padded, valid);
  return check_launch("pad_gt");
}

extern "C" int demf_query_pos_rows(int R, int nreg, const float* reg_rows, const float* base_xyz,
                                   float* out8, demf_stream_t stream) {
  DEMF_REQUIRE(R >= 0 && nreg >= 6, "query_pos_rows: bad sizes R=%d nreg=%d", R, nreg);
  if (R == 0) return DEMF_OK;
  DEMF_REQUIRE(reg_rows && base_xyz && out8, "query_pos_rows: null pointer");
  hipLaunchKernelGGL(query_pos_rows_k, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, R, nreg, reg_rows,
                     base_xyz, out8);
  return check_launch("query_pos_rows");
}

extern "C" int demf_target_weights(int R, const float* objectness_masks,
                                   const int64_t* objectness_targets, float* objectness_weights,
                                   float* box_loss_weights, demf_stream_t stream) {
  DEMF_REQUIRE(R >= 0, "target_weights: bad size");
  if (R == 0) return DEMF_OK;
  DEMF_REQUIRE(objectness_masks && objectness_targets && objectness_weights && box_loss_weights,
               "target_weights: null pointer");
  hipLaunchKernelGGL(target_weights_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, R, objectness_masks,
                     (const long long*)objectness_targets, objectness_weights, box_loss_weights);
  return check_launch("target_weights");
}

extern "C" int demf_loss_total(int n, const float* const* vecs, const float* vote, float* out8, demf_stream_t stream) {
  DEMF_REQUIRE(n >= 1 && n <= 4 && vecs && out8, "loss_total: 1..4 seven-vectors");
  LossTotalArgs a{};
  for (int i = 0; i < n; ++i) {
    DEMF_REQUIRE(vecs[i], "loss_total: null vector %d", i);
    a.v[i] = vecs[i];
  }
  hipLaunchKernelGGL(loss_total_k, dim3(1), dim3(64), 0, (hipStream_t)stream, n, a, vote, out8);
  return check_launch("loss_total");
}

extern "C" int demf_loss_total_bwd(int n, const float* g8, float* gvecs, float* gvote, demf_stream_t stream) {
  DEMF_REQUIRE(n >= 1 && n <= 4 && g8 && gvecs, "loss_total_bwd: 1..4 seven-vectors");
  hipLaunchKernelGGL(loss_total_bwd_k, dim3(1), dim3(64), 0, (hipStream_t)stream, n, g8, gvecs, gvote);
  return check_launch("loss_total_bwd");
}
