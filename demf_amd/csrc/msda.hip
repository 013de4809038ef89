// Multi-scale deformable attention core (forward + backward) for gfx950.
//
// This is the DeMF fusion gather: the 256 vote-cluster queries sample the 4-level
// image pyramid at learned offsets.  Reference call chain:
//   DeMFVoteHead.transformer_decoder (class_agnostic_vote_head.py:493)
//   -> DeMFTransformerDecoderLayer.forward (demf/modeling/layers/transformer.py:73)
//   -> mmcv MultiScaleDeformableAttention -> MultiScaleDeformableAttnFunction
//      (ms_deform_attn im2col / col2im CUDA kernels upstream).
//
// Mapping for wave64: one (batch, query, head) item is served by G = Dh/4 adjacent
// lanes, each owning 4 consecutive channels, so a bilinear corner is ONE
// Dh*4-byte contiguous read split into float4 lanes (Dh=32: 8 lanes x 16 B = a
// full 128-byte line) and a wave covers 64/G heads of consecutive queries; the
// (B,Q,H*Dh) output row is written as coalesced float4.  The sampling loop is
// branch-free (clamped addresses, zeroed corner weights) so all L*P*4 corner loads
// of an item are in flight together - the kernel is a latency/L2-gather problem,
// not an ALU one.  Backward reduces grad_loc / grad_weight across the G lanes with
// DPP adds (no LDS) and scatters grad_value with hardware fp32 atomics.
#include "common.h"
#include <stdlib.h>

namespace demf {

struct Corner {
  int off[4];    // element offsets (without channel) of the 4 corners, clamped
  float cw[4];   // bilinear weights, zero for out-of-image corners
  float ok[4];   // 1/0 validity of each corner
  float lh, lw, hh, hw;
  bool inside;
};

__device__ __forceinline__ Corner make_corner(float lx, float ly, int Hl, int Wl,
                                              int start, int H, int Dh, int h) {
  Corner c;
  const float h_im = ly * (float)Hl - 0.5f;
  const float w_im = lx * (float)Wl - 0.5f;
  c.inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hl) && (w_im < (float)Wl);
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = (int)hf, w_low = (int)wf;
  const int h_high = h_low + 1, w_high = w_low + 1;
  c.lh = h_im - hf;
  c.lw = w_im - wf;
  c.hh = 1.f - c.lh;
  c.hw = 1.f - c.lw;
  const bool hl = c.inside && h_low >= 0, hh_ = c.inside && h_high <= Hl - 1;
  const bool wl = w_low >= 0, wh = w_high <= Wl - 1;
  c.ok[0] = (hl && wl) ? 1.f : 0.f;
  c.ok[1] = (hl && wh) ? 1.f : 0.f;
  c.ok[2] = (hh_ && wl) ? 1.f : 0.f;
  c.ok[3] = (hh_ && wh) ? 1.f : 0.f;
  c.cw[0] = c.hh * c.hw * c.ok[0];
  c.cw[1] = c.hh * c.lw * c.ok[1];
  c.cw[2] = c.lh * c.hw * c.ok[2];
  c.cw[3] = c.lh * c.lw * c.ok[3];
  const int hlc = min(max(h_low, 0), Hl - 1), hhc = min(max(h_high, 0), Hl - 1);
  const int wlc = min(max(w_low, 0), Wl - 1), whc = min(max(w_high, 0), Wl - 1);
  const int hd = H * Dh;
  c.off[0] = (start + hlc * Wl + wlc) * hd + h * Dh;
  c.off[1] = (start + hlc * Wl + whc) * hd + h * Dh;
  c.off[2] = (start + hhc * Wl + wlc) * hd + h * Dh;
  c.off[3] = (start + hhc * Wl + whc) * hd + h * Dh;
  return c;
}

__device__ __forceinline__ float4 f4_fma(float s, float4 v, float4 a) {
  a.x = __builtin_fmaf(s, v.x, a.x);
  a.y = __builtin_fmaf(s, v.y, a.y);
  a.z = __builtin_fmaf(s, v.z, a.z);
  a.w = __builtin_fmaf(s, v.w, a.w);
  return a;
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b) {
  return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}

constexpr int MAX_L = 8;

// four consecutive channels of a value row: fp32 rows, or bf16 rows (2 bytes per element: the gather moves
// half the bytes; demf_msda_{fwd,bwd}_bf16) widened exactly
template <typename VT>
__device__ __forceinline__ float4 ld4(const VT* p);
template <>
__device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 ld4<uint16_t>(const uint16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}
template <typename VT>
__device__ __forceinline__ float ld1(const VT* p);
template <>
__device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld1<uint16_t>(const uint16_t* p) { return __uint_as_float((unsigned)*p << 16); }

// TL,TP > 0: compile-time levels/points (fully unrolled); 0: runtime loop.
template <int G, int TL, int TP, typename VT = float>
__global__ __launch_bounds__(256) void msda_fwd_kernel(
    int S, int H, int Dh, int L, int Q, int P, const VT* __restrict__ value,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attw, float* __restrict__ out,
    long long items) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long item = t / G;
  if (item >= items) return;
  const int sub = (int)(t - item * G);
  const int h = (int)(item % H);
  const int b = (int)(item / ((long long)Q * H));
  const VT* vb = value + (size_t)b * S * H * Dh + sub * 4;
  if constexpr (TL > 0) {
    L = TL;
    P = TP;
  }
  const int nlp = L * P;
  const float* lp = loc + item * nlp * 2;
  const float* wp = attw + item * nlp;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < nlp; ++i) {
    const int l = i / P;
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const int start = (int)lsi[l];
    const Corner c = make_corner(lp[2 * i], lp[2 * i + 1], Hl, Wl, start, H, Dh, h);
    const float aw = wp[i];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      v = f4_fma(c.cw[k], ld4<VT>(vb + c.off[k]), v);
    acc = f4_fma(aw, v, acc);
  }
  *reinterpret_cast<float4*>(out + item * Dh + sub * 4) = acc;
}

// The encoder's self-attention form (DeformableDetrEncoder, demf/modeling/layers/deform_detr_encoder.py:68-154 ->
// mmcv MultiScaleDeformableAttention.forward): sampling offsets and attention LOGITS arrive raw, as columns of the
// projection GEMM's output row (csrc/rows_gemm.hip), the projected value as further columns of the same rows
// (pitch vpitch floats).  The kernel does what upstream does in separate elementwise launches on the way in:
//   weights = softmax over the L*P logits of a head,   loc = ref[level] + offset / (W_level, H_level)
// (two (R, 384)-float tensors never exist).  G lanes per (query, head), 4 channels per lane, as msda_fwd_kernel.
template <int G, int TL, int TP>
__global__ __launch_bounds__(256) void msda_fwd_raw_kernel(
    int S, int H, int Dh, int Q, const float* __restrict__ value, long long vpitch,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ raw, long long ldraw, int off_col0, int lgt_col0,
    const float* __restrict__ ref, float* __restrict__ out, long long items) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long item = t / G;
  if (item >= items) return;
  const int sub = (int)(t - item * G);
  const int h = (int)(item % H);
  const long long row = item / H;                     // b * Q + q
  const int b = (int)(row / Q);
  constexpr int nlp = TL * TP;
  const int vh = (int)(vpitch / Dh);                  // row pitch of the value columns, in heads
  const float* vb = value + (size_t)b * S * vpitch + sub * 4;
  const float* op = raw + row * ldraw + off_col0 + h * nlp * 2;
  const float* gp = raw + row * ldraw + lgt_col0 + h * nlp;
  const float* rp = ref + row * TL * 2;
  float lg[nlp], mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < nlp; ++i) {
    lg[i] = gp[i];
    mx = fmaxf(mx, lg[i]);
  }
  float den = 0.f;
#pragma unroll
  for (int i = 0; i < nlp; ++i) {
    lg[i] = expf(lg[i] - mx);
    den += lg[i];
  }
  const float inv = 1.f / den;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < nlp; ++i) {
    const int l = i / TP;
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const int start = (int)lsi[l];
    const float lx = rp[2 * l] + op[2 * i] / (float)Wl;
    const float ly = rp[2 * l + 1] + op[2 * i + 1] / (float)Hl;
    const Corner c = make_corner(lx, ly, Hl, Wl, start, vh, Dh, h);
    const float aw = lg[i] * inv;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      v = f4_fma(c.cw[k], ld4<float>(vb + c.off[k]), v);
    acc = f4_fma(aw, v, acc);
  }
  *reinterpret_cast<float4*>(out + item * Dh + sub * 4) = acc;
}

// One WAVE per query for the shape the encoder runs (8 heads x 32 channels: lane = head * 8 + channel quad).  In
// msda_fwd_raw_kernel each of the 8 lanes of a head redoes the head's softmax and all L*P corner set-ups (~45 VALU
// operations per point, 16 points: two thirds of the kernel's instruction stream).  Here the 8 * L*P (head, point)
// pairs of the query are dealt one or two per lane - coalesced reads of the logits / offsets row -, the softmax is a
// reduction over the L*P lanes of a head, every pair's four corner offsets and four (bilinear x attention) weights
// are computed ONCE and passed to the head's lanes through 4 KB of LDS per wave ([point][head] 16-byte slots:
// the 8 heads' slots of a point are 128 contiguous bytes, a broadcast read without bank conflicts).
// 540 -> 465 us at 8 x 18 609 queries, P = 4 (what is left is the gather itself: 9.8 GB of 16-byte loads; dealing
// each XCD a contiguous eighth of the queries instead of every eighth workgroup measured no better, 483 us).
template <int TL, int TP, bool MSDA_NT = false>
__global__ __launch_bounds__(256) void msda_fwd_raw_wave_kernel(
    int S, int Q, const float* __restrict__ value, long long vpitch,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ raw, long long ldraw, int off_col0, int lgt_col0,
    const float* __restrict__ ref, float* __restrict__ out, long long rows, int xcd_blocks) {
  constexpr int H = 8, Dh = 32, NP = TL * TP, PPL = NP / 8;      // pairs per lane: 2 (P = 4) or 1 (P = 2)
  __shared__ int4 s_off[4][NP * H];
  __shared__ float4 s_w[4][NP * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long blk = blockIdx.x;
  if (xcd_blocks) {       // XCD x (= workgroup id mod 8) takes the x-th contiguous eighth of the query blocks
    const unsigned per = (gridDim.x + 7) / 8;
    blk = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  const long long row = blk * 4 + wave;                            // b * Q + q
  if (row >= rows) return;                                         // (wave-uniform; no workgroup barrier below)
  const int b = (int)(row / Q);
  const int vh = (int)(vpitch / Dh);
  const float* rrow = raw + row * ldraw;
#pragma unroll
  for (int s = 0; s < PPL; ++s) {
    const int pi = lane + 64 * s, h = pi / NP, i = pi - h * NP, l = i / TP;
    // (read-once rows: non-temporal, so that they do not push the value rows - the data with reuse - out of L2)
    const float lg = MSDA_NT ? __builtin_nontemporal_load(rrow + lgt_col0 + pi) : rrow[lgt_col0 + pi];
    float2 of;
    if (MSDA_NT) {
      of.x = __builtin_nontemporal_load(rrow + off_col0 + 2 * pi);
      of.y = __builtin_nontemporal_load(rrow + off_col0 + 2 * pi + 1);
    } else {
      of = *reinterpret_cast<const float2*>(rrow + off_col0 + 2 * pi);
    }
    float mx = lg;
#pragma unroll
    for (int o = NP / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = expf(lg - mx);
    float den = e;
#pragma unroll
    for (int o = NP / 2; o > 0; o >>= 1) den += __shfl_xor(den, o);
    const float aw = e / den;
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const float2 rp = *reinterpret_cast<const float2*>(ref + row * (TL * 2) + 2 * l);
    const Corner c = make_corner(rp.x + of.x / (float)Wl, rp.y + of.y / (float)Hl, Hl, Wl, (int)lsi[l], vh, Dh, h);
    s_off[wave][i * H + h] = make_int4(c.off[0], c.off[1], c.off[2], c.off[3]);
    s_w[wave][i * H + h] = make_float4(c.cw[0] * aw, c.cw[1] * aw, c.cw[2] * aw, c.cw[3] * aw);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave's own LDS writes (in order, one wave)
  __builtin_amdgcn_wave_barrier();
  const int h = lane >> 3, sub = lane & 7;
  const float* vb = value + (size_t)b * S * vpitch + sub * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int4 o = s_off[wave][i * H + h];
    const float4 w = s_w[wave][i * H + h];
    acc = f4_fma(w.x, ld4<float>(vb + o.x), acc);
    acc = f4_fma(w.y, ld4<float>(vb + o.y), acc);
    acc = f4_fma(w.z, ld4<float>(vb + o.z), acc);
    acc = f4_fma(w.w, ld4<float>(vb + o.w), acc);
  }
  if (MSDA_NT) {
    float* o = out + row * (H * Dh) + lane * 4;
    __builtin_nontemporal_store(acc.x, o); __builtin_nontemporal_store(acc.y, o + 1);
    __builtin_nontemporal_store(acc.z, o + 2); __builtin_nontemporal_store(acc.w, o + 3);
  } else {
    *reinterpret_cast<float4*>(out + row * (H * Dh) + lane * 4) = acc;
  }
}

// (scene, head)-major form of the above with the COARSE levels' value rows resident in LDS (round 5).  The gather is
// bound by the vector memory path: 64 KB of 16-byte loads per query, ~1 000 TA cycles, 250 us per encoder layer at the
// very best and 400-415 measured.  A head's slice of levels LS .. 3 - 875 + 234 tokens x 128 bytes = 139 KB at the
// reference's pyramid - fits a CU's 160 KB: a workgroup takes one (scene, head) and a share of the queries, stages that
// slice once, and serves half of every query's samples from LDS (128 B/clk beside the 64 B/clk of the vector cache).
// One wave per SIMD (the slice leaves room for four waves' exchange buffers only): the memory parallelism comes from
// inside the wave - all global corner loads of eight queries' fine-level samples are requested before the LDS half is
// summed.  Lane = (query of the wave's eight, channel quad); a lane sets up one or two (level, point) pairs of its
// query's head, the group's softmax is three shuffles, offsets / weights go round through 4 KB of LDS per wave.
constexpr int MSDA_HW = 8;          // waves per workgroup of msda_fwd_raw_head_kernel (20 bytes of exchange slot per sample)
template <int TP, int LS>
__global__ __launch_bounds__(64 * MSDA_HW, 1) void msda_fwd_raw_head_kernel(
    int S, int Q, const float* __restrict__ value, long long vpitch,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ raw, long long ldraw, int off_col0, int lgt_col0,
    const float* __restrict__ ref, float* __restrict__ out, int qpc) {
  constexpr int TL = 4, Dh = 32, H = 8, NP = TL * TP, PPL = NP / 8, NG = LS * TP;   // NG: samples served from memory
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  // exchange slots [wave][query of 8][sample]: the four (bilinear x attention) weights and ONE word - the offset of the
  // (low, low) corner, a multiple of four, with "the high column / row is a different token" in its two low bits
  float4* s_w = reinterpret_cast<float4*>(hsm);
  int* s_off = reinterpret_cast<int*>(hsm + MSDA_HW * 8 * NP * 16);
  float4* s_val = reinterpret_cast<float4*>(hsm + MSDA_HW * 8 * NP * 20);   // [S - t0][8] float4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int t0 = (int)lsi[LS], nst = S - t0;
  const int vh = (int)(vpitch / Dh);
  const float* vscene = value + (size_t)b * S * vpitch;
  for (int i = tid; i < nst * 8; i += 64 * MSDA_HW)
    s_val[i] = *reinterpret_cast<const float4*>(vscene + (size_t)(t0 + (i >> 3)) * vpitch + h * Dh + 4 * (i & 7));
  __syncthreads();
  const int grp = lane >> 3, sub = lane & 7;
  const float* vb = vscene + sub * 4;
  int* my_off = s_off + (wave * 8 + grp) * NP;
  float4* my_w = s_w + (wave * 8 + grp) * NP;
  const int qbeg = blockIdx.x * qpc, qend = min(Q, qbeg + qpc);
  // a pass's raw logits / offsets / reference points are requested one pass ahead (under the previous pass's gather)
  float lg[PPL], lgn[PPL];
  float2 of[PPL], ofn[PPL], rp[PPL], rpn[PPL];
  auto load_raw = [&](int q0, float (&g)[PPL], float2 (&o)[PPL], float2 (&r)[PPL]) {
    const int q = min(q0 + grp, qend - 1);                        // (a padded group repeats the last query, never stored)
    const long long row = (long long)b * Q + q;
    const float* rrow = raw + row * ldraw;
#pragma unroll
    for (int s2 = 0; s2 < PPL; ++s2) {
      const int i = sub + 8 * s2;
      g[s2] = __builtin_nontemporal_load(rrow + lgt_col0 + h * NP + i);
      o[s2].x = __builtin_nontemporal_load(rrow + off_col0 + 2 * (h * NP + i));
      o[s2].y = __builtin_nontemporal_load(rrow + off_col0 + 2 * (h * NP + i) + 1);
      r[s2] = *reinterpret_cast<const float2*>(ref + row * (TL * 2) + 2 * (i / TP));
    }
  };
  // the lane's (level, point) pairs are the same in every pass: their level constants once
  int cHl[PPL], cWl[PPL], cst[PPL];
  float fHl[PPL], fWl[PPL];
#pragma unroll
  for (int s2 = 0; s2 < PPL; ++s2) {
    const int l = (sub + 8 * s2) / TP;
    cHl[s2] = (int)shapes[2 * l]; cWl[s2] = (int)shapes[2 * l + 1];
    fHl[s2] = (float)cHl[s2]; fWl[s2] = (float)cWl[s2];
    cst[s2] = l < LS ? (int)lsi[l] : (int)lsi[l] - t0;
  }
  // row strides of the four levels, in the units of the packed offsets (floats in memory, float4 slots in LDS)
  int rowstride[TL];
#pragma unroll
  for (int l = 0; l < TL; ++l) rowstride[l] = (int)shapes[2 * l + 1] * (l < LS ? (int)vpitch : 8);
  if (qbeg + 8 * wave < qend) load_raw(qbeg + 8 * wave, lg, of, rp);
  for (int q0 = qbeg + 8 * wave; q0 < qend; q0 += 8 * MSDA_HW) {
    const int q = min(q0 + grp, qend - 1);
    const long long row = (long long)b * Q + q;
    float e[PPL];
    float mx = -3.0e38f;
#pragma unroll
    for (int s2 = 0; s2 < PPL; ++s2) mx = fmaxf(mx, lg[s2]);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float den = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < PPL; ++s2) { e[s2] = expf(lg[s2] - mx); den += e[s2]; }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) den += __shfl_xor(den, o);
#pragma unroll
    for (int s2 = 0; s2 < PPL; ++s2) {
      const int i = sub + 8 * s2, l = i / TP;
      const float aw = e[s2] / den;
      const int Hl = cHl[s2], Wl = cWl[s2];
      const float lx = rp[s2].x + of[s2].x / fWl[s2], ly = rp[s2].y + of[s2].y / fHl[s2];
      // fine levels: float offsets into the scene's value rows; staged levels: float4 index into the LDS slice
      const Corner c = l < LS ? make_corner(lx, ly, Hl, Wl, cst[s2], vh, Dh, h)
                              : make_corner(lx, ly, Hl, Wl, cst[s2], 1, 8, 0);
      my_off[i] = c.off[0] | (c.off[1] != c.off[0] ? 1 : 0) | (c.off[2] != c.off[0] ? 2 : 0);
      my_w[i] = make_float4(c.cw[0] * aw, c.cw[1] * aw, c.cw[2] * aw, c.cw[3] * aw);
    }
    if (q0 + 8 * MSDA_HW < qend) load_raw(q0 + 8 * MSDA_HW, lgn, ofn, rpn);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's own LDS writes (in order, one wave)
    __builtin_amdgcn_wave_barrier();
    // two batches: the fine-level corners of half the samples are requested, half the LDS samples are summed under them
    constexpr int NB = NG / 2, NLH = (NP - NG) / 2;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int bt = 0; bt < 2; ++bt) {
      // (LS == 2: a batch is exactly one level)
      const int rs_g = rowstride[bt & (TL - 1)], rs_l = rowstride[(LS + bt) & (TL - 1)];
      float4 gv[NB][4];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = bt * NB + j;
        const int o = my_off[i];
        const int sx = (o & 1) ? (int)vpitch : 0, sy = (o & 2) ? (NB == TP ? rs_g : rowstride[i / TP]) : 0;
        const float* v0 = vb + (o & ~3);
        gv[j][0] = ld4<float>(v0); gv[j][1] = ld4<float>(v0 + sx);
        gv[j][2] = ld4<float>(v0 + sy); gv[j][3] = ld4<float>(v0 + sx + sy);
      }
      __builtin_amdgcn_sched_barrier(0);          // all of the batch requested before the LDS samples are summed
#pragma unroll
      for (int j = 0; j < NLH; ++j) {
        const int i = NG + bt * NLH + j;
        const int o = my_off[i];
        const int sx = (o & 1) ? 8 : 0, sy = (o & 2) ? (NLH == TP ? rs_l : rowstride[i / TP]) : 0;
        const float4 w = my_w[i];
        const float4* v0 = s_val + (o & ~3) + sub;
        acc = f4_fma(w.x, v0[0], acc);
        acc = f4_fma(w.y, v0[sx], acc);
        acc = f4_fma(w.z, v0[sy], acc);
        acc = f4_fma(w.w, v0[sx + sy], acc);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float4 w = my_w[bt * NB + j];
        acc = f4_fma(w.x, gv[j][0], acc);
        acc = f4_fma(w.y, gv[j][1], acc);
        acc = f4_fma(w.z, gv[j][2], acc);
        acc = f4_fma(w.w, gv[j][3], acc);
      }
    }
    if (q0 + grp < qend) {
      float* o = out + row * (H * Dh) + h * Dh + sub * 4;
      __builtin_nontemporal_store(acc.x, o); __builtin_nontemporal_store(acc.y, o + 1);
      __builtin_nontemporal_store(acc.z, o + 2); __builtin_nontemporal_store(acc.w, o + 3);
    }
    __builtin_amdgcn_wave_barrier();                              // the exchange slots are rewritten by the next pass
#pragma unroll
    for (int s2 = 0; s2 < PPL; ++s2) { lg[s2] = lgn[s2]; of[s2] = ofn[s2]; rp[s2] = rpn[s2]; }
  }
}

template <int G, int TL, int TP, typename VT = float>
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    int S, int H, int Dh, int L, int Q, int P, const VT* __restrict__ value,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attw,
    const float* __restrict__ gout, float* __restrict__ gvalue, float* __restrict__ gloc,
    float* __restrict__ gattw, long long items) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long item = t / G;
  const bool live = item < items;
  if (!live) item = items - 1;  // keep the lane in the DPP reductions
  const int sub = (int)(t - (t / G) * G);
  const int h = (int)(item % H);
  const int b = (int)(item / ((long long)Q * H));
  const size_t boff = (size_t)b * S * H * Dh + sub * 4;
  const VT* vb = value + boff;
  float* gvb = gvalue + boff;
  if constexpr (TL > 0) {
    L = TL;
    P = TP;
  }
  const int nlp = L * P;
  const float* lp = loc + item * nlp * 2;
  const float* wp = attw + item * nlp;
  const float4 top = *reinterpret_cast<const float4*>(gout + item * Dh + sub * 4);
#pragma unroll
  for (int i = 0; i < nlp; ++i) {
    const int l = i / P;
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const int start = (int)lsi[l];
    const Corner c = make_corner(lp[2 * i], lp[2 * i + 1], Hl, Wl, start, H, Dh, h);
    const float aw = wp[i];
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = ld4<VT>(vb + c.off[k]);
      v[k].x *= c.ok[k]; v[k].y *= c.ok[k]; v[k].z *= c.ok[k]; v[k].w *= c.ok[k];
    }
    // d(sample)/d(h), d(sample)/d(w), and the sample itself, per channel
    float4 gh, gw, val;
#define MSDA_CH(m)                                                                     \
    gh.m = c.hw * (v[2].m - v[0].m) + c.lw * (v[3].m - v[1].m);                        \
    gw.m = c.hh * (v[1].m - v[0].m) + c.lh * (v[3].m - v[2].m);                        \
    val.m = c.hh * c.hw * v[0].m + c.hh * c.lw * v[1].m + c.lh * c.hw * v[2].m +       \
            c.lh * c.lw * v[3].m;
    MSDA_CH(x) MSDA_CH(y) MSDA_CH(z) MSDA_CH(w)
#undef MSDA_CH
    float g_w = f4_dot(top, val);
    float g_x = f4_dot(top, gw) * aw * (float)Wl;
    float g_y = f4_dot(top, gh) * aw * (float)Hl;
    g_w = group_allsum<G>(g_w);
    g_x = group_allsum<G>(g_x);
    g_y = group_allsum<G>(g_y);
    if (live) {
      if (sub == 0) {
        gattw[item * nlp + i] = g_w;
        gloc[(item * nlp + i) * 2 + 0] = g_x;
        gloc[(item * nlp + i) * 2 + 1] = g_y;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (gvalue != nullptr && c.ok[k] != 0.f) {
          const float s = c.cw[k] * aw;
          float* g = gvb + c.off[k];
          atomicAdd(g + 0, s * top.x);
          atomicAdd(g + 1, s * top.y);
          atomicAdd(g + 2, s * top.z);
          atomicAdd(g + 3, s * top.w);
        }
      }
    }
  }
}

// Any Dh (multiple of nothing): one thread per (b,q,h), serial over channels.
template <typename VT = float>
__global__ __launch_bounds__(256) void msda_fwd_generic(
    int S, int H, int Dh, int L, int Q, int P, const VT* __restrict__ value,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attw, float* __restrict__ out,
    long long items) {
  const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= items) return;
  const int h = (int)(item % H);
  const int b = (int)(item / ((long long)Q * H));
  const VT* vb = value + (size_t)b * S * H * Dh;
  const int nlp = L * P;
  for (int ch = 0; ch < Dh; ++ch) {
    float acc = 0.f;
    for (int i = 0; i < nlp; ++i) {
      const int l = i / P;
      const Corner c = make_corner(loc[(item * nlp + i) * 2], loc[(item * nlp + i) * 2 + 1],
                                   (int)shapes[2 * l], (int)shapes[2 * l + 1], (int)lsi[l], H,
                                   Dh, h);
      float v = 0.f;
      for (int k = 0; k < 4; ++k) v = __builtin_fmaf(c.cw[k], ld1<VT>(vb + c.off[k] + ch), v);
      acc = __builtin_fmaf(attw[item * nlp + i], v, acc);
    }
    out[item * Dh + ch] = acc;
  }
}

template <typename VT = float>
__global__ __launch_bounds__(256) void msda_bwd_generic(
    int S, int H, int Dh, int L, int Q, int P, const VT* __restrict__ value,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attw,
    const float* __restrict__ gout, float* __restrict__ gvalue, float* __restrict__ gloc,
    float* __restrict__ gattw, long long items) {
  const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= items) return;
  const int h = (int)(item % H);
  const int b = (int)(item / ((long long)Q * H));
  const size_t boff = (size_t)b * S * H * Dh;
  const int nlp = L * P;
  for (int i = 0; i < nlp; ++i) {
    const int l = i / P;
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const Corner c = make_corner(loc[(item * nlp + i) * 2], loc[(item * nlp + i) * 2 + 1], Hl,
                                 Wl, (int)lsi[l], H, Dh, h);
    const float aw = attw[item * nlp + i];
    float g_w = 0.f, g_x = 0.f, g_y = 0.f;
    for (int ch = 0; ch < Dh; ++ch) {
      const float top = gout[item * Dh + ch];
      float v[4];
      for (int k = 0; k < 4; ++k) v[k] = ld1<VT>(value + boff + c.off[k] + ch) * c.ok[k];
      const float gh = c.hw * (v[2] - v[0]) + c.lw * (v[3] - v[1]);
      const float gw = c.hh * (v[1] - v[0]) + c.lh * (v[3] - v[2]);
      const float val = c.hh * c.hw * v[0] + c.hh * c.lw * v[1] + c.lh * c.hw * v[2] +
                        c.lh * c.lw * v[3];
      g_w += top * val;
      g_x += top * gw;
      g_y += top * gh;
      for (int k = 0; k < 4; ++k)
        if (gvalue != nullptr && c.ok[k] != 0.f)
          atomicAdd(gvalue + boff + c.off[k] + ch, c.cw[k] * aw * top);
    }
    gattw[item * nlp + i] = g_w;
    gloc[(item * nlp + i) * 2 + 0] = g_x * aw * (float)Wl;
    gloc[(item * nlp + i) * 2 + 1] = g_y * aw * (float)Hl;
  }
}

template <int G, typename VT>
static void launch_fwd(dim3 grid, hipStream_t s, int S, int H, int Dh, int L, int Q,
                       int P, const VT* value, const int64_t* shapes, const int64_t* lsi,
                       const float* loc, const float* w, float* out, long long items) {
  if (L == 4 && P == 2)
    hipLaunchKernelGGL((msda_fwd_kernel<G, 4, 2, VT>), grid, dim3(256), 0, s, S, H, Dh, L, Q, P, value,
                       shapes, lsi, loc, w, out, items);
  else if (L == 4 && P == 4)
    hipLaunchKernelGGL((msda_fwd_kernel<G, 4, 4, VT>), grid, dim3(256), 0, s, S, H, Dh, L, Q, P,
                       value, shapes, lsi, loc, w, out, items);
  else
    hipLaunchKernelGGL((msda_fwd_kernel<G, 0, 0, VT>), grid, dim3(256), 0, s, S, H, Dh, L, Q, P, value,
                       shapes, lsi, loc, w, out, items);
}

template <int G, typename VT>
static void launch_bwd(dim3 grid, hipStream_t s, int S, int H, int Dh, int L, int Q,
                       int P, const VT* value, const int64_t* shapes, const int64_t* lsi,
                       const float* loc, const float* w, const float* gout, float* gvalue,
                       float* gloc, float* gw, long long items) {
  if (L == 4 && P == 2)
    hipLaunchKernelGGL((msda_bwd_kernel<G, 4, 2, VT>), grid, dim3(256), 0, s, S, H, Dh, L, Q, P, value,
                       shapes, lsi, loc, w, gout, gvalue, gloc, gw, items);
  else if (L == 4 && P == 4)
    hipLaunchKernelGGL((msda_bwd_kernel<G, 4, 4, VT>), grid, dim3(256), 0, s, S, H, Dh, L, Q, P,
                       value, shapes, lsi, loc, w, gout, gvalue, gloc, gw, items);
  else
    hipLaunchKernelGGL((msda_bwd_kernel<G, 0, 0, VT>), grid, dim3(256), 0, s, S, H, Dh, L, Q, P, value,
                       shapes, lsi, loc, w, gout, gvalue, gloc, gw, items);
}

static int msda_check(int B, int S, int H, int Dh, int L, int Q, int P) {
  DEMF_REQUIRE(B >= 0 && S >= 1 && H >= 1 && Dh >= 1 && L >= 1 && L <= MAX_L && Q >= 0 && P >= 1,
               "msda: bad sizes B=%d S=%d H=%d Dh=%d L=%d Q=%d P=%d", B, S, H, Dh, L, Q, P);
  DEMF_REQUIRE((long long)S * H * Dh < (1LL << 31), "msda: one scene of value exceeds 2^31 elements");
  return DEMF_OK;
}

}  // namespace demf

using namespace demf;

template <typename VT>
static int msda_fwd_impl(int B, int S, int H, int Dh, int L, int Q, int P,
                         const VT* value, const int64_t* spatial_shapes,
                         const int64_t* level_start_index, const float* sampling_loc,
                         const float* attn_weight, float* out, demf_stream_t stream) {
  if (int e = msda_check(B, S, H, Dh, L, Q, P)) return e;
  if (B == 0 || Q == 0) return DEMF_OK;
  DEMF_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out,
               "msda_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const long long items = (long long)B * Q * H;
  const int G = Dh / 4;
  const bool fast = (Dh % 4 == 0) && (G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 ||
                                      G == 64) &&
                    (((uintptr_t)value % (4 * sizeof(VT))) == 0 && (uintptr_t)out % 16 == 0);
  if (!fast) {
    hipLaunchKernelGGL(msda_fwd_generic<VT>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s,
                       S, H, Dh, L, Q, P, value, spatial_shapes, level_start_index, sampling_loc,
                       attn_weight, out, items);
    return check_launch("msda_fwd_generic");
  }
  const dim3 grid((unsigned)((items * G + 255) / 256));
#define GO(g)                                                                              \
  launch_fwd<g, VT>(grid, s, S, H, Dh, L, Q, P, value, spatial_shapes, level_start_index, \
                sampling_loc, attn_weight, out, items)
  switch (G) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    case 32: GO(32); break;
    default: GO(64); break;
  }
#undef GO
  return check_launch("msda_fwd");
}

template <typename VT>
static int msda_bwd_impl(int B, int S, int H, int Dh, int L, int Q, int P,
                         const VT* value, const int64_t* spatial_shapes,
                         const int64_t* level_start_index, const float* sampling_loc,
                         const float* attn_weight, const float* grad_out,
                         float* grad_value, float* grad_sampling_loc,
                         float* grad_attn_weight, demf_stream_t stream) {
  if (int e = msda_check(B, S, H, Dh, L, Q, P)) return e;
  if (B == 0 || Q == 0) return DEMF_OK;
  DEMF_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight &&
                   grad_out && grad_sampling_loc && grad_attn_weight,
               "msda_bwd: null pointer");   // grad_value may be NULL: no gradient wrt value wanted
  hipStream_t s = (hipStream_t)stream;
  const long long items = (long long)B * Q * H;
  const int G = Dh / 4;
  const bool fast = (Dh % 4 == 0) && (G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 ||
                                      G == 64) &&
                    (((uintptr_t)value % (4 * sizeof(VT))) == 0 && (uintptr_t)grad_out % 16 == 0);
  if (!fast) {
    hipLaunchKernelGGL(msda_bwd_generic<VT>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s,
                       S, H, Dh, L, Q, P, value, spatial_shapes, level_start_index, sampling_loc,
                       attn_weight, grad_out, grad_value, grad_sampling_loc, grad_attn_weight,
                       items);
    return check_launch("msda_bwd_generic");
  }
  const dim3 grid((unsigned)((items * G + 255) / 256));
#define GO(g)                                                                              \
  launch_bwd<g, VT>(grid, s, S, H, Dh, L, Q, P, value, spatial_shapes, level_start_index, \
                sampling_loc, attn_weight, grad_out, grad_value, grad_sampling_loc,        \
                grad_attn_weight, items)
  switch (G) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    case 32: GO(32); break;
    default: GO(64); break;
  }
#undef GO
  return check_launch("msda_bwd");
}

extern "C" int demf_msda_fwd_f32(int B, int S, int H, int Dh, int L, int Q, int P, const float* value,
                                 const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const float* sampling_loc, const float* attn_weight, float* out,
                                 demf_stream_t stream) {
  return msda_fwd_impl<float>(B, S, H, Dh, L, Q, P, value, spatial_shapes, level_start_index, sampling_loc,
                              attn_weight, out, stream);
}

extern "C" int demf_msda_bwd_f32(int B, int S, int H, int Dh, int L, int Q, int P, const float* value,
                                 const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const float* sampling_loc, const float* attn_weight, const float* grad_out,
                                 float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                                 demf_stream_t stream) {
  return msda_bwd_impl<float>(B, S, H, Dh, L, Q, P, value, spatial_shapes, level_start_index, sampling_loc,
                              attn_weight, grad_out, grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

extern "C" int demf_msda_fwd_bf16(int B, int S, int H, int Dh, int L, int Q, int P, const uint16_t* value,
                                  const int64_t* spatial_shapes, const int64_t* level_start_index,
                                  const float* sampling_loc, const float* attn_weight, float* out,
                                  demf_stream_t stream) {
  return msda_fwd_impl<uint16_t>(B, S, H, Dh, L, Q, P, value, spatial_shapes, level_start_index, sampling_loc,
                                 attn_weight, out, stream);
}

extern "C" int demf_msda_bwd_bf16(int B, int S, int H, int Dh, int L, int Q, int P, const uint16_t* value,
                                  const int64_t* spatial_shapes, const int64_t* level_start_index,
                                  const float* sampling_loc, const float* attn_weight, const float* grad_out,
                                  float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                                  demf_stream_t stream) {
  return msda_bwd_impl<uint16_t>(B, S, H, Dh, L, Q, P, value, spatial_shapes, level_start_index, sampling_loc,
                                 attn_weight, grad_out, grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

// demf_msda_fwd_raw_f32 for H = 8, Dh = 32, L = 4 with the value rows of levels first_staged_level .. 3 (staged_tokens =
// S - level_start_index[first_staged_level] of them - the CALLER knows the pyramid's shapes on the host, the library
// never reads a device array back) resident in LDS per (scene, head): msda_fwd_raw_head_kernel.  Same results up to the
// summation order of a query's samples.
extern "C" int demf_msda_fwd_raw_f32(int B, int S, int H, int Dh, int L, int Q, int P, const float* value,
                                     long long vpitch, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                     const float* raw, long long ldraw, int off_col0, int lgt_col0, const float* ref,
                                     float* out, demf_stream_t stream);

extern "C" int demf_msda_fwd_raw_head_f32(int B, int S, int Q, int P, const float* value, long long vpitch,
                                          const int64_t* spatial_shapes, const int64_t* level_start_index,
                                          const float* raw, long long ldraw, int off_col0, int lgt_col0,
                                          const float* ref, float* out, int first_staged_level, int staged_tokens,
                                          demf_stream_t stream) {
  DEMF_REQUIRE(B >= 0 && S >= 1 && Q >= 0 && (P == 2 || P == 4), "msda_fwd_raw_head: bad sizes B=%d S=%d Q=%d P=%d", B, S, Q, P);
  if (B == 0 || Q == 0) return DEMF_OK;
  DEMF_REQUIRE(value && spatial_shapes && level_start_index && raw && ref && out, "msda_fwd_raw_head: null pointer");
  DEMF_REQUIRE(vpitch % 32 == 0 && vpitch >= 256 && (long long)S * vpitch < (1LL << 31) && (uintptr_t)value % 16 == 0 &&
                   (uintptr_t)out % 16 == 0 && ldraw % 2 == 0 && off_col0 % 2 == 0 && (uintptr_t)raw % 8 == 0 &&
                   (uintptr_t)ref % 8 == 0, "msda_fwd_raw_head: alignment / pitch");
  const int NP = 4 * P;
  const size_t lds = (size_t)MSDA_HW * 8 * NP * 20 + (size_t)staged_tokens * 128;
  DEMF_REQUIRE((first_staged_level == 2 || first_staged_level == 3) && staged_tokens >= 1 && staged_tokens < S &&
                   lds <= 160 * 1024, "msda_fwd_raw_head: levels %d .. 3 = %d tokens do not fit LDS", first_staged_level,
               staged_tokens);
  hipStream_t s = (hipStream_t)stream;
  static const int target_wgs = getenv("DEMF_MSDA_HEAD_WGS") ? atoi(getenv("DEMF_MSDA_HEAD_WGS")) : 256;   // A/B knob
  int chunks = (target_wgs + B * 8 - 1) / (B * 8);
  if (chunks < 1) chunks = 1;
  int qpc = ((Q + chunks - 1) / chunks + 8 * MSDA_HW - 1) / (8 * MSDA_HW) * (8 * MSDA_HW);
  chunks = (Q + qpc - 1) / qpc;
  const dim3 grid(chunks, 8, B);
#define HEAD_GO(PV, LSV)                                                                                             \
  {                                                                                                                  \
    static unsigned long long reserved = 0;      /* bit per device */                                               \
    if (!reserve_lds(reinterpret_cast<const void*>(&msda_fwd_raw_head_kernel<PV, LSV>), 160 * 1024, &reserved))      \
      /* no 160 KB of LDS on this device: the wave-per-query form serves the same call */                            \
      return demf_msda_fwd_raw_f32(B, S, 8, 32, 4, Q, P, value, vpitch, spatial_shapes, level_start_index, raw,      \
                                   ldraw, off_col0, lgt_col0, ref, out, stream);                                      \
    hipLaunchKernelGGL((msda_fwd_raw_head_kernel<PV, LSV>), grid, dim3(64 * MSDA_HW), lds, s, S, Q, value, vpitch, spatial_shapes, \
                       level_start_index, raw, ldraw, off_col0, lgt_col0, ref, out, qpc);                            \
  }
  if (P == 4) { if (first_staged_level == 2) HEAD_GO(4, 2) else HEAD_GO(4, 3) }
  else { if (first_staged_level == 2) HEAD_GO(2, 2) else HEAD_GO(2, 3) }
#undef HEAD_GO
  return check_launch("msda_fwd_raw_head");
}

extern "C" int demf_msda_fwd_raw_f32(int B, int S, int H, int Dh, int L, int Q, int P, const float* value,
                                     long long vpitch, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                     const float* raw, long long ldraw, int off_col0, int lgt_col0, const float* ref,
                                     float* out, demf_stream_t stream) {
  if (int e = msda_check(B, S, H, Dh, L, Q, P)) return e;
  if (B == 0 || Q == 0) return DEMF_OK;
  DEMF_REQUIRE(value && spatial_shapes && level_start_index && raw && ref && out, "msda_fwd_raw: null pointer");
  DEMF_REQUIRE(Dh == 32 && L == 4 && (P == 4 || P == 2), "msda_fwd_raw: built for Dh = 32, L = 4, P in {2, 4}");
  DEMF_REQUIRE(vpitch % Dh == 0 && vpitch >= (long long)H * Dh && (long long)S * vpitch < (1LL << 31) &&
                   (uintptr_t)value % 16 == 0 && (uintptr_t)out % 16 == 0 && vpitch % 4 == 0,
               "msda_fwd_raw: value pitch %lld", vpitch);
  const long long items = (long long)B * Q * H;
  const dim3 grid((unsigned)((items * 8 + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  static const bool lanes8 = getenv("DEMF_MSDA_RAW_LANES") && atoi(getenv("DEMF_MSDA_RAW_LANES"));   // A/B switch
  if (H == 8 && !lanes8 && ldraw % 2 == 0 && off_col0 % 2 == 0 && (uintptr_t)raw % 8 == 0 && (uintptr_t)ref % 8 == 0) {
    const long long rows = (long long)B * Q;
    static const int xcd = getenv("DEMF_MSDA_XCD") ? atoi(getenv("DEMF_MSDA_XCD")) : 0;      // A/B switch
    const dim3 g2((unsigned)(xcd ? ((rows + 3) / 4 + 7) / 8 * 8 : (rows + 3) / 4));
    static const int nt = getenv("DEMF_MSDA_NT") ? atoi(getenv("DEMF_MSDA_NT")) : 1;         // A/B switch (12.94 -> 12.65 ms per 6 layers)
#define RAW_GO(PV, NTV)                                                                                              \
    hipLaunchKernelGGL((msda_fwd_raw_wave_kernel<4, PV, NTV>), g2, dim3(256), 0, s, S, Q, value, vpitch, spatial_shapes, \
                       level_start_index, raw, ldraw, off_col0, lgt_col0, ref, out, rows, xcd)
    if (P == 4) { if (nt) RAW_GO(4, true); else RAW_GO(4, false); }
    else { if (nt) RAW_GO(2, true); else RAW_GO(2, false); }
#undef RAW_GO
    return check_launch("msda_fwd_raw_wave");
  }
  if (P == 4)
    hipLaunchKernelGGL((msda_fwd_raw_kernel<8, 4, 4>), grid, dim3(256), 0, s, S, H, Dh, Q, value, vpitch, spatial_shapes,
                       level_start_index, raw, ldraw, off_col0, lgt_col0, ref, out, items);
  else
    hipLaunchKernelGGL((msda_fwd_raw_kernel<8, 4, 2>), grid, dim3(256), 0, s, S, H, Dh, Q, value, vpitch, spatial_shapes,
                       level_start_index, raw, ldraw, off_col0, lgt_col0, ref, out, items);
  return check_launch("msda_fwd_raw");
}
