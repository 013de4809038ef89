// Dense building blocks of the DeMF fusion decoder layer on gfx950: one strided / batched /
// split-K fp32-MFMA GEMM with fused prologue + epilogue, and the row kernels that sit between the
// GEMMs (dropout + residual + LayerNorm, attention softmax, sampling-location preparation).
//
// Reference: demf/modeling/layers/transformer.py:55-80 delegates to an mmcv
// DetrTransformerDecoderLayer (configs/demf/demf_votenet.py:71-91): nn.MultiheadAttention self
// attention, MultiScaleDeformableAttention cross attention into the image pyramid, FFN, three
// LayerNorms, dropout 0.4 / 0.1.  Upstream runs it as ~190 library launches forward and as many
// backward for 2 048 query rows (8 scenes x 256 queries): at that size every launch is latency,
// not work.  Here the layer is ~17 launches forward and ~30 backward, all of them this file's
// kernels + the MSDA kernels of msda.hip.
//
// GEMM: C[m,n] (+)= epi( alpha * sum_k (A[m,k] (+ A2[m,k])) * B[n,k] ), operands addressed by
// element strides so that  Y = X.W^T (both K-contiguous),  dX = dY.W (B given K-strided) and
// dW = dY^T.X (both operands M-contiguous along the reduction) are the same kernel; 64x64 block
// tiles, 4 waves x one 32x32 MFMA accumulator, K staged in steps of 32 through LDS (fragment trick
// of mlp.hip: a lane feeds component m of its float4 along K to MFMA m).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "rng.h"

namespace demf {

using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));

// flags / descriptor: include/demf_hip.h (demf_gemm_desc)
enum {
  GEMM_RELU = DEMF_GEMM_RELU, GEMM_DROPOUT = DEMF_GEMM_DROPOUT, GEMM_GATE = DEMF_GEMM_GATE,
  GEMM_ACCUM = DEMF_GEMM_ACCUM, GEMM_ROWBIAS = DEMF_GEMM_ROWBIAS, GEMM_ACCUM2 = DEMF_GEMM_ACCUM2,
  GEMM_FP32 = DEMF_GEMM_FP32,
};
using GemmArgs = demf_gemm_desc;

constexpr int G_BM = 64, G_BN = 64, G_BK = 64, G_LD = G_BK + 4;

// One 64 x 64 operand tile (rows r0.., reduction k0..; element (r,k) at P + r*sr + k*sk) travels
// global -> registers (gemm_fetch, issued one K step ahead so that the loads are in flight while the
// MFMAs of the current step run) -> LDS (gemm_commit, + the optional second addend).
struct GemmRegs {
  float4 v[4], w[4];            // modes 1 / 2: four float4 per thread (+ second operand)
};

template <int mode, bool ADD2>
__device__ __forceinline__ void gemm_fetch(GemmRegs& g, float (&sc)[16], const float* __restrict__ P,
                                           const float* __restrict__ P2, long long sr, long long sk,
                                           int r0, int R, int k0, int K1) {
  const int t = threadIdx.x;
  if constexpr (mode == 1) {                       // k-contiguous, 16-byte aligned rows: float4 along K
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (t >> 4) + 16 * i, kq = (t & 15) * 4;
      g.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (ADD2) g.w[i] = g.v[i];
      if (r0 + row < R && k0 + kq < K1) {    // K1 % 4 == 0 in this mode
        const size_t o = (size_t)(r0 + row) * sr + k0 + kq;
        g.v[i] = *reinterpret_cast<const float4*>(P + o);
        if constexpr (ADD2) g.w[i] = *reinterpret_cast<const float4*>(P2 + o);
      }
    }
  } else if constexpr (mode == 2) {      // row-contiguous (reduction strided): float4 along rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (t >> 4) + 16 * i, rq = (t & 15) * 4;
      g.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (ADD2) g.w[i] = g.v[i];
      if (k0 + k < K1 && r0 + rq < R) {      // R % 4 == 0 in this mode
        const size_t o = (size_t)(k0 + k) * sk + r0 + rq;
        g.v[i] = *reinterpret_cast<const float4*>(P + o);
        if constexpr (ADD2) g.w[i] = *reinterpret_cast<const float4*>(P2 + o);
      }
    }
  } else {                               // anything else: guarded scalar loads
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = t + 256 * i, row = e >> 6, k = e & 63;
      float v = 0.f;
      if (r0 + row < R && k0 + k < K1) {
        const size_t o = (size_t)(r0 + row) * sr + (size_t)(k0 + k) * sk;
        v = P[o];
        if constexpr (ADD2) v += P2[o];
      }
      sc[i] = v;
    }
  }
}

constexpr int G_LDB = 2 * (G_BK + 8);      // bf16 tiles: bytes per row

__device__ __forceinline__ float4 add4(float4 v, const float4& w) {
  v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  return v;
}

// bf16 flavour of gemm_commit: element (row, k) at byte row * G_LDB + 2k
template <int mode, bool ADD2>
__device__ __forceinline__ void gemm_commit_bf16(char* __restrict__ s, const GemmRegs& g, const float (&sc)[16]) {
  const int t = threadIdx.x;
  if constexpr (mode == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = ADD2 ? add4(g.v[i], g.w[i]) : g.v[i];
      bf16x4 b;
      b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(s + ((t >> 4) + 16 * i) * G_LDB + 2 * (t & 15) * 4) = b;
    }
  } else if constexpr (mode == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = ADD2 ? add4(g.v[i], g.w[i]) : g.v[i];
      char* d = s + (t & 15) * 4 * G_LDB + 2 * ((t >> 4) + 16 * i);
      *reinterpret_cast<__bf16*>(d) = (__bf16)v.x;
      *reinterpret_cast<__bf16*>(d + G_LDB) = (__bf16)v.y;
      *reinterpret_cast<__bf16*>(d + 2 * G_LDB) = (__bf16)v.z;
      *reinterpret_cast<__bf16*>(d + 3 * G_LDB) = (__bf16)v.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = t + 256 * i;
      *reinterpret_cast<__bf16*>(s + (e >> 6) * G_LDB + 2 * (e & 63)) = (__bf16)sc[i];
    }
  }
}

template <int mode, bool ADD2>
__device__ __forceinline__ void gemm_commit(float* __restrict__ s, const GemmRegs& g, const float (&sc)[16]) {
  const int t = threadIdx.x;
  if constexpr (mode == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = g.v[i];
      if constexpr (ADD2) { v.x += g.w[i].x; v.y += g.w[i].y; v.z += g.w[i].z; v.w += g.w[i].w; }
      *reinterpret_cast<float4*>(s + ((t >> 4) + 16 * i) * G_LD + (t & 15) * 4) = v;
    }
  } else if constexpr (mode == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = g.v[i];
      if constexpr (ADD2) { v.x += g.w[i].x; v.y += g.w[i].y; v.z += g.w[i].z; v.w += g.w[i].w; }
      float* d = s + (t & 15) * 4 * G_LD + (t >> 4) + 16 * i;
      d[0] = v.x; d[G_LD] = v.y; d[2 * G_LD] = v.z; d[3 * G_LD] = v.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = t + 256 * i;
      s[(e >> 6) * G_LD + (e & 63)] = sc[i];
    }
  }
}

// one output element through the fused epilogue
__device__ __forceinline__ void gemm_store(const GemmArgs& p, int z, size_t oc, int sp, int m, int n,
                                           float bias, float accv) {
  float v = p.alpha * accv;
  if (p.flags & GEMM_ROWBIAS) {
    if (sp == 0) v = __builtin_fmaf(bias, p.rowscale[(size_t)m * p.srs_m + (size_t)z * p.srs_b], v);
  } else {
    v += bias;
  }
  if (p.flags & GEMM_RELU) v = fmaxf(v, 0.f);
  if (p.flags & GEMM_DROPOUT) {
    const unsigned long long idx = ((unsigned long long)z * p.M + m) * p.N + n;
    v = dropout_keep((const unsigned long long*)p.rng, (unsigned)p.op_id, idx, p.drop_p)
            ? v * (1.0f / (1.0f - p.drop_p)) : 0.f;
  }
  if (p.flags & GEMM_GATE)
    v = p.gate[(size_t)z * p.sgb + (size_t)m * p.sgm + n] != 0.f ? v * p.gate_scale : 0.f;
  float* dst = p.C + oc + (size_t)m * p.scm + n;
  if (p.splitk > 1) atomicAdd(dst, v);
  else if (p.flags & GEMM_ACCUM) *dst += v;
  else *dst = v;
  if (p.C2 != nullptr) {                          // same values to a second consumer
    float* d2 = p.C2 + oc + (size_t)m * p.scm + n;
    if (p.splitk > 1) atomicAdd(d2, v);
    else if (p.flags & GEMM_ACCUM2) *d2 += v;
    else *d2 = v;
  }
}

// MA / MB: staging mode of the A / B operand (stage_mode); ADD = a second addend on either operand
// Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every XCD has its own L2.  With
// (x = column tile, y = row tile) and a column count that is a multiple of 8 tiles, XCD i would own COLUMN
// tiles i, i + 8, ... of every row block: each of the 8 L2s pulls the whole A operand (measured: 25-28 MB
// fetched per 2 048 x 256 x 256 launch for 4.5 MB of operands).  Re-deal: linear id L -> row block
// (L mod 8) + 8 * (L / (8 gx)), column tile (L / 8) mod gx, so that all column tiles of a row block run on ONE
// XCD back to back (A rows fetched once in total, the small B operand once per XCD).
__device__ __forceinline__ void xcd_rows(int& bx, int& by, int gx, int gy) {
  if ((gy & 7) == 0) {
    const int L = bx + gx * by;
    by = (L & 7) + 8 * (L / (8 * gx));
    bx = (L >> 3) % gx;
  }
}

template <int MA, int MB, bool ADD, bool BF16>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float s_a[G_BM * G_LD];
  __shared__ __attribute__((aligned(16))) float s_b[G_BN * G_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  int bx = blockIdx.x, by = blockIdx.y;
  xcd_rows(bx, by, gridDim.x, gridDim.y);
  const int m0 = by * G_BM, n0 = bx * G_BN;
  const int z = blockIdx.z / p.splitk, sp = blockIdx.z - z * p.splitk;
  // split-K: contiguous K ranges, multiples of the K step
  const int kchunk = ((p.K + p.splitk - 1) / p.splitk + G_BK - 1) / G_BK * G_BK;
  const int kbeg = sp * kchunk, kend = min(p.K, kbeg + kchunk);
  // two-level batch index: z = zo * zdiv + zi, every operand has an (outer, inner) stride pair
  const int zo = z / p.zdiv, zi = z - zo * p.zdiv;
  const size_t oa = (size_t)zo * p.sab + (size_t)zi * p.sab2, ob = (size_t)zo * p.sbb + (size_t)zi * p.sbb2;
  const float* A = p.A + oa;
  const float* A2 = (p.A2 != nullptr && n0 < p.a2_cols) ? p.A2 + oa : nullptr;
  const float* B = p.B + ob;
  const float* B2 = (p.B2 != nullptr && m0 < p.b2_rows) ? p.B2 + ob : nullptr;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // Register prefetch ring, NS K-steps deep: with a few hundred rows there is one workgroup per CU
  // and nothing else to hide the global-load latency (~2 us) behind, so the loads of the next NS
  // steps are all in flight while a step's tiles go through LDS and the MFMAs (K = 256: the whole
  // reduction is requested up front; measured 19 -> 11 us per 2048 x 256 x 256 launch).
  constexpr int NS = ADD ? 2 : 3;
  GemmRegs ga[NS], gb[NS];
  float sa16[NS][16], sb16[NS][16];
  auto fetch = [&](auto stage, int k0) {
    constexpr int S = decltype(stage)::value;
    if (ADD && A2 != nullptr) gemm_fetch<MA, ADD>(ga[S], sa16[S], A, A2, p.sam, p.sak, m0, p.M, k0, kend);
    else gemm_fetch<MA, false>(ga[S], sa16[S], A, nullptr, p.sam, p.sak, m0, p.M, k0, kend);
    if (ADD && B2 != nullptr) gemm_fetch<MB, ADD>(gb[S], sb16[S], B, B2, p.sbn, p.sbk, n0, p.N, k0, kend);
    else gemm_fetch<MB, false>(gb[S], sb16[S], B, nullptr, p.sbn, p.sbk, n0, p.N, k0, kend);
  };
  auto step = [&](auto stage, int k0) {
    constexpr int S = decltype(stage)::value;
    lds_barrier();                                    // everyone is done reading the previous tiles
    if constexpr (BF16) {
      char* ca = reinterpret_cast<char*>(s_a);
      char* cb = reinterpret_cast<char*>(s_b);
      if (ADD && A2 != nullptr) gemm_commit_bf16<MA, ADD>(ca, ga[S], sa16[S]); else gemm_commit_bf16<MA, false>(ca, ga[S], sa16[S]);
      if (ADD && B2 != nullptr) gemm_commit_bf16<MB, ADD>(cb, gb[S], sb16[S]); else gemm_commit_bf16<MB, false>(cb, gb[S], sb16[S]);
      lds_barrier();
      if (k0 + NS * G_BK < kend) fetch(stage, k0 + NS * G_BK);
#pragma unroll
      for (int c16 = 0; c16 < G_BK / 16; ++c16) {
        const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(ca + (wm * 32 + lr) * G_LDB + 2 * (c16 * 16 + 8 * lh));
        const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(cb + (wn * 32 + lr) * G_LDB + 2 * (c16 * 16 + 8 * lh));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
      }
    } else {
      if (ADD && A2 != nullptr) gemm_commit<MA, ADD>(s_a, ga[S], sa16[S]); else gemm_commit<MA, false>(s_a, ga[S], sa16[S]);
      if (ADD && B2 != nullptr) gemm_commit<MB, ADD>(s_b, gb[S], sb16[S]); else gemm_commit<MB, false>(s_b, gb[S], sb16[S]);
      lds_barrier();
      if (k0 + NS * G_BK < kend) fetch(stage, k0 + NS * G_BK);   // in flight during the next NS steps
#pragma unroll
      for (int c8 = 0; c8 < G_BK / 8; ++c8) {
        const float4 a4 = *reinterpret_cast<const float4*>(s_a + (wm * 32 + lr) * G_LD + c8 * 8 + 4 * lh);
        const float4 b4 = *reinterpret_cast<const float4*>(s_b + (wn * 32 + lr) * G_LD + c8 * 8 + 4 * lh);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, NS - 1>;
  if (kbeg < kend) fetch(S0{}, kbeg);
  if (kbeg + G_BK < kend) fetch(S1{}, kbeg + G_BK);
  if constexpr (NS == 3) {
    if (kbeg + 2 * G_BK < kend) fetch(S2{}, kbeg + 2 * G_BK);
  }
  for (int k0 = kbeg; k0 < kend; k0 += NS * G_BK) {
    step(S0{}, k0);
    if (k0 + G_BK < kend) step(S1{}, k0 + G_BK);
    if constexpr (NS == 3) {
      if (k0 + 2 * G_BK < kend) step(S2{}, k0 + 2 * G_BK);
    }
  }
  // epilogue: C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int n = n0 + wn * 32 + lr;
  if (n >= p.N) return;
  float bias = 0.f;
  if (p.bias != nullptr && sp == 0) bias = p.bias[(size_t)z * p.sbias_b + n];
  const size_t oc = (size_t)zo * p.scb + (size_t)zi * p.scb2;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < p.M) gemm_store(p, z, oc, sp, m, n, bias, acc[r]);
  }
}

// ---- few-tile launches: 32 x 32 output tiles, the K step of 64 split over the 4 waves ---------------
// A 2048 x 256 output is 128 tiles of 64 x 64 - half the CUs, one workgroup each, every K step a
// serial load -> LDS -> MFMA round.  Here a workgroup owns a 32 x 32 tile (4x as many workgroups,
// several per CU) and its 4 waves each contract a quarter of the staged K range; the four partial
// accumulators are folded through LDS and each wave applies the epilogue to 8 of the 32 rows.
struct GemmRegsS {
  float4 v[2], w[2];
};

template <int mode, bool ADD2>
__device__ __forceinline__ void gemm_fetch_s(GemmRegsS& g, float (&sc)[8], const float* __restrict__ P,
                                             const float* __restrict__ P2, long long sr, long long sk,
                                             int r0, int R, int k0, int K1) {
  const int t = threadIdx.x;
  if constexpr (mode == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (t >> 4) + 16 * i, kq = (t & 15) * 4;
      g.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (ADD2) g.w[i] = g.v[i];
      if (r0 + row < R && k0 + kq < K1) {
        const size_t o = (size_t)(r0 + row) * sr + k0 + kq;
        g.v[i] = *reinterpret_cast<const float4*>(P + o);
        if constexpr (ADD2) g.w[i] = *reinterpret_cast<const float4*>(P2 + o);
      }
    }
  } else if constexpr (mode == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = (t >> 3) + 32 * i, rq = (t & 7) * 4;
      g.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (ADD2) g.w[i] = g.v[i];
      if (k0 + k < K1 && r0 + rq < R) {
        const size_t o = (size_t)(k0 + k) * sk + r0 + rq;
        g.v[i] = *reinterpret_cast<const float4*>(P + o);
        if constexpr (ADD2) g.w[i] = *reinterpret_cast<const float4*>(P2 + o);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = t + 256 * i, row = e >> 6, k = e & 63;
      float v = 0.f;
      if (r0 + row < R && k0 + k < K1) {
        const size_t o = (size_t)(r0 + row) * sr + (size_t)(k0 + k) * sk;
        v = P[o];
        if constexpr (ADD2) v += P2[o];
      }
      sc[i] = v;
    }
  }
}

template <int mode, bool ADD2, bool BF16>
__device__ __forceinline__ void gemm_commit_s(float* __restrict__ s, const GemmRegsS& g, const float (&sc)[8]) {
  const int t = threadIdx.x;
  char* sb = reinterpret_cast<char*>(s);
  if constexpr (mode == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v = ADD2 ? add4(g.v[i], g.w[i]) : g.v[i];
      const int row = (t >> 4) + 16 * i, kq = (t & 15) * 4;
      if constexpr (BF16) {
        bf16x4 b;
        b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(sb + row * G_LDB + 2 * kq) = b;
      } else {
        *reinterpret_cast<float4*>(s + row * G_LD + kq) = v;
      }
    }
  } else if constexpr (mode == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v = ADD2 ? add4(g.v[i], g.w[i]) : g.v[i];
      const int k = (t >> 3) + 32 * i, rq = (t & 7) * 4;
      if constexpr (BF16) {
        char* d = sb + rq * G_LDB + 2 * k;
        *reinterpret_cast<__bf16*>(d) = (__bf16)v.x;
        *reinterpret_cast<__bf16*>(d + G_LDB) = (__bf16)v.y;
        *reinterpret_cast<__bf16*>(d + 2 * G_LDB) = (__bf16)v.z;
        *reinterpret_cast<__bf16*>(d + 3 * G_LDB) = (__bf16)v.w;
      } else {
        float* d = s + rq * G_LD + k;
        d[0] = v.x; d[G_LD] = v.y; d[2 * G_LD] = v.z; d[3 * G_LD] = v.w;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = t + 256 * i;
      if constexpr (BF16) *reinterpret_cast<__bf16*>(sb + (e >> 6) * G_LDB + 2 * (e & 63)) = (__bf16)sc[i];
      else s[(e >> 6) * G_LD + (e & 63)] = sc[i];
    }
  }
}

template <int MA, int MB, bool ADD, bool BF16>
__device__ __forceinline__ void gemm_ks_body(const GemmArgs& p, int bx, int by, int bz, float* __restrict__ s_a,
                                             float* __restrict__ s_b, float (*s_red)[16][64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = by * 32, n0 = bx * 32;
  const int z = bz / p.splitk, sp = bz - z * p.splitk;
  // asum (weight-gradient launches, A reduction-strided): row sums of the A operand = the bias gradient
  // of the linear layer, taken by the first column of workgroups from the rows they stage anyway
  const bool do_asum = MA == 2 && p.asum != nullptr && bx == 0;
  float4 rs4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int kchunk = ((p.K + p.splitk - 1) / p.splitk + G_BK - 1) / G_BK * G_BK;
  const int kbeg = sp * kchunk, kend = min(p.K, kbeg + kchunk);
  const int zo = z / p.zdiv, zi = z - zo * p.zdiv;
  const size_t oa = (size_t)zo * p.sab + (size_t)zi * p.sab2, ob = (size_t)zo * p.sbb + (size_t)zi * p.sbb2;
  const float* A = p.A + oa;
  const float* A2 = (p.A2 != nullptr && n0 < p.a2_cols) ? p.A2 + oa : nullptr;
  const float* B = p.B + ob;
  const float* B2 = (p.B2 != nullptr && m0 < p.b2_rows) ? p.B2 + ob : nullptr;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int NS = 3;
  GemmRegsS ga[NS], gb[NS];
  float sa8[NS][8], sb8[NS][8];
  auto fetch = [&](auto stage, int k0) {
    constexpr int S = decltype(stage)::value;
    if (ADD && A2 != nullptr) gemm_fetch_s<MA, ADD>(ga[S], sa8[S], A, A2, p.sam, p.sak, m0, p.M, k0, kend);
    else gemm_fetch_s<MA, false>(ga[S], sa8[S], A, nullptr, p.sam, p.sak, m0, p.M, k0, kend);
    if (ADD && B2 != nullptr) gemm_fetch_s<MB, ADD>(gb[S], sb8[S], B, B2, p.sbn, p.sbk, n0, p.N, k0, kend);
    else gemm_fetch_s<MB, false>(gb[S], sb8[S], B, nullptr, p.sbn, p.sbk, n0, p.N, k0, kend);
  };
  auto step = [&](auto stage, int k0) {
    constexpr int S = decltype(stage)::value;
    lds_barrier();
    if (ADD && A2 != nullptr) gemm_commit_s<MA, ADD, BF16>(s_a, ga[S], sa8[S]); else gemm_commit_s<MA, false, BF16>(s_a, ga[S], sa8[S]);
    if constexpr (MA == 2) {
      if (do_asum) {           // this thread's two float4 = 4 consecutive rows m at two reduction steps
        rs4 = add4(rs4, add4(ga[S].v[0], ga[S].v[1]));
      }
    }
    if (ADD && B2 != nullptr) gemm_commit_s<MB, ADD, BF16>(s_b, gb[S], sb8[S]); else gemm_commit_s<MB, false, BF16>(s_b, gb[S], sb8[S]);
    lds_barrier();
    if (k0 + NS * G_BK < kend) fetch(stage, k0 + NS * G_BK);
    // this wave's quarter of the K step: k = 16*wave .. 16*wave + 15
    if constexpr (BF16) {
      const char* ca = reinterpret_cast<const char*>(s_a);
      const char* cb = reinterpret_cast<const char*>(s_b);
      const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(ca + lr * G_LDB + 2 * (wave * 16 + 8 * lh));
      const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(cb + lr * G_LDB + 2 * (wave * 16 + 8 * lh));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int c8 = 0; c8 < 2; ++c8) {
        const float4 a4 = *reinterpret_cast<const float4*>(s_a + lr * G_LD + wave * 16 + c8 * 8 + 4 * lh);
        const float4 b4 = *reinterpret_cast<const float4*>(s_b + lr * G_LD + wave * 16 + c8 * 8 + 4 * lh);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  if (kbeg < kend) fetch(S0{}, kbeg);
  if (kbeg + G_BK < kend) fetch(S1{}, kbeg + G_BK);
  if (kbeg + 2 * G_BK < kend) fetch(S2{}, kbeg + 2 * G_BK);
  for (int k0 = kbeg; k0 < kend; k0 += NS * G_BK) {
    step(S0{}, k0);
    if (k0 + G_BK < kend) step(S1{}, k0 + G_BK);
    if (k0 + 2 * G_BK < kend) step(S2{}, k0 + 2 * G_BK);
  }
  // fold the 4 partial tiles; wave q finishes registers 4q..4q+3 = rows 8q + (0..3) + 4*lh
#pragma unroll
  for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[r];
  lds_barrier();
  const int n = n0 + lr;
  if (n < p.N) {
    float bias = 0.f;
    if (p.bias != nullptr && sp == 0) bias = p.bias[(size_t)z * p.sbias_b + n];
    const size_t oc = (size_t)zo * p.scb + (size_t)zi * p.scb2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * wave + q;
      const int m = m0 + q + 8 * wave + 4 * lh;
      const float v = (s_red[0][r][lane] + s_red[1][r][lane]) + (s_red[2][r][lane] + s_red[3][r][lane]);
      if (m < p.M) gemm_store(p, z, oc, sp, m, n, bias, v);
    }
  }
  if constexpr (MA == 2) {
    if (do_asum) {
      // thread t staged rows 4*(t & 7) .. +3 at reduction steps (t >> 3) + 32 i: 32 threads per row quad
      lds_barrier();
      float4* q4 = reinterpret_cast<float4*>(&s_red[0][0][0]);
      q4[threadIdx.x] = rs4;
      lds_barrier();
      if (threadIdx.x < 32 && m0 + (int)threadIdx.x < p.M) {
        const int rq = threadIdx.x >> 2, c = threadIdx.x & 3;
        const float* f = &s_red[0][0][0];
        float t = 0.f;
        for (int j = 0; j < 32; ++j) t += f[(8 * j + rq) * 4 + c];
        atomicAdd(p.asum + (size_t)z * p.M + m0 + threadIdx.x, t);
      }
    }
  }
}

template <int MA, int MB, bool ADD, bool BF16>
__global__ __launch_bounds__(256, ADD ? 2 : 4) void gemm_ks_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float s_a[32 * G_LD];
  __shared__ __attribute__((aligned(16))) float s_b[32 * G_LD];
  __shared__ float s_red[4][16][64];
  int bx = blockIdx.x, by = blockIdx.y;
  xcd_rows(bx, by, gridDim.x, gridDim.y);
  gemm_ks_body<MA, MB, ADD, BF16>(p, bx, by, blockIdx.z, s_a, s_b, s_red);
}

// Several small GEMMs of the SAME operand form in ONE launch (the weight gradients of a decoder layer:
// ten reduction-strided products that depend on nothing but saved tensors): the descriptors travel by
// value in the kernel arguments, a workgroup finds its problem from the tile prefix table.
constexpr int GEMM_GROUP_MAX = 12;
struct GemmGroup {
  GemmArgs p[GEMM_GROUP_MAX];
  int start[GEMM_GROUP_MAX + 1];
  int n;
  int dbg;
};
template <int MA, int MB, bool ADD, bool BF16>
__global__ __launch_bounds__(256, ADD ? 2 : 4) void gemm_ks_group_kernel(GemmGroup g) {
  __shared__ __attribute__((aligned(16))) float s_a[32 * G_LD];
  __shared__ __attribute__((aligned(16))) float s_b[32 * G_LD];
  __shared__ float s_red[4][16][64];
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && b >= g.start[i + 1]) ++i;
  const GemmArgs& p = g.p[i];
  const int local = b - g.start[i];
  const int gx = (p.N + 31) / 32, gy = (p.M + 31) / 32;
  // the K slice is the fastest index: with 8 slices (and tile counts that are multiples of 8) XCD i runs
  // slice i of every tile, so each reduction range of both operands lands in exactly one L2
  const int sp = local % p.splitk, rest = local / p.splitk;
  const int bx = rest % gx, by = (rest / gx) % gy, bz = (rest / (gx * gy)) * p.splitk + sp;
  gemm_ks_body<MA, MB, ADD, BF16>(p, bx, by, bz, s_a, s_b, s_red);
}

// ---- weight-gradient group on the bf16 matrix cores: 64 x 64 output tiles, fp32 operands as P bf16 planes ------
// C[m][n] += sum_k A[k * sak + m] * (B[k * sbk + n] (+ B2)), both operands contiguous ACROSS the reduction
// (dW = dY^T.X of a linear layer: m = output channel, n = input channel, k = row).  The 32 x 32-tile kernel above
// spends a launch of 7 552 workgroups on the ten weight gradients of a decoder layer (118 us, 268 MB fetched for
// 30 MB of operands: every tile re-reads 32-column strips, the fp32 MFMA issues 64 cycles per 32 x 32 x 2).
// Here a workgroup owns a 64 x 64 tile of one K slice; per 32-row step the two 32 x 64 fp32 strips go
// global -> registers (one step ahead), are split ONCE into P bf16 terms (P = 3: x = h + m + l exactly, the six
// products of weight >= 2^-16, as csrc/mlp.hip's fp32-grade mode; P = 1: rounded, the bf16 compute mode) and land as
// [column][step] planes in a double-buffered LDS tile (ONE barrier per step); a wave (2 x 2 over the tile) reads
// its fragments as ds_read_b128 and issues v_mfma_f32_32x32x16_bf16.  (A first version kept the fp32 strips in
// LDS and split per use: every value split by two waves, ~400 VALU cycles per 192 MFMA cycles - 65 us.)  Row sums of A (the bias gradient) ride on the staged registers of the n-tile-0
// workgroups.  The K slice is the fastest workgroup index (XCD i = slice i when splitk == 8).
using f32x2 = float __attribute__((ext_vector_type(2)));
using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
// two fp32 values -> P packed bf16 pairs (low half = a): one v_cvt_pk_bf16_f32 per plane, the residuals
// on both elements at once (x = h + m + l exactly for P = 3)
template <int P>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&o)[P]) {
  const f32x2 x = {a, b};
  const bf16x2 h = __builtin_convertvector(x, bf16x2);
  o[0] = __builtin_bit_cast(unsigned, h);
  if constexpr (P == 3) {
    const f32x2 r = x - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r, bf16x2);
    o[1] = __builtin_bit_cast(unsigned, m);
    const f32x2 l = r - __builtin_convertvector(m, f32x2);
    o[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2));
  }
}
// byte offset of 16-byte chunk `chunk` (8 reduction steps) of the 64-byte row of column `col`; chunks are
// XOR-swizzled by bits 2-3 of the column so that a fragment read (lane = column) touches every bank once
__device__ __forceinline__ int tn_swz(int col, int chunk) { return col * 64 + ((chunk ^ ((col >> 2) & 3)) << 4); }

template <int P>
__global__ __launch_bounds__(256, 3) void gemm_tn_group_kernel(GemmGroup g) {
  // [buffer][operand][plane][64 columns x 32 reduction steps] bf16
  __shared__ __attribute__((aligned(16))) char s_pl[2][2][P][4096];
  __shared__ float4 s_rs[128];
  const int b = blockIdx.x;
  // which problem: lane i compares against start[i + 1] (one load + a ballot instead of a chain of dependent loads)
  const int li = threadIdx.x & 63;
  const int gi = __builtin_amdgcn_readfirstlane(
      __popcll(__ballot(li + 1 < g.n && b >= g.start[li + 1 < GEMM_GROUP_MAX ? li + 1 : GEMM_GROUP_MAX])));
  const GemmArgs& p = g.p[gi];
  const int local = b - g.start[gi];
  const int gx = (p.N + 63) / 64, gy = (p.M + 63) / 64;
  const int sp = local % p.splitk, rest = local / p.splitk;
  const int bx = rest % gx, by = (rest / gx) % gy, z = rest / (gx * gy);
  const int m0 = by * 64, n0 = bx * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 31, lh = lane >> 5, wm = wave & 1, wn = wave >> 1;
  const int kchunk = ((p.K + p.splitk - 1) / p.splitk + 31) / 32 * 32;
  const int kbeg = sp * kchunk, kend = min(p.K, kbeg + kchunk);
  const int zo = z / p.zdiv, zi = z - zo * p.zdiv;
  const size_t ob = (size_t)zo * p.sbb + (size_t)zi * p.sbb2;
  // staging: waves 0-1 move the A strip, waves 2-3 the B strip; thread u = t & 127 takes reduction steps
  // 4 (u >> 4) .. + 3 of columns 4 (u & 15) .. + 3, splits them ONCE and leaves [column][step] planes in LDS
  const int opd = t >> 7, u = t & 127, rg = u >> 4, sc = (u & 15) * 4;
  const float* src = opd == 0 ? p.A + (size_t)zo * p.sab + (size_t)zi * p.sab2 + m0 + sc : p.B + ob + n0 + sc;
  const float* src2 = (opd == 1 && p.B2 != nullptr && m0 < p.b2_rows) ? p.B2 + ob + n0 + sc : nullptr;
  const long long sk = opd == 0 ? p.sak : p.sbk;
  const bool col_ok = opd == 0 ? m0 + sc < p.M : n0 + sc < p.N;        // M, N % 4 == 0 in this form
  const bool do_asum = p.asum != nullptr && bx == 0 && opd == 0;
  // two register stages: the loads of steps s + 1 and s + 2 are in flight while step s is contracted (a step's
  // MFMA phase is ~0.3 us, a load from another XCD's rows well over that); the second addend is added at commit
  // time, not at fetch time (an add right behind the loads would park the wave on them)
  float4 rv[2][4], rw[2][4], rs4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](auto stage, int k0) {
    constexpr int S = decltype(stage)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 4 * rg + j;
      rv[S][j] = rw[S][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col_ok && k < kend && !(g.dbg & 8)) {
        rv[S][j] = *reinterpret_cast<const float4*>(src + (size_t)k * sk);
        if (src2 != nullptr) rw[S][j] = *reinterpret_cast<const float4*>(src2 + (size_t)k * sk);
      }
    }
  };
  auto commit = [&](auto stage, int buf) {
    constexpr int S = decltype(stage)::value;
    if (src2 != nullptr) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rv[S][j] = add4(rv[S][j], rw[S][j]);
    }
    if (do_asum) rs4 = add4(rs4, add4(add4(rv[S][0], rv[S][1]), add4(rv[S][2], rv[S][3])));
    const float* f = reinterpret_cast<const float*>(rv[S]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned lo[P], hi[P];
      split_pair<P>(f[c], f[4 + c], lo);
      split_pair<P>(f[8 + c], f[12 + c], hi);
      const int off = tn_swz(sc + c, rg >> 1) + 8 * (rg & 1);
#pragma unroll
      for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(&s_pl[buf][opd][q][off]) = make_uint2(lo[q], hi[q]);
    }
  };
  f32x16 acc, acc1;       // two accumulator chains (one per 16-step chunk): dependent MFMAs wait on each other
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
  auto step = [&](auto stage, int k0, int buf) {
    if (!(g.dbg & 4)) commit(stage, buf);
    lds_barrier();
    if (k0 + 64 < kend) fetch(stage, k0 + 64);
    if (g.dbg & 2) return;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 pa[P], pb[P];
#pragma unroll
      for (int q = 0; q < P; ++q) {
        pa[q] = *reinterpret_cast<const bf16x8*>(&s_pl[buf][0][q][tn_swz(32 * wm + lc, 2 * c + lh)]);
        pb[q] = *reinterpret_cast<const bf16x8*>(&s_pl[buf][1][q][tn_swz(32 * wn + lc, 2 * c + lh)]);
      }
      f32x16& ac = c == 0 ? acc : acc1;
      if constexpr (P == 3) {   // smallest products first
        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[2], pb[0], ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], pb[2], ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1], pb[1], ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1], pb[0], ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], pb[1], ac, 0, 0, 0);
      }
      ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], pb[0], ac, 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (kbeg < kend) fetch(S0{}, kbeg);
  if (kbeg + 32 < kend) fetch(S1{}, kbeg + 32);
  for (int k0 = kbeg; k0 < kend; k0 += 64) {
    step(S0{}, k0, 0);
    if (k0 + 32 < kend) step(S1{}, k0 + 32, 1);
  }
  // accumulator register r = row (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32 x 32 tile, column lc
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
  const int n = n0 + 32 * wn + lc;
  if (n < p.N && !(g.dbg & 16)) {
    float* C = p.C + (size_t)zo * p.scb + (size_t)zi * p.scb2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < p.M) {
        float* dst = C + (size_t)m * p.scm + n;
        const float v = p.alpha * acc[r];
        if (g.dbg & 1) *dst = v;
        else if (p.splitk > 1) atomicAdd(dst, v);
        else if (p.flags & GEMM_ACCUM) *dst += v;
        else *dst = v;
      }
    }
  }
  if (p.asum != nullptr && bx == 0) {       // (workgroup-uniform)
    if (opd == 0) s_rs[u] = rs4;
    lds_barrier();
    if (t < 64 && m0 + t < p.M) {
      const float* f = reinterpret_cast<const float*>(s_rs);
      float v = 0.f;
      for (int j = 0; j < 8; ++j) v += f[(16 * j + (t >> 2)) * 4 + (t & 3)];
      atomicAdd(p.asum + (size_t)z * p.M + m0 + t, v);
    }
  }
}

// staging mode of an operand with rows R, reduction K, strides (sr, sk).  Global dwordx4 loads only
// need 4-byte alignment on gfx9 (parameters live at arbitrary float offsets of the flat optimizer
// buffer), so contiguity and a length that is a multiple of 4 are the only requirements.
static int env_tiles() {
  static const int v = getenv("DEMF_GEMM_SMALL_TILES") ? atoi(getenv("DEMF_GEMM_SMALL_TILES")) : 1024;
  return v;
}

static int stage_mode(long long sr, long long sk, int R, int K) {
  if (sk == 1 && K % 4 == 0) return 1;
  if (sr == 1 && R % 4 == 0) return 2;
  return 0;
}

// ---- dropout + residual + LayerNorm over rows of C <= 1024 channels (C % 64 == 0) ---------------
//   s = identity + dropout(x) ; y = (s - mean) * rstd * gamma + beta
// One wave per row, VPL = C/64 values per lane.  `s` is stored (it may alias x) for the backward.
template <int VPL>
__global__ __launch_bounds__(256) void add_dropout_ln_fwd_k(int R, const float* __restrict__ x,
                                                            const float* __restrict__ identity,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float p, const unsigned long long* __restrict__ rng,
                                                            unsigned op, float* __restrict__ s_out,
                                                            float* __restrict__ y,
                                                            float* __restrict__ stats) {
  constexpr int C = 64 * VPL;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float v[VPL];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    float xv = x[(size_t)row * C + c];
    if (p > 0.f) xv = dropout_keep(rng, op, (unsigned long long)row * C + c, p) ? xv * inv_keep : 0.f;
    v[i] = (identity != nullptr ? identity[(size_t)row * C + c] : 0.f) + xv;
    sum += v[i];
  }
  const float mean = group_allsum<64>(sum) * (1.0f / C);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) sq = __builtin_fmaf(v[i] - mean, v[i] - mean, sq);
  const float rstd = 1.0f / sqrtf(group_allsum<64>(sq) * (1.0f / C) + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    if (s_out != nullptr) s_out[(size_t)row * C + c] = v[i];
    y[(size_t)row * C + c] = __builtin_fmaf((v[i] - mean) * rstd, gamma[c], beta[c]);
  }
  if (lane == 0 && stats != nullptr) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// backward: ds = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma ;  dx = ds * keep/(1-p)
// dgamma += dy * xhat, dbeta += dy: a lane owns fixed columns over the block's rows, the four waves
// fold through LDS, one atomic per column per block.  `ds_accum`: ds is ADDED to what ds_out holds.
template <int VPL>
__global__ __launch_bounds__(256) void add_dropout_ln_bwd_k(int R, int rows_per_wave,
                                                            const float* __restrict__ dy,
                                                            const float* __restrict__ dy2,
                                                            const float* __restrict__ s,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, float p,
                                                            const unsigned long long* __restrict__ rng,
                                                            unsigned op, float* __restrict__ ds_out,
                                                            int ds_accum, float* __restrict__ dx_out,
                                                            float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  constexpr int C = 64 * VPL;
  __shared__ float s_red[4][2][C];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float ag[VPL], ab[VPL], gm[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) { ag[i] = ab[i] = 0.f; gm[i] = gamma[lane + 64 * i]; }
  const int row0 = (blockIdx.x * 4 + wave) * rows_per_wave;
  for (int row = row0; row < min(R, row0 + rows_per_wave); ++row) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float d[VPL], xh[VPL];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const size_t o = (size_t)row * C + lane + 64 * i;
      d[i] = dy[o] + (dy2 != nullptr ? dy2[o] : 0.f);
      xh[i] = (s[o] - mean) * rstd;
      ag[i] = __builtin_fmaf(d[i], xh[i], ag[i]);
      ab[i] += d[i];
      const float g = d[i] * gm[i];
      m1 += g;
      m2 = __builtin_fmaf(g, xh[i], m2);
    }
    m1 = group_allsum<64>(m1) * (1.0f / C);
    m2 = group_allsum<64>(m2) * (1.0f / C);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      const size_t o = (size_t)row * C + c;
      const float v = rstd * (d[i] * gm[i] - m1 - xh[i] * m2);
      if (ds_out != nullptr) ds_out[o] = ds_accum ? ds_out[o] + v : v;
      if (dx_out != nullptr) {
        float xv = v;
        if (p > 0.f) xv = dropout_keep(rng, op, (unsigned long long)row * C + c, p) ? v * inv_keep : 0.f;
        dx_out[o] = xv;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) { s_red[wave][0][lane + 64 * i] = ag[i]; s_red[wave][1][lane + 64 * i] = ab[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, s_red[0][0][c] + s_red[1][0][c] + s_red[2][0][c] + s_red[3][0][c]);
    atomicAdd(dbeta + c, s_red[0][1][c] + s_red[1][1][c] + s_red[2][1][c] + s_red[3][1][c]);
  }
}

// ---- attention softmax (+ dropout) over rows of S <= 1024 keys ------------------------------------
// prob = softmax(scores) (kept for the backward) ; out = dropout(prob).  VPL = ceil(S/64).
template <int VPL>
__global__ __launch_bounds__(256) void softmax_dropout_fwd_k(int R, int S, const float* __restrict__ sc,
                                                             float p, const unsigned long long* __restrict__ rng,
                                                             unsigned op, float* __restrict__ prob,
                                                             float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float v[VPL];
  float mx = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < S ? sc[(size_t)row * S + c] : -__builtin_inff();
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_allmax(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) { v[i] = lane + 64 * i < S ? __expf(v[i] - mx) : 0.f; sum += v[i]; }
  const float inv = 1.0f / group_allsum<64>(sum);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (lane + 64 * i >= S) continue;
    const size_t o = (size_t)row * S + lane + 64 * i;
    const float pr = v[i] * inv;
    prob[o] = pr;
    float q = pr;
    if (p > 0.f) q = dropout_keep(rng, op, o, p) ? pr * inv_keep : 0.f;
    out[o] = q;
  }
}

// dscores = prob * (dprob - sum(dprob * prob)),  dprob = dout * keep / (1-p)   (in place over dout)
template <int VPL>
__global__ __launch_bounds__(256) void softmax_dropout_bwd_k(int R, int S, const float* __restrict__ prob,
                                                             float p, const unsigned long long* __restrict__ rng,
                                                             unsigned op, float* __restrict__ dio) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float d[VPL], pr[VPL];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    d[i] = pr[i] = 0.f;
    if (lane + 64 * i >= S) continue;
    const size_t o = (size_t)row * S + lane + 64 * i;
    d[i] = dio[o];
    pr[i] = prob[o];
    if (p > 0.f) d[i] = dropout_keep(rng, op, o, p) ? d[i] * inv_keep : 0.f;
    dot = __builtin_fmaf(d[i], pr[i], dot);
  }
  dot = group_allsum<64>(dot);
#pragma unroll
  for (int i = 0; i < VPL; ++i)
    if (lane + 64 * i < S) dio[(size_t)row * S + lane + 64 * i] = pr[i] * (d[i] - dot);
}

// ---- sampling-location preparation of the fusion attention ----------------------------------------
// Reference: DeMFVoteHead.get_reference_points (class_agnostic_vote_head.py:524-547: undo the 3-D
// augmentation, depth2img, 2-D scale / flip, /(W-1,H-1), clamp[0,1] - composed on the host into one
// 4x4 M and (au,bu,av,bv) per scene), the valid-ratio scaling of DeMFTransformerDecoderLayer.forward
// (transformer.py:62-68) and mmcv MultiScaleDeformableAttention.forward: offsets / (W_l, H_l) added to
// the reference point, softmax of the H x (L*P) attention logits.
//   raw (R, H*L*P*3): [offsets (H,L,P,2) | logits (H,L,P)] from one projection of query + pos
//   -> loc (R,H,L,P,2), w (R,H,L,P), uv (R,2) (clamped reference point, kept for the backward)
// one thread per (row, head); LP = L*P <= 16
__global__ void msda_prep_fwd_k(int R, int Q, int H, int L, int P, const float* __restrict__ pts,
                                const float* __restrict__ M, const float* __restrict__ ab,
                                const float* __restrict__ vr, const long long* __restrict__ shapes,
                                const float* __restrict__ raw, float* __restrict__ loc,
                                float* __restrict__ w, float* __restrict__ uvw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * H) return;
  const int row = t / H, h = t - row * H, b = row / Q;
  const float x = pts[3 * row], y = pts[3 * row + 1], zc = pts[3 * row + 2];
  const float* m = M + 16 * b;
  const float px = m[0] * x + m[1] * y + m[2] * zc + m[3];
  const float py = m[4] * x + m[5] * y + m[6] * zc + m[7];
  const float pw = m[8] * x + m[9] * y + m[10] * zc + m[11];
  const float u0 = px / pw * ab[4 * b] + ab[4 * b + 1];
  const float v0 = py / pw * ab[4 * b + 2] + ab[4 * b + 3];
  const float u = fminf(fmaxf(u0, 0.f), 1.f), v = fminf(fmaxf(v0, 0.f), 1.f);
  if (h == 0) {
    uvw[4 * row] = u0; uvw[4 * row + 1] = v0; uvw[4 * row + 2] = px / pw; uvw[4 * row + 3] = py / pw;
  }
  const int LP = L * P, HLP = H * LP;
  const float* off = raw + (size_t)row * HLP * 3 + h * LP * 2;
  const float* lg = raw + (size_t)row * HLP * 3 + HLP * 2 + h * LP;
  float mx = -__builtin_inff();
  for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lg[i]);
  float e[16], sum = 0.f;
  for (int i = 0; i < LP; ++i) { e[i] = __expf(lg[i] - mx); sum += e[i]; }
  const float inv = 1.0f / sum;
  for (int l = 0; l < L; ++l) {
    const float rx = u * vr[(b * L + l) * 2], ry = v * vr[(b * L + l) * 2 + 1];
    const float iw = 1.0f / (float)shapes[2 * l + 1], ih = 1.0f / (float)shapes[2 * l];
    for (int q = 0; q < P; ++q) {
      const int i = l * P + q;
      const size_t o = ((size_t)row * H + h) * LP + i;
      loc[2 * o] = rx + off[2 * i] * iw;
      loc[2 * o + 1] = ry + off[2 * i + 1] * ih;
      w[o] = e[i] * inv;
    }
  }
}

// backward: draw (offsets, logits) per (row, head); dpts (R,3) accumulated per row over heads by the
// thread of head 0 looping the heads (R threads do the reduction: it is tiny)
__global__ void msda_prep_bwd_k(int R, int Q, int H, int L, int P, const float* __restrict__ pts,
                                const float* __restrict__ M, const float* __restrict__ ab,
                                const float* __restrict__ vr, const long long* __restrict__ shapes,
                                const float* __restrict__ w, const float* __restrict__ uvw,
                                const float* __restrict__ dloc, const float* __restrict__ dloc2,
                                const float* __restrict__ dw, const float* __restrict__ dw2,
                                float* __restrict__ draw, float* __restrict__ dpts) {
  // (dloc2, dw2): optional second contribution (the keep-mask sampling of sample-then-project)
  auto DL = [&](size_t i) { return dloc[i] + (dloc2 != nullptr ? dloc2[i] : 0.f); };
  auto DW = [&](size_t i) { return dw[i] + (dw2 != nullptr ? dw2[i] : 0.f); };
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * H) return;
  const int row = t / H, h = t - row * H, b = row / Q;
  const int LP = L * P, HLP = H * LP;
  float* doff = draw + (size_t)row * HLP * 3 + h * LP * 2;
  float* dlg = draw + (size_t)row * HLP * 3 + HLP * 2 + h * LP;
  const size_t base = ((size_t)row * H + h) * LP;
  float dot = 0.f;
  for (int i = 0; i < LP; ++i) dot = __builtin_fmaf(DW(base + i), w[base + i], dot);
  for (int l = 0; l < L; ++l) {
    const float iw = 1.0f / (float)shapes[2 * l + 1], ih = 1.0f / (float)shapes[2 * l];
    for (int q = 0; q < P; ++q) {
      const int i = l * P + q;
      doff[2 * i] = DL(2 * (base + i)) * iw;
      doff[2 * i + 1] = DL(2 * (base + i) + 1) * ih;
      dlg[i] = w[base + i] * (DW(base + i) - dot);
    }
  }
  if (dpts == nullptr) return;
  // d(u,v): sum over heads / levels / points of dloc * valid ratio, gated by the clamp.  The H threads
  // of a row are neighbours in a wave (H a power of two <= 64, 256 % H == 0): each sums its own head,
  // a butterfly over the H lanes adds them; any other H: the thread of head 0 loops the heads.
  float du = 0.f, dv = 0.f;
  const bool pow2 = (H & (H - 1)) == 0 && H <= 64;
  if (pow2) {
    for (int l = 0; l < L; ++l) {
      const float vx = vr[(b * L + l) * 2], vy = vr[(b * L + l) * 2 + 1];
      for (int q = 0; q < P; ++q) {
        const size_t o = base + l * P + q;
        du = __builtin_fmaf(DL(2 * o), vx, du);
        dv = __builtin_fmaf(DL(2 * o + 1), vy, dv);
      }
    }
    for (int m = 1; m < H; m <<= 1) { du += __shfl_xor(du, m, 64); dv += __shfl_xor(dv, m, 64); }
    if (h != 0) return;
  } else {
    if (h != 0) return;
    for (int hh = 0; hh < H; ++hh)
      for (int l = 0; l < L; ++l) {
        const float vx = vr[(b * L + l) * 2], vy = vr[(b * L + l) * 2 + 1];
        for (int q = 0; q < P; ++q) {
          const size_t o = ((size_t)row * H + hh) * LP + l * P + q;
          du = __builtin_fmaf(DL(2 * o), vx, du);
          dv = __builtin_fmaf(DL(2 * o + 1), vy, dv);
        }
      }
  }
  const float u0 = uvw[4 * row], v0 = uvw[4 * row + 1], xw = uvw[4 * row + 2], yw = uvw[4 * row + 3];
  if (!(u0 >= 0.f && u0 <= 1.f)) du = 0.f;          // torch.clamp passes the gradient on [min, max]
  if (!(v0 >= 0.f && v0 <= 1.f)) dv = 0.f;
  const float x = pts[3 * row], y = pts[3 * row + 1], zc = pts[3 * row + 2];
  const float* m = M + 16 * b;
  const float pw = m[8] * x + m[9] * y + m[10] * zc + m[11];
  const float gx = du * ab[4 * b] / pw, gy = dv * ab[4 * b + 2] / pw;     // d / d(px), d / d(py)
  const float gw = -(gx * xw + gy * yw);                                   // d / d(pw)
  dpts[3 * row] = gx * m[0] + gy * m[4] + gw * m[8];
  dpts[3 * row + 1] = gx * m[1] + gy * m[5] + gw * m[9];
  dpts[3 * row + 2] = gx * m[2] + gy * m[6] + gw * m[10];
}

// ---- row L2 normalisation (VoteModule norm_feats) -------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void l2norm_fwd_k(int R, const float* __restrict__ x,
                                                    float* __restrict__ y, float* __restrict__ norm) {
  constexpr int C = 64 * VPL;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  float v[VPL], sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) { v[i] = x[(size_t)row * C + lane + 64 * i]; sq = __builtin_fmaf(v[i], v[i], sq); }
  const float n = sqrtf(group_allsum<64>(sq));
#pragma unroll
  for (int i = 0; i < VPL; ++i) y[(size_t)row * C + lane + 64 * i] = v[i] / n;
  if (lane == 0) norm[row] = n;
}

template <int VPL>
__global__ __launch_bounds__(256) void l2norm_bwd_k(int R, const float* __restrict__ y,
                                                    const float* __restrict__ norm,
                                                    const float* __restrict__ dy, float* __restrict__ dx) {
  constexpr int C = 64 * VPL;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  float yv[VPL], g[VPL], dot = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    yv[i] = y[(size_t)row * C + lane + 64 * i];
    g[i] = dy[(size_t)row * C + lane + 64 * i];
    dot = __builtin_fmaf(yv[i], g[i], dot);
  }
  dot = group_allsum<64>(dot);
  const float inv = 1.0f / norm[row];
#pragma unroll
  for (int i = 0; i < VPL; ++i) dx[(size_t)row * C + lane + 64 * i] = (g[i] - yv[i] * dot) * inv;
}

// ---- VoteModule tail as one pass each way (class_agnostic_vote_head.py:394-403 -> mmdet3d VoteModule
// with vote_per_seed = 1, with_res_feat, norm_feats): votes (R, 3 + C) = conv_out's output rows;
//   vote_xyz = seed_xyz + votes[:, :3];   s = rows + votes[:, 3:];   y = s / |s|
template <int VPL>
__global__ __launch_bounds__(256) void vote_combine_fwd_k(int R, const float* __restrict__ rows,
                                                          const float* __restrict__ votes,
                                                          const float* __restrict__ seed_xyz,
                                                          float* __restrict__ vote_xyz,
                                                          float* __restrict__ y, float* __restrict__ norm) {
  constexpr int C = 64 * VPL;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* vr = votes + (size_t)row * (C + 3);
  float v[VPL], sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = rows[(size_t)row * C + lane + 64 * i] + vr[3 + lane + 64 * i];
    sq = __builtin_fmaf(v[i], v[i], sq);
  }
  const float n = sqrtf(group_allsum<64>(sq));
#pragma unroll
  for (int i = 0; i < VPL; ++i) y[(size_t)row * C + lane + 64 * i] = v[i] / n;
  if (lane == 0) norm[row] = n;
  if (lane < 3) vote_xyz[(size_t)row * 3 + lane] = seed_xyz[(size_t)row * 3 + lane] + vr[lane];
}

// dvotes (R, 3 + C) is written completely: [:, :3] = dxyz (or 0), [:, 3:] = drows = the gradient of s
template <int VPL>
__global__ __launch_bounds__(256) void vote_combine_bwd_k(int R, const float* __restrict__ y,
                                                          const float* __restrict__ norm,
                                                          const float* __restrict__ dy,
                                                          const float* __restrict__ dxyz,
                                                          float* __restrict__ dvotes,
                                                          float* __restrict__ drows) {
  constexpr int C = 64 * VPL;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  float yv[VPL], g[VPL], dot = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    yv[i] = y[(size_t)row * C + lane + 64 * i];
    g[i] = dy != nullptr ? dy[(size_t)row * C + lane + 64 * i] : 0.f;
    dot = __builtin_fmaf(yv[i], g[i], dot);
  }
  dot = group_allsum<64>(dot);
  const float inv = 1.0f / norm[row];
  float* dv = dvotes + (size_t)row * (C + 3);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float d = (g[i] - yv[i] * dot) * inv;
    dv[3 + lane + 64 * i] = d;
    drows[(size_t)row * C + lane + 64 * i] = d;
  }
  if (lane < 3) dv[lane] = dxyz != nullptr ? dxyz[(size_t)row * 3 + lane] : 0.f;
}

__global__ void rng_advance_k(unsigned long long* rng) { rng[1] += 1; }
// advance + snapshot: the forward of a dropout-carrying node draws its own step and hands the backward
// the (seed, step) pair it used, so two forwards before one backward re-derive the right masks
__global__ void rng_next_k(unsigned long long* rng, unsigned long long* snap) {
  const unsigned long long s = rng[1] + 1;
  rng[1] = s;
  snap[0] = rng[0];
  snap[1] = s;
}

__global__ void dropout_mask_k(long long n, float p, const unsigned long long* __restrict__ rng,
                               unsigned op, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = dropout_keep(rng, op, (unsigned long long)i, p) ? 1.0f / (1.0f - p) : 0.f;
}

}  // namespace demf

using namespace demf;

static int gemm_validate(const GemmArgs& p) {
  DEMF_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.batch > 0 && p.splitk > 0 && p.zdiv > 0,
               "gemm: bad sizes %d %d %d %d %d %d", p.M, p.N, p.K, p.batch, p.splitk, p.zdiv);
  DEMF_REQUIRE(p.A && p.B && p.C, "gemm: null operand");
  DEMF_REQUIRE(!(p.flags & GEMM_DROPOUT) || p.rng, "gemm: dropout needs the rng state");
  DEMF_REQUIRE(!(p.flags & GEMM_GATE) || p.gate, "gemm: gate flag without a gate tensor");
  DEMF_REQUIRE(!(p.flags & GEMM_ROWBIAS) || (p.rowscale && p.bias), "gemm: rowbias needs bias and rowscale");
  DEMF_REQUIRE(p.splitk == 1 || !(p.flags & (GEMM_RELU | GEMM_DROPOUT | GEMM_GATE)),
               "gemm: split-K cannot carry a non-linear epilogue");
  DEMF_REQUIRE((long long)p.batch * p.splitk <= 65535, "gemm: batch*splitk too large");
  return DEMF_OK;
}

static bool gemm_is_small(const GemmArgs& p) {
  const long long tiles = (long long)cdiv(p.N, G_BN) * cdiv(p.M, G_BM) * p.batch * p.splitk;
  return tiles <= env_tiles() || p.N <= 32;
}

extern "C" int demf_gemm_f32(const demf_gemm_desc* d, demf_stream_t stream) {
  DEMF_REQUIRE(d != nullptr, "gemm: null descriptor");
  GemmArgs p = *d;
  if (int e = gemm_validate(p)) return e;
  const int modeA = stage_mode(p.sam, p.sak, p.M, p.K);
  const int modeB = stage_mode(p.sbn, p.sbk, p.N, p.K);
  dim3 grid(cdiv(p.N, G_BN), cdiv(p.M, G_BM), p.batch * p.splitk);
  const bool add = p.A2 != nullptr || p.B2 != nullptr;
  const bool bf = compute_bf16() && !(p.flags & GEMM_FP32);
  // few 64 x 64 tiles (or a narrow output): 32 x 32 tiles with the K step split over the waves
  const bool small = gemm_is_small(p);
  if (p.asum != nullptr && !(small && modeA == 2 && p.A2 == nullptr)) {
    // The in-kernel row sums ride on the small-tile, reduction-strided staging only.  Any other launch
    // shape (more than DEMF_GEMM_SMALL_TILES tiles: a wider FFN, a long ops.linear; a lowered A/B knob)
    // runs the product without them and takes the sums - the column sums of the (K x M) row-major
    // matrix behind A - as demf_colsum_f32 launches, so a caller never has to mirror this predicate.
    DEMF_REQUIRE(p.sam == 1 && p.A2 == nullptr,
                 "gemm: asum needs A contiguous along M (sam == 1) and no A2");
    float* asum = p.asum;
    p.asum = nullptr;
    if (int e = demf_gemm_f32(&p, stream)) return e;
    for (int z = 0; z < p.batch; ++z) {
      const int zo = z / p.zdiv, zi = z - zo * p.zdiv;
      const float* a = p.A + (size_t)zo * p.sab + (size_t)zi * p.sab2;
      if (int e = demf_colsum_f32(p.K, p.M, (int)p.sak, a, asum + (size_t)z * p.M, stream)) return e;
    }
    return DEMF_OK;
  }
  if (small) {
    grid = dim3(cdiv(p.N, 32), cdiv(p.M, 32), p.batch * p.splitk);
#define GEMM_LAUNCH_S(MA_, MB_)                                                                        \
  do {                                                                                                 \
    if (bf) {                                                                                          \
      if (add) hipLaunchKernelGGL((gemm_ks_kernel<MA_, MB_, true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);   \
      else hipLaunchKernelGGL((gemm_ks_kernel<MA_, MB_, false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);      \
    } else {                                                                                           \
      if (add) hipLaunchKernelGGL((gemm_ks_kernel<MA_, MB_, true, false>), grid, dim3(256), 0, (hipStream_t)stream, p);  \
      else hipLaunchKernelGGL((gemm_ks_kernel<MA_, MB_, false, false>), grid, dim3(256), 0, (hipStream_t)stream, p);     \
    }                                                                                                  \
  } while (0)
    switch (modeA * 3 + modeB) {
      case 0: GEMM_LAUNCH_S(0, 0); break;
      case 1: GEMM_LAUNCH_S(0, 1); break;
      case 2: GEMM_LAUNCH_S(0, 2); break;
      case 3: GEMM_LAUNCH_S(1, 0); break;
      case 4: GEMM_LAUNCH_S(1, 1); break;
      case 5: GEMM_LAUNCH_S(1, 2); break;
      case 6: GEMM_LAUNCH_S(2, 0); break;
      case 7: GEMM_LAUNCH_S(2, 1); break;
      default: GEMM_LAUNCH_S(2, 2); break;
    }
#undef GEMM_LAUNCH_S
    return check_launch("gemm_ks_kernel");
  }
#define GEMM_LAUNCH(MA_, MB_)                                                                          \
  do {                                                                                                 \
    if (bf) {                                                                                          \
      if (add) hipLaunchKernelGGL((gemm_kernel<MA_, MB_, true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);   \
      else hipLaunchKernelGGL((gemm_kernel<MA_, MB_, false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);      \
    } else {                                                                                           \
      if (add) hipLaunchKernelGGL((gemm_kernel<MA_, MB_, true, false>), grid, dim3(256), 0, (hipStream_t)stream, p);  \
      else hipLaunchKernelGGL((gemm_kernel<MA_, MB_, false, false>), grid, dim3(256), 0, (hipStream_t)stream, p);     \
    }                                                                                                  \
  } while (0)
  switch (modeA * 3 + modeB) {
    case 0: GEMM_LAUNCH(0, 0); break;
    case 1: GEMM_LAUNCH(0, 1); break;
    case 2: GEMM_LAUNCH(0, 2); break;
    case 3: GEMM_LAUNCH(1, 0); break;
    case 4: GEMM_LAUNCH(1, 1); break;
    case 5: GEMM_LAUNCH(1, 2); break;
    case 6: GEMM_LAUNCH(2, 0); break;
    case 7: GEMM_LAUNCH(2, 1); break;
    default: GEMM_LAUNCH(2, 2); break;
  }
#undef GEMM_LAUNCH
  return check_launch("gemm_kernel");
}

extern "C" int demf_gemm_group_f32(const demf_gemm_desc* descs, int n, demf_stream_t stream) {
  DEMF_REQUIRE(descs != nullptr && n >= 1, "gemm_group: no descriptors");
  const bool bf = compute_bf16();
  // modes 1 / 2: the 64 x 64-tile kernel on the bf16 matrix cores (one / three terms per operand); mode 0
  // (native fp32 MFMA) keeps the 32 x 32-tile kernel
  static const bool tn_off = getenv("DEMF_GEMM_TN3") && atoi(getenv("DEMF_GEMM_TN3")) == 0;   // A/B switch
  const bool tn3 = compute_mode() != 0 && !tn_off;
  static const int tn_split = getenv("DEMF_GEMM_TN_SPLIT") ? atoi(getenv("DEMF_GEMM_TN_SPLIT")) : 0;   // A/B: forced K split
  int i0 = 0;
  while (i0 < n) {
    // longest run of consecutive problems that can share one launch: small-tile path, both operands
    // reduction-strided (the weight-gradient form), no non-linear epilogue flags that need other staging
    GemmGroup g{};
    int cnt = 0, tiles = 0;
    while (i0 + cnt < n && cnt < GEMM_GROUP_MAX) {
      const GemmArgs& p = descs[i0 + cnt];
      if (int e = gemm_validate(p)) return e;
      const bool ok = gemm_is_small(p) && stage_mode(p.sam, p.sak, p.M, p.K) == 2 &&
                      stage_mode(p.sbn, p.sbk, p.N, p.K) == 2 && !(p.flags & GEMM_FP32) && p.A2 == nullptr &&
                      // the 64 x 64-tile kernel carries alpha, accumulation and the row sums, no other epilogue
                      (!tn3 || ((p.flags & ~GEMM_ACCUM) == 0 && p.bias == nullptr && p.C2 == nullptr &&
                                (p.B2 == nullptr || p.b2_rows % 64 == 0 || p.b2_rows >= p.M)));
      if (!ok) break;
      g.p[cnt] = p;
      g.start[cnt] = tiles;
      tiles += (tn3 ? cdiv(p.N, 64) * cdiv(p.M, 64) : cdiv(p.N, 32) * cdiv(p.M, 32)) * p.batch * g.p[cnt].splitk;
      ++cnt;
    }
    if (cnt == 0) {                       // not groupable: its own launch
      if (int e = demf_gemm_f32(descs + i0, stream)) return e;
      ++i0;
      continue;
    }
    if (tn3) {
      // K split of the run: one resident round of workgroups (3 per CU) instead of the callers' 8 slices -
      // every extra slice repeats the launch ramp and the atomic epilogue (measured on the ten weight
      // gradients of a decoder layer, 252 tiles: 8 slices 60 us, 3 slices 51 us, 2 slices 56 us)
      int base = 0;
      for (int i = 0; i < cnt; ++i) base += cdiv(g.p[i].N, 64) * cdiv(g.p[i].M, 64) * g.p[i].batch;
      const int want = tn_split > 0 ? tn_split : (768 + base / 2) / base;
      tiles = 0;
      for (int i = 0; i < cnt; ++i) {
        GemmArgs& q = g.p[i];
        if (q.splitk > 1) q.splitk = want < 2 ? 2 : (want < q.splitk ? want : q.splitk);   // (>= 2: C stays accumulated)
        g.start[i] = tiles;
        tiles += cdiv(q.N, 64) * cdiv(q.M, 64) * q.batch * q.splitk;
      }
    }
    g.start[cnt] = tiles;
    g.n = cnt;
    static const int tn_dbg = getenv("DEMF_TN_DBG") ? atoi(getenv("DEMF_TN_DBG")) : 0;   // phase-skip timing experiments
    g.dbg = tn_dbg;
    if (tn3) {
      if (bf) hipLaunchKernelGGL((gemm_tn_group_kernel<1>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, g);
      else hipLaunchKernelGGL((gemm_tn_group_kernel<3>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, g);
      if (int e = check_launch("gemm_tn_group_kernel")) return e;
      i0 += cnt;
      continue;
    }
    if (bf) hipLaunchKernelGGL((gemm_ks_group_kernel<2, 2, true, true>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((gemm_ks_group_kernel<2, 2, true, false>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, g);
    if (int e = check_launch("gemm_ks_group_kernel")) return e;
    i0 += cnt;
  }
  return DEMF_OK;
}

#define LN_DISPATCH(C, CALL)                       \
  switch ((C) / 64) {                              \
    case 1: CALL(1); break;                        \
    case 2: CALL(2); break;                        \
    case 4: CALL(4); break;                        \
    case 8: CALL(8); break;                        \
    case 16: CALL(16); break;                      \
    default:                                       \
      demf::set_error("row kernel: %d channels unsupported (64/128/256/512/1024)", (C)); \
      return DEMF_EUNSUPPORTED;                    \
  }

// softmax rows: any S <= 1024, VPL = ceil(S/64) rounded up to a power of two
#define SM_DISPATCH(S, CALL)                       \
  if ((S) <= 0 || (S) > 1024) {                    \
    demf::set_error("softmax: %d keys unsupported (1..1024)", (S)); \
    return DEMF_EUNSUPPORTED;                      \
  } else if ((S) <= 64) { CALL(1); }               \
  else if ((S) <= 128) { CALL(2); }                \
  else if ((S) <= 256) { CALL(4); }                \
  else if ((S) <= 512) { CALL(8); }                \
  else { CALL(16); }

extern "C" int demf_add_dropout_ln_fwd(int R, int C, const float* x, const float* identity,
                                       const float* gamma, const float* beta, float eps, float p,
                                       const void* rng, int op_id, float* s_out, float* y,
                                       float* stats, demf_stream_t stream) {
  // (s_out / stats may be NULL: inference callers - the frozen encoder - keep neither)
  DEMF_REQUIRE(R > 0 && x && gamma && beta && y, "add_dropout_ln_fwd: bad arguments");
  DEMF_REQUIRE(p == 0.f || rng, "add_dropout_ln_fwd: dropout needs the rng state");
#define CALL(V) hipLaunchKernelGGL(add_dropout_ln_fwd_k<V>, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, \
                                   R, x, identity, gamma, beta, eps, p, (const unsigned long long*)rng,        \
                                   (unsigned)op_id, s_out, y, stats)
  LN_DISPATCH(C, CALL)
#undef CALL
  return check_launch("add_dropout_ln_fwd_k");
}

extern "C" int demf_add_dropout_ln_bwd(int R, int C, const float* dy, const float* dy2, const float* s,
                                       const float* stats, const float* gamma, float p, const void* rng,
                                       int op_id, float* ds_out, int ds_accum, float* dx_out,
                                       float* dgamma, float* dbeta, demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && dy && s && stats && gamma && dgamma && dbeta, "add_dropout_ln_bwd: bad arguments");
  DEMF_REQUIRE(p == 0.f || rng, "add_dropout_ln_bwd: dropout needs the rng state");
  // ~256 blocks (one per CU): each wave walks rows_per_wave rows one after the other (two wave
  // reductions per row, ~2 us each), so a column sees <= 256 atomics
  static const int ln_blocks = getenv("DEMF_LN_BWD_BLOCKS") ? max(1, atoi(getenv("DEMF_LN_BWD_BLOCKS"))) : 256;
  const int rpw = max(1, cdiv(R, ln_blocks * 4));
  const int blocks = cdiv(R, 4 * rpw);
#define CALL(V) hipLaunchKernelGGL(add_dropout_ln_bwd_k<V>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,  \
                                   R, rpw, dy, dy2, s, stats, gamma, p, (const unsigned long long*)rng,       \
                                   (unsigned)op_id, ds_out, ds_accum, dx_out, dgamma, dbeta)
  LN_DISPATCH(C, CALL)
#undef CALL
  return check_launch("add_dropout_ln_bwd_k");
}

extern "C" int demf_softmax_dropout_fwd(int R, int S, const float* scores, float p, const void* rng,
                                        int op_id, float* prob, float* out, demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && scores && prob && out, "softmax_dropout_fwd: bad arguments");
  DEMF_REQUIRE(p == 0.f || rng, "softmax_dropout_fwd: dropout needs the rng state");
#define CALL(V) hipLaunchKernelGGL(softmax_dropout_fwd_k<V>, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, \
                                   R, S, scores, p, (const unsigned long long*)rng, (unsigned)op_id, prob, out)
  SM_DISPATCH(S, CALL)
#undef CALL
  return check_launch("softmax_dropout_fwd_k");
}

extern "C" int demf_softmax_dropout_bwd(int R, int S, const float* prob, float p, const void* rng,
                                        int op_id, float* dio, demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && prob && dio, "softmax_dropout_bwd: bad arguments");
  DEMF_REQUIRE(p == 0.f || rng, "softmax_dropout_bwd: dropout needs the rng state");
#define CALL(V) hipLaunchKernelGGL(softmax_dropout_bwd_k<V>, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, \
                                   R, S, prob, p, (const unsigned long long*)rng, (unsigned)op_id, dio)
  SM_DISPATCH(S, CALL)
#undef CALL
  return check_launch("softmax_dropout_bwd_k");
}

extern "C" int demf_msda_prep_fwd(int R, int Q, int H, int L, int P, const float* pts, const float* M,
                                  const float* ab, const float* valid_ratios, const int64_t* shapes,
                                  const float* raw, float* loc, float* w, float* uvw,
                                  demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && Q > 0 && R % Q == 0 && H > 0 && L > 0 && P > 0 && L * P <= 16,
               "msda_prep_fwd: bad sizes (L*P <= 16)");
  DEMF_REQUIRE(pts && M && ab && valid_ratios && shapes && raw && loc && w && uvw, "msda_prep_fwd: null pointer");
  hipLaunchKernelGGL(msda_prep_fwd_k, dim3(cdiv(R * H, 256)), dim3(256), 0, (hipStream_t)stream, R, Q, H, L,
                     P, pts, M, ab, valid_ratios, (const long long*)shapes, raw, loc, w, uvw);
  return check_launch("msda_prep_fwd_k");
}

extern "C" int demf_msda_prep_bwd(int R, int Q, int H, int L, int P, const float* pts, const float* M,
                                  const float* ab, const float* valid_ratios, const int64_t* shapes,
                                  const float* w, const float* uvw, const float* dloc, const float* dloc2,
                                  const float* dw, const float* dw2, float* draw, float* dpts,
                                  demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && Q > 0 && R % Q == 0 && H > 0 && L > 0 && P > 0 && L * P <= 16,
               "msda_prep_bwd: bad sizes (L*P <= 16)");
  DEMF_REQUIRE(pts && M && ab && valid_ratios && shapes && w && uvw && dloc && dw && draw, "msda_prep_bwd: null pointer");
  hipLaunchKernelGGL(msda_prep_bwd_k, dim3(cdiv(R * H, 256)), dim3(256), 0, (hipStream_t)stream, R, Q, H, L,
                     P, pts, M, ab, valid_ratios, (const long long*)shapes, w, uvw, dloc, dloc2, dw, dw2, draw, dpts);
  return check_launch("msda_prep_bwd_k");
}

extern "C" int demf_l2norm_rows_fwd(int R, int C, const float* x, float* y, float* norm,
                                    demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && x && y && norm, "l2norm_rows_fwd: bad arguments");
#define CALL(V) hipLaunchKernelGGL(l2norm_fwd_k<V>, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, R, x, y, norm)
  LN_DISPATCH(C, CALL)
#undef CALL
  return check_launch("l2norm_fwd_k");
}

extern "C" int demf_l2norm_rows_bwd(int R, int C, const float* y, const float* norm, const float* dy,
                                    float* dx, demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && y && norm && dy && dx, "l2norm_rows_bwd: bad arguments");
#define CALL(V) hipLaunchKernelGGL(l2norm_bwd_k<V>, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, R, y, norm, dy, dx)
  LN_DISPATCH(C, CALL)
#undef CALL
  return check_launch("l2norm_bwd_k");
}

extern "C" int demf_vote_combine_fwd(int R, int C, const float* rows, const float* votes,
                                     const float* seed_xyz, float* vote_xyz, float* vote_feats,
                                     float* norm, demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && rows && votes && seed_xyz && vote_xyz && vote_feats && norm,
               "vote_combine_fwd: bad arguments");
#define CALL(V) hipLaunchKernelGGL(vote_combine_fwd_k<V>, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, R, rows, votes, seed_xyz, vote_xyz, vote_feats, norm)
  LN_DISPATCH(C, CALL)
#undef CALL
  return check_launch("vote_combine_fwd_k");
}

extern "C" int demf_vote_combine_bwd(int R, int C, const float* vote_feats, const float* norm,
                                     const float* d_vote_feats, const float* d_vote_xyz,
                                     float* d_votes, float* d_rows, demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && vote_feats && norm && d_votes && d_rows, "vote_combine_bwd: bad arguments");
#define CALL(V) hipLaunchKernelGGL(vote_combine_bwd_k<V>, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, R, vote_feats, norm, d_vote_feats, d_vote_xyz, d_votes, d_rows)
  LN_DISPATCH(C, CALL)
#undef CALL
  return check_launch("vote_combine_bwd_k");
}

extern "C" int demf_rng_advance(void* rng, demf_stream_t stream) {
  DEMF_REQUIRE(rng, "rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_k, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)rng);
  return check_launch("rng_advance_k");
}

extern "C" int demf_rng_next(void* rng, void* snapshot, demf_stream_t stream) {
  DEMF_REQUIRE(rng && snapshot, "rng_next: null state");
  hipLaunchKernelGGL(rng_next_k, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)rng,
                     (unsigned long long*)snapshot);
  return check_launch("rng_next_k");
}

extern "C" int demf_dropout_mask(long long n, float p, const void* rng, int op_id, float* out,
                                 demf_stream_t stream) {
  DEMF_REQUIRE(n > 0 && rng && out && p >= 0.f && p < 1.f, "dropout_mask: bad arguments");
  hipLaunchKernelGGL(dropout_mask_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, p,
                     (const unsigned long long*)rng, (unsigned)op_id, out);
  return check_launch("dropout_mask_k");
}
