// Long-row linear layers of the frozen image stream on gfx950: C = epi((A [+ A2]) . W^T + bias) for
// R ~ 150 000 token rows, K, N in {256, 1024}, fp32 in / out, the products on the bf16 matrix cores.
//
// Reference: the six layers of DeformableDetrEncoder (demf/modeling/layers/deform_detr_encoder.py:68-154, built
// from configs/demf/demf_votenet.py:28-47): per layer sampling_offsets / attention_weights / value_proj /
// output_proj of mmcv's MultiScaleDeformableAttention and the two FFN linears, each followed upstream by its own
// elementwise launches (query + pos, masked_fill, ReLU, dropout(identity in eval) + residual, LayerNorm).  Upstream
// (and this package until round 4) runs them as library GEMMs: 2.0 ms of hipBLASLt fp32 (~60 TF/s) + 1.0 ms of
// elementwise passes per layer at 8 scenes x 18 609 tokens.
//
// Here: one kernel, 128-row x BN-column tiles (BN = 128, or 256 = a whole row when LayerNorm rides in the
// epilogue), 8 waves as 2 (rows) x 4 (columns), reduction in steps of 32 through a double-buffered LDS tile with
// ONE barrier per step.  fp32-grade arithmetic is the three-term split of csrc/mlp.hip (x = h + m + l in bf16,
// the six products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16, fp32 accumulate); the A rows are split once
// per tile on their way into LDS, the weights - frozen - arrive PRE-split as bf16 planes (P, N, K) and are copied
// global -> LDS without touching the VALU.  P = 1 is the bf16 compute mode (operands rounded once).
// Fused: the positional addend on a column range (q = x + pos feeds offsets / weights, x alone feeds the value
// projection: one launch for all three), bias, ReLU, the padding-mask zeroing of value rows, and
// residual + LayerNorm.  Workgroups are dealt so that all column tiles of a row block run on one XCD.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace demf {

using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x2 = float __attribute__((ext_vector_type(2)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));

struct RowsGemmArgs {
  int R, N, K;
  const float* A; long long lda;
  const float* A2; int a2_cols;        // output columns < a2_cols (a multiple of the column tile) are fed by A + A2
  int a2_replace;                      // ... or by A2 INSTEAD of A (a second, pre-added operand: q = x + pos)
  const __bf16* W;                     // (P, N, K) bf16 planes of the (N, K) weight
  const float* bias;                   // (N) or null
  int mode;                            // 0 bias, 1 bias + ReLU, 2 bias + residual + LayerNorm (N == 256)
  const unsigned char* row_mask;       // rows to zero in columns >= mask_col0 (padding mask), or null
  int mask_col0;
  const float* resid; long long ldr;   // mode 2
  const float* gamma; const float* beta; float eps;
  float* C; long long ldc;
  int dbg;   // timing experiments (DEMF_RG_DBG): 1 no MFMA, 2 no B loads, 4 no A loads, 8 no stores, 16 no commit
};

template <int P>
__device__ __forceinline__ void rg_split_pair(float a, float b, unsigned (&o)[P]) {
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
// byte offset of 16-byte chunk `chunk` (8 reduction steps) of the 64-byte row `row` of a plane; chunks are
// XOR-swizzled by bits 2-3 of the row: a fragment read (lane = row, same chunk) touches every bank once
__device__ __forceinline__ int rg_swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int P>
__device__ __forceinline__ void rg_mfma(f32x16& acc, const bf16x8 (&a)[P], const bf16x8 (&b)[P]) {
  if constexpr (P == 3) {   // smallest products first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

constexpr int RG_BM = 128;

// TWO = true (BN = 128): ONE LDS stage (48 KB at P = 3) and one register stage, two barriers per step - so that two
// workgroups share a CU and one's MFMA phase runs under the other's load / split phase.
template <int P, int BN, bool TWO>
__global__ __launch_bounds__(512, TWO ? 4 : 2) void rows_gemm_kernel(RowsGemmArgs p) {
  constexpr int NT = BN / 128;                  // 32-column tiles per wave
  constexpr int A_BYTES = RG_BM * 64, B_BYTES = BN * 64;
  constexpr int STAGE = P * (A_BYTES + B_BYTES);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 31, lh = lane >> 5, wm = wave & 1, wn = wave >> 1;
  // row block / column tile of this workgroup: linear id L -> row block (L mod 8) + 8 (L / (8 gx)), column tile
  // (L / 8) mod gx - all column tiles of a row block on ONE XCD (its A rows are fetched into one L2)
  const int gx = p.N / BN;
  const int L = blockIdx.x;
  const int by = (L & 7) + 8 * (L / (8 * gx)), bx = (L >> 3) % gx;
  const int m0 = by * RG_BM, n0 = bx * BN;
  if (m0 >= p.R) return;
  const bool alt = p.A2 != nullptr && n0 < p.a2_cols;
  const bool add2 = alt && !p.a2_replace;
  const float* __restrict__ Ap = (alt && p.a2_replace) ? p.A2 : p.A;
  // staging roles: A - thread t moves float4 kq = t & 7 of rows (t >> 3) and (t >> 3) + 64;
  //                B - 16-byte chunk (t & 3) of column (t >> 2) [+ 128] of every plane
  const int ar = t >> 3, akq = t & 7, bc = t >> 2, bch = t & 3;
  // two register stages: the loads of steps s + 1 and s + 2 are in flight while step s is contracted (one
  // workgroup per CU: nothing else hides a ~1 us first-touch load behind a ~0.6 us MFMA phase)
  float4 ra[2][2], ra2[2][2];
  u32x4 rb[2][P * NT];
  auto fetch = [&](auto stage, int k0) {
    constexpr int G = decltype(stage)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + ar + 64 * i;
      ra[G][i] = ra2[G][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < p.R && !(p.dbg & 4)) {
        ra[G][i] = *reinterpret_cast<const float4*>(Ap + (size_t)row * p.lda + k0 + 4 * akq);
        if (add2) ra2[G][i] = *reinterpret_cast<const float4*>(p.A2 + (size_t)row * p.lda + k0 + 4 * akq);
      }
    }
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (!(p.dbg & 2)) rb[G][q * NT + j] = *reinterpret_cast<const u32x4*>(p.W + ((size_t)q * p.N + n0 + bc + 128 * j) * p.K + k0 + 8 * bch);
  };
  auto commit = [&](auto stage, int buf) {
    constexpr int G = decltype(stage)::value;
    char* sa = smem + buf * STAGE;
    char* sb = sa + P * A_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v = ra[G][i];
      if (add2) { v.x += ra2[G][i].x; v.y += ra2[G][i].y; v.z += ra2[G][i].z; v.w += ra2[G][i].w; }
      unsigned lo[P], hi[P];
      rg_split_pair<P>(v.x, v.y, lo);
      rg_split_pair<P>(v.z, v.w, hi);
      const int off = rg_swz(ar + 64 * i, akq >> 1) + 8 * (akq & 1);
#pragma unroll
      for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(sa + q * A_BYTES + off) = make_uint2(lo[q], hi[q]);
    }
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<u32x4*>(sb + q * B_BYTES + rg_swz(bc + 128 * j, bch)) = rb[G][q * NT + j];
  };
  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // Step s contracts LDS buffer s & 1 while the SAME waves split and store the rows of step s + 1 into the other
  // buffer (independent instruction streams in one block: the scheduler threads the VALU / ds_write work between
  // the MFMAs, whose pipe runs on its own) and the loads of step s + 2 are issued; one barrier per step.  With the
  // split in front of the barrier instead (a first version), every wave of the CU was in its VALU phase at the
  // same time and the matrix pipe idled for it: 27 % of the mode's MFMA rate.
  auto step = [&](auto cur, int k0) {
    constexpr int CUR = decltype(cur)::value;
    using Nxt = std::integral_constant<int, CUR ^ 1>;
    if (k0 + 64 < p.K) fetch(cur, k0 + 64);
    const char* sa = smem + CUR * STAGE;
    const char* sb = sa + P * A_BYTES;
    if (k0 + 32 < p.K && !(p.dbg & 16)) commit(Nxt{}, CUR ^ 1);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 pa[2][P], pb[NT][P];
#pragma unroll
      for (int q = 0; q < P; ++q) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          pa[i][q] = *reinterpret_cast<const bf16x8*>(sa + q * A_BYTES + rg_swz(64 * wm + 32 * i + lc, 2 * c + lh));
#pragma unroll
        for (int j = 0; j < NT; ++j)
          pb[j][q] = *reinterpret_cast<const bf16x8*>(sb + q * B_BYTES + rg_swz(32 * NT * wn + 32 * j + lc, 2 * c + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (!(p.dbg & 1)) rg_mfma<P>(acc[i][j], pa[i], pb[j]);
    }
    lds_barrier();
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if constexpr (TWO) {
    fetch(S0{}, 0);
    for (int k0 = 0; k0 < p.K; k0 += 32) {
      commit(S0{}, 0);
      lds_barrier();
      if (k0 + 32 < p.K) fetch(S0{}, k0 + 32);
      const char* sa = smem;
      const char* sb = sa + P * A_BYTES;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 pa[2][P], pb[NT][P];
#pragma unroll
        for (int q = 0; q < P; ++q) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
            pa[i][q] = *reinterpret_cast<const bf16x8*>(sa + q * A_BYTES + rg_swz(64 * wm + 32 * i + lc, 2 * c + lh));
#pragma unroll
          for (int j = 0; j < NT; ++j)
            pb[j][q] = *reinterpret_cast<const bf16x8*>(sb + q * B_BYTES + rg_swz(32 * NT * wn + 32 * j + lc, 2 * c + lh));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) rg_mfma<P>(acc[i][j], pa[i], pb[j]);
      }
      lds_barrier();
    }
  } else {
    fetch(S0{}, 0);
    if (32 < p.K) fetch(S1{}, 32);
    commit(S0{}, 0);
    lds_barrier();
    for (int k0 = 0; k0 < p.K; k0 += 64) {
      step(S0{}, k0);
      if (k0 + 32 < p.K) step(S1{}, k0 + 32);
    }
  }
  // accumulator register r of tile (i, j) = row 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 lh, column 32 NT wn + 32 j + lc
  if (p.mode != 2) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + 32 * NT * wn + 32 * j + lc;
      const float bias = p.bias != nullptr ? p.bias[col] : 0.f;
      const bool masked = p.row_mask != nullptr && col >= p.mask_col0;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < p.R && !(p.dbg & 8)) {
            float v = acc[i][j][r] + bias;
            if (p.mode == 1) v = fmaxf(v, 0.f);
            if (masked && p.row_mask[row]) v = 0.f;
            p.C[(size_t)row * p.ldc + col] = v;
          }
        }
    }
    return;
  }
  if constexpr (BN == 256) {
    // residual + LayerNorm: the tile is 128 whole rows.  Sums through an fp32 LDS tile (the stage buffers are
    // free after the last step), then one wave per row: 4 values per lane, two butterfly reductions.
    constexpr int LD = 260;
    float* tile = reinterpret_cast<float*>(smem);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = 32 * NT * wn + 32 * j + lc;
      const float bias = p.bias != nullptr ? p.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
          tile[rl * LD + col] = acc[i][j][r] + bias;
        }
    }
    __syncthreads();
    const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + 4 * lane);
    const float4 b4 = *reinterpret_cast<const float4*>(p.beta + 4 * lane);
    for (int rr = 0; rr < 16; ++rr) {
      const int rl = 16 * wave + rr, row = m0 + rl;
      if (row >= p.R) break;                       // (wave-uniform)
      float4 v = *reinterpret_cast<const float4*>(tile + rl * LD + 4 * lane);
      const float4 x = *reinterpret_cast<const float4*>(p.resid + (size_t)row * p.ldr + 4 * lane);
      v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
      float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float mean = s * (1.f / 256.f);
      const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
      float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = 1.f / sqrtf(q * (1.f / 256.f) + p.eps);
      float4 y;
      y.x = dx * rstd * g4.x + b4.x; y.y = dy * rstd * g4.y + b4.y;
      y.z = dz * rstd * g4.z + b4.z; y.w = dw * rstd * g4.w + b4.w;
      *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + 4 * lane) = y;
    }
  }
}

// Four-wave form of the 128 x 128 tile: 2 x 2 waves of 64 x 64 (four accumulator tiles per wave), one LDS stage
// (48 KB at P = 3), three workgroups per CU.  Per MFMA a wave reads 0.5 KB of fragments instead of 0.75 KB (the
// 8-wave form's 64 x 32 wave tiles put LDS reads + stores of two co-resident workgroups at ~3 000 cycles per step
// pair, level with the MFMA time), and three workgroups rotate through load / split / MFMA instead of two.
template <int P, bool ADD2>
__global__ __launch_bounds__(256, (ADD2 && P == 3) ? 2 : 3) void rows_gemm4_kernel(RowsGemmArgs p) {
  constexpr int BN = 128;
  constexpr int A_BYTES = RG_BM * 64, B_BYTES = BN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 31, lh = lane >> 5, wm = wave & 1, wn = wave >> 1;
  const int gx = p.N / BN;
  const int L = blockIdx.x;
  const int by = (L & 7) + 8 * (L / (8 * gx)), bx = (L >> 3) % gx;
  const int m0 = by * RG_BM, n0 = bx * BN;
  if (m0 >= p.R) return;
  const bool alt = p.A2 != nullptr && n0 < p.a2_cols;
  const bool add2 = ADD2 && alt && !p.a2_replace;
  const float* __restrict__ Ap = (alt && p.a2_replace) ? p.A2 : p.A;
  // staging: A - float4 kq = t & 7 of rows (t >> 3) + 32 i; B - 16-byte chunk (t & 3) of columns (t >> 2) + 64 j
  const int ar = t >> 3, akq = t & 7, bc = t >> 2, bch = t & 3;
  float4 ra[4], ra2[ADD2 ? 4 : 1];
  u32x4 rb[P * 2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + ar + 32 * i;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (ADD2) ra2[i] = ra[i];
      if (row < p.R) {
        ra[i] = *reinterpret_cast<const float4*>(Ap + (size_t)row * p.lda + k0 + 4 * akq);
        if constexpr (ADD2)
          if (add2) ra2[i] = *reinterpret_cast<const float4*>(p.A2 + (size_t)row * p.lda + k0 + 4 * akq);
      }
    }
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        rb[q * 2 + j] = *reinterpret_cast<const u32x4*>(p.W + ((size_t)q * p.N + n0 + bc + 64 * j) * p.K + k0 + 8 * bch);
  };
  auto commit = [&]() {
    char* sa = smem;
    char* sb = sa + P * A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = ra[i];
      if constexpr (ADD2) { v.x += ra2[i].x; v.y += ra2[i].y; v.z += ra2[i].z; v.w += ra2[i].w; }   // (zeros if !add2)
      unsigned lo[P], hi[P];
      rg_split_pair<P>(v.x, v.y, lo);
      rg_split_pair<P>(v.z, v.w, hi);
      const int off = rg_swz(ar + 32 * i, akq >> 1) + 8 * (akq & 1);
#pragma unroll
      for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(sa + q * A_BYTES + off) = make_uint2(lo[q], hi[q]);
    }
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<u32x4*>(sb + q * B_BYTES + rg_swz(bc + 64 * j, bch)) = rb[q * 2 + j];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  fetch(0);
  for (int k0 = 0; k0 < p.K; k0 += 32) {
    commit();
    lds_barrier();
    if (k0 + 32 < p.K) fetch(k0 + 32);
    const char* sa = smem;
    const char* sb = sa + P * A_BYTES;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 pa[2][P];
#pragma unroll
      for (int q = 0; q < P; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          pa[i][q] = *reinterpret_cast<const bf16x8*>(sa + q * A_BYTES + rg_swz(64 * wm + 32 * i + lc, 2 * c + lh));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8 pb[P];
#pragma unroll
        for (int q = 0; q < P; ++q)
          pb[q] = *reinterpret_cast<const bf16x8*>(sb + q * B_BYTES + rg_swz(64 * wn + 32 * j + lc, 2 * c + lh));
#pragma unroll
        for (int i = 0; i < 2; ++i) rg_mfma<P>(acc[i][j], pa[i], pb);
      }
    }
    lds_barrier();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 64 * wn + 32 * j + lc;
    const float bias = p.bias != nullptr ? p.bias[col] : 0.f;
    const bool masked = p.row_mask != nullptr && col >= p.mask_col0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.R) {
          float v = acc[i][j][r] + bias;
          if (p.mode == 1) v = fmaxf(v, 0.f);
          if (masked && p.row_mask[row]) v = 0.f;
          p.C[(size_t)row * p.ldc + col] = v;
        }
      }
  }
}

template <int P, bool ADD2>
static int rows_gemm4_launch(const RowsGemmArgs& a, hipStream_t s) {
  constexpr int lds = P * (RG_BM * 64 + 128 * 64);
  const int gx = a.N / 128, gy = (cdiv(a.R, RG_BM) + 7) / 8 * 8;
  hipLaunchKernelGGL((rows_gemm4_kernel<P, ADD2>), dim3(gx * gy), dim3(256), lds, s, a);
  return check_launch("rows_gemm4_kernel");
}

// LayerNorm form with TWO workgroups per CU: 64 rows x 256 columns per workgroup (8 waves, each 64 rows x 32
// columns), one LDS stage (12 + 48 KB at P = 3; the fp32 LayerNorm tile of 64 x 260 floats reuses it).  The
// 128-row form above holds 64 accumulator registers per lane and runs alone on its CU in lockstep phases (§3.9c).
template <int P>
__global__ __launch_bounds__(512, 4) void rows_gemm_ln64_kernel(RowsGemmArgs p) {
  constexpr int BM = 64, BN = 256;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM;
  if (m0 >= p.R) return;
  // staging: A - float4 kq = t & 7 of row t >> 3; B - 16-byte chunk (t & 3) of columns (t >> 2) + 128 j
  const int ar = t >> 3, akq = t & 7, bc = t >> 2, bch = t & 3;
  float4 ra;
  u32x4 rb[P * 2];
  auto fetch = [&](int k0) {
    const int row = m0 + ar;
    ra = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < p.R) ra = *reinterpret_cast<const float4*>(p.A + (size_t)row * p.lda + k0 + 4 * akq);
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        rb[q * 2 + j] = *reinterpret_cast<const u32x4*>(p.W + ((size_t)q * p.N + bc + 128 * j) * p.K + k0 + 8 * bch);
  };
  auto commit = [&]() {
    char* sa = smem;
    char* sb = sa + P * A_BYTES;
    unsigned lo[P], hi[P];
    rg_split_pair<P>(ra.x, ra.y, lo);
    rg_split_pair<P>(ra.z, ra.w, hi);
    const int off = rg_swz(ar, akq >> 1) + 8 * (akq & 1);
#pragma unroll
    for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(sa + q * A_BYTES + off) = make_uint2(lo[q], hi[q]);
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<u32x4*>(sb + q * B_BYTES + rg_swz(bc + 128 * j, bch)) = rb[q * 2 + j];
  };
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  fetch(0);
  for (int k0 = 0; k0 < p.K; k0 += 32) {
    commit();
    lds_barrier();
    if (k0 + 32 < p.K) fetch(k0 + 32);
    const char* sa = smem;
    const char* sb = sa + P * A_BYTES;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 pa[2][P], pb[P];
#pragma unroll
      for (int q = 0; q < P; ++q) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          pa[i][q] = *reinterpret_cast<const bf16x8*>(sa + q * A_BYTES + rg_swz(32 * i + lc, 2 * c + lh));
        pb[q] = *reinterpret_cast<const bf16x8*>(sb + q * B_BYTES + rg_swz(32 * wave + lc, 2 * c + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) rg_mfma<P>(acc[i], pa[i], pb);
    }
    lds_barrier();
  }
  constexpr int LD = 260;
  float* tile = reinterpret_cast<float*>(smem);
  {
    const int col = 32 * wave + lc;
    const float bias = p.bias != nullptr ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh) * LD + col] = acc[i][r] + bias;
  }
  __syncthreads();
  const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + 4 * lane);
  const float4 b4 = *reinterpret_cast<const float4*>(p.beta + 4 * lane);
  for (int rr = 0; rr < 8; ++rr) {
    const int rl = 8 * wave + rr, row = m0 + rl;
    if (row >= p.R) break;                       // (wave-uniform)
    float4 v = *reinterpret_cast<const float4*>(tile + rl * LD + 4 * lane);
    const float4 x = *reinterpret_cast<const float4*>(p.resid + (size_t)row * p.ldr + 4 * lane);
    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    float sum = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.f / 256.f);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q * (1.f / 256.f) + p.eps);
    float4 y;
    y.x = dx * rstd * g4.x + b4.x; y.y = dy * rstd * g4.y + b4.y;
    y.z = dz * rstd * g4.z + b4.z; y.w = dw * rstd * g4.w + b4.w;
    *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + 4 * lane) = y;
  }
}

template <int P>
static int rows_gemm_ln64_launch(const RowsGemmArgs& a, hipStream_t s) {
  constexpr int stage = P * (64 * 64 + 256 * 64), ln = 64 * 260 * 4;
  constexpr int lds = stage > ln ? stage : ln;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_gemm_ln64_kernel<P>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_error("rows_gemm: cannot reserve %d bytes of LDS", lds);
      return DEMF_ELAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((rows_gemm_ln64_kernel<P>), dim3(cdiv(a.R, 64)), dim3(512), lds, s, a);
  return check_launch("rows_gemm_ln64_kernel");
}

template <int P, int BN, bool TWO>
static int rows_gemm_launch(const RowsGemmArgs& a, hipStream_t s) {
  constexpr int bytes = (TWO ? 1 : 2) * P * (RG_BM * 64 + BN * 64);
  constexpr int ln_bytes = BN == 256 ? 128 * 260 * 4 : 0;
  constexpr int lds = bytes > ln_bytes ? bytes : ln_bytes;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_gemm_kernel<P, BN, TWO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_error("rows_gemm: cannot reserve %d bytes of LDS", lds);
      return DEMF_ELAUNCH;
    }
    attr_done = true;
  }
  const int gx = a.N / BN, gy = (cdiv(a.R, RG_BM) + 7) / 8 * 8;
  hipLaunchKernelGGL((rows_gemm_kernel<P, BN, TWO>), dim3(gx * gy), dim3(512), lds, s, a);
  return check_launch("rows_gemm_kernel");
}

}  // namespace demf

using namespace demf;

extern "C" int demf_rows_gemm_f32(int R, int N, int K, const float* A, long long lda, const float* A2, int a2_cols,
                                  int a2_op, const void* w_planes, int planes, const float* bias, int mode,
                                  const unsigned char* row_mask, int mask_col0, const float* resid, long long ldr,
                                  const float* gamma, const float* beta, float eps, float* C, long long ldc,
                                  demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && N > 0 && K > 0 && A != nullptr && w_planes != nullptr && C != nullptr, "rows_gemm: bad arguments");
  DEMF_REQUIRE(planes == 1 || planes == 3, "rows_gemm: planes must be 1 (bf16) or 3 (fp32 as three bf16 terms)");
  DEMF_REQUIRE(K % 32 == 0 && lda % 4 == 0, "rows_gemm: K %% 32 and lda %% 4 required (K = %d, lda = %lld)", K, lda);
  DEMF_REQUIRE(mode >= 0 && mode <= 2, "rows_gemm: mode %d", mode);
  RowsGemmArgs a{};
  DEMF_REQUIRE(a2_op == 0 || a2_op == 1, "rows_gemm: a2_op %d", a2_op);
  a.R = R; a.N = N; a.K = K; a.A = A; a.lda = lda; a.A2 = A2; a.a2_cols = a2_cols; a.a2_replace = a2_op;
  a.W = reinterpret_cast<const __bf16*>(w_planes); a.bias = bias; a.mode = mode;
  a.row_mask = row_mask; a.mask_col0 = mask_col0; a.resid = resid; a.ldr = ldr;
  a.gamma = gamma; a.beta = beta; a.eps = eps; a.C = C; a.ldc = ldc;
  static const int rg_dbg = getenv("DEMF_RG_DBG") ? atoi(getenv("DEMF_RG_DBG")) : 0;    // phase-skip timing experiments
  a.dbg = rg_dbg;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 2) {
    DEMF_REQUIRE(N == 256 && resid != nullptr && gamma != nullptr && beta != nullptr && ldr % 4 == 0 && ldc % 4 == 0,
                 "rows_gemm: the LayerNorm epilogue needs N == 256, a residual and gamma / beta");
    static const int ln128 = getenv("DEMF_RG_LN128") ? atoi(getenv("DEMF_RG_LN128")) : 0;   // A/B: the 128-row form
    if (!ln128 && A2 == nullptr) return planes == 3 ? rows_gemm_ln64_launch<3>(a, s) : rows_gemm_ln64_launch<1>(a, s);
    return planes == 3 ? rows_gemm_launch<3, 256, false>(a, s) : rows_gemm_launch<1, 256, false>(a, s);
  }
  DEMF_REQUIRE(N % 128 == 0, "rows_gemm: N %% 128 required (N = %d)", N);
  DEMF_REQUIRE(A2 == nullptr || a2_cols % 128 == 0 || a2_cols >= N, "rows_gemm: a2_cols must be a multiple of 128");
  static const int w8 = getenv("DEMF_RG_WAVES8") ? atoi(getenv("DEMF_RG_WAVES8")) : 0;    // A/B: the 8-wave forms
  if (!w8) {
    // (the adding form needs 16 more staging registers: 116 bytes of scratch at three planes, 631 vs 400 us on the
    // encoder's input projection - it stays on the 8-wave form; callers with a pre-added operand use a2_op = 1)
    if (A2 == nullptr || a2_op == 1)
      return planes == 3 ? rows_gemm4_launch<3, false>(a, s) : rows_gemm4_launch<1, false>(a, s);
    static const int add4 = getenv("DEMF_RG_ADD4") ? atoi(getenv("DEMF_RG_ADD4")) : 0;      // A/B: adding form on 4 waves
    if (planes == 1) return rows_gemm4_launch<1, true>(a, s);
    if (add4) return rows_gemm4_launch<3, true>(a, s);
  }
  static const bool one = getenv("DEMF_RG_ONE") && atoi(getenv("DEMF_RG_ONE"));     // A/B: one workgroup per CU
  if (one) return planes == 3 ? rows_gemm_launch<3, 128, false>(a, s) : rows_gemm_launch<1, 128, false>(a, s);
  return planes == 3 ? rows_gemm_launch<3, 128, true>(a, s) : rows_gemm_launch<1, 128, true>(a, s);
}

namespace demf {
// y = LayerNorm(resid + x) * gamma + beta over rows of 256 channels, and ypos = y + pos (the next layer's query):
// one wave per row, 4 values per lane.  The encoder's FFN tail (identity + dropout(.) in eval, nn.LayerNorm) and
// the `query + query_pos` of the NEXT layer's attention in one pass.
__global__ __launch_bounds__(256) void rows_ln_pos_kernel(int R, const float* __restrict__ x, const float* __restrict__ resid,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, const float* __restrict__ pos,
                                                          float* __restrict__ y, float* __restrict__ ypos) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const size_t o = (size_t)row * 256 + 4 * lane;
  float4 v = *reinterpret_cast<const float4*>(x + o);
  if (resid != nullptr) {
    const float4 r = *reinterpret_cast<const float4*>(resid + o);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  float4 out = v;
  if (gamma != nullptr) {
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
    const float mean = s * (1.f / 256.f);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) q += __shfl_xor(q, k);
    const float rstd = 1.f / sqrtf(q * (1.f / 256.f) + eps);
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + 4 * lane);
    const float4 b4 = *reinterpret_cast<const float4*>(beta + 4 * lane);
    out.x = dx * rstd * g4.x + b4.x; out.y = dy * rstd * g4.y + b4.y;
    out.z = dz * rstd * g4.z + b4.z; out.w = dw * rstd * g4.w + b4.w;
  }
  if (y != nullptr) *reinterpret_cast<float4*>(y + o) = out;
  if (ypos != nullptr) {
    const float4 p4 = *reinterpret_cast<const float4*>(pos + o);
    out.x += p4.x; out.y += p4.y; out.z += p4.z; out.w += p4.w;
    *reinterpret_cast<float4*>(ypos + o) = out;
  }
}
}  // namespace demf

extern "C" int demf_rows_ln_pos_f32(int R, int C, const float* x, const float* resid, const float* gamma,
                                    const float* beta, float eps, const float* pos, float* y, float* ypos,
                                    demf_stream_t stream) {
  DEMF_REQUIRE(R > 0 && C == 256 && x != nullptr && (y != nullptr || ypos != nullptr) && (ypos == nullptr || pos != nullptr) &&
                   (gamma == nullptr) == (beta == nullptr),
               "rows_ln_pos: bad arguments (C must be 256)");
  hipLaunchKernelGGL(demf::rows_ln_pos_kernel, dim3(demf::cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, R, x, resid,
                     gamma, beta, eps, pos, y, ypos);
  return demf::check_launch("rows_ln_pos_kernel");
}
