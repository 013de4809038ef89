// Fused backward of one shared-MLP layer for gfx950: input gradient, weight gradient and the
// BatchNorm-backward sums of the layer below from ONE pass over the layer's saved output.
//
// Reference: the backward of Conv2d(1x1) -> BatchNorm2d(train) -> ReLU inside mmdet3d's PointSAModule
// stacks (built by the reference at demf/modeling/heads/class_agnostic_vote_head.py:383 and by the
// backbone config configs/demf/demf_votenet.py:48-62), which autograd runs as separate passes.
//
// Until round 2 a layer's backward was two launches here - demf_mlp_gemm_bwd_dx_red (dX = dY.W + the
// sums of layer l-1) and demf_mlp_gemm_bwd_dw (dW = dY^T.act(Y_{l-1})) - each re-reading Y_l (and the
// upstream gradient) and each re-doing the BN-backward transform dY = gi*dZ + a*y + b, and in the
// fp32-grade mode (three bf16 terms per operand, csrc/mlp.hip) each splitting its operands per USE.
// Here one workgroup of 4 or 8 waves walks 32-row slabs and
//   * builds the dY tile ONCE: every wave owns one 32-channel slice of it (32 rows x 32 channels),
//     loads it as 4-row x 4-channel register patches (full 128-byte lines), applies the BN-backward
//     transform, splits it ONCE into bf16 planes h|m|l and leaves those planes in LDS in BOTH
//     orientations - [row][channel] for the dX contraction (reduction over channels) and
//     [channel][row] for the dW contraction (reduction over rows) - 16-byte chunks XOR-swizzled so
//     that the MFMA fragment reads (ds_read_b128, 8 consecutive rows) are conflict-free without padding;
//   * keeps the layer weight W as MFMA B fragments in REGISTERS for the whole launch (split once at
//     kernel start), so dX = dY.W needs no weight staging at all; a wave contracts one K tile over a channel
//     half and the two partial tiles of a K tile meet in an fp32 LDS tile in two ordered rounds of
//     plain store / read-add-write on disjoint row chunks (LDS float atomics were measured at ~170
//     cycles per wave instruction on gfx950: 1.15 ms of a 1.6 ms launch - not an option);
//   * stages act(Y_{l-1}) - BN + ReLU of the layer below, split once - as [k][row] planes shared by
//     the waves of a row group, the B operand of dW += dY^T.act(Y_{l-1}); dW tiles stay in
//     accumulators for the whole row range of the workgroup and are flushed once;
//   * finishes dX out of the LDS tile with coalesced float4 rows: either stores it and takes the BN-
//     backward sums of layer l-1 on the way (RED), or - first layer of SA1, 4-float input rows, no input
//     gradient - never stores it and takes the raw sums of layer 0's whole backward instead (FIRST, see
//     mlp_first_finish_k in csrc/mlp.hip).
// Algorithmic HBM bytes per row: 4*(N [Y_l] + N [G, dense only] + K [Y_{l-1}] + K [dX, RED only]).
#include "common.h"
#include "bn_fin.h"

namespace demf {

using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));
using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));

struct FusedBwdArgs {
  int R, N, K, ns;
  const float* Yl;    // (R x N) pre-BN output of this layer
  const float* G;     // dense upstream gradient (R x N), or null
  const float* dP;    // sparse: pooled gradient (R/ns x N)
  const int* arg;     //         arg-max slot   (R/ns x N)
  const float* vec;   // 5 x N backward vectors of this layer (scale, shift, gi, a, b)
  const float* Xp;    // (R x K) pre-BN output of layer l-1
  const float* pss;   // [scale|shift] of layer l-1 (2K)
  const float* pmi;   // [mean|invstd] of layer l-1 (2K)
  const float* W;     // (N x K) weight of this layer
  float* dX;          // (R x K) gradient of layer l-1's activation (RED), unused for FIRST
  float* dW;          // (N x K) accumulated (arrives zeroed)
  double* g12;        // RED: sum dZ | sum dZ*xhat of layer l-1 (2K), accumulated
  const float* fX;    // FIRST: (R x 4) input rows of layer 0
  double* fsum;       // FIRST: g1(K) | g2(K) | P(K x 4) | Q(K x 4) | cx(4), accumulated
  BnVecFin vfin;      // RED: layer l-1's backward vectors by the last workgroup (ticket != null)
  // A launch may cover K of the fld >= K channels of layer l-1 (columns fc0 .. fc0 + K): Xp, dX, W, dW
  // then arrive offset by fc0 with row strides ldk / ldw; pss, pmi, g12 and vfin are the WHOLE layer's.
  int ldk, ldw, fld, fc0;
  const float* W0;    // FIRST with ST bit 2: (K x 4) weight of layer 0 - Y_0 = fX.W0^T is recomputed, Xp is not read
  int dbg;            // DEMF_BWD_DBG (timing experiments, results wrong): bit 0 = the dW flush is skipped
};

// one fp32 value -> P bf16 planes: P = 1: rounded; P = 3: x = h + m + l exactly (csrc/mlp.hip, mode 2)
template <int P>
__device__ __forceinline__ void split_planes(float x, __bf16 (&o)[P]) {
  if constexpr (P == 2) {       // two fp16 terms (csrc/common.h): the planes carry fp16 bit patterns
    const _Float16 h = (_Float16)x;
    o[0] = __builtin_bit_cast(__bf16, h);
    o[1] = __builtin_bit_cast(__bf16, (_Float16)(x - (float)h));
    return;
  }
  o[0] = (__bf16)x;
  if constexpr (P == 3) {
    const float r = x - (float)o[0];
    o[1] = (__bf16)r;
    o[2] = (__bf16)(r - (float)o[1]);
  }
}

// Two fp32 values -> P packed bf16 pairs (low half = a, high half = b): one v_cvt_pk_bf16_f32 per plane,
// the differences on both elements at once.  Pairs are formed along ROWS (same channel, rows j / j+1):
// the [channel][row] planes take them as they are, the [row][channel] planes re-pair two channels with
// one v_perm_b32.
using f32x2 = float __attribute__((ext_vector_type(2)));
template <int P>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&o)[P]) {
  if constexpr (P == 2) {
    split2_f16(a, b, o[0], o[1]);
    return;
  }
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
// (lo16(p0) | lo16(p1) << 16) and (hi16(p0) | hi16(p1) << 16)
__device__ __forceinline__ unsigned pair_lo(unsigned p0, unsigned p1) { return __builtin_amdgcn_perm(p1, p0, 0x05040100u); }
__device__ __forceinline__ unsigned pair_hi(unsigned p0, unsigned p1) { return __builtin_amdgcn_perm(p1, p0, 0x07060302u); }

// byte offset of 16-byte chunk `chunk` (0..3) of 64-byte row `row`.  Chunks are XOR-swizzled by bits 2-3
// of the row: a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32),
// and within each group the four rows that share row & 3 (= the same 16-dword bank window) differ in
// (row >> 2) & 3, so a fragment read (lane = row, same chunk index) touches all 64 banks exactly once.
// (Measured with the earlier (row >> 1) & 3: SQ_LDS_BANK_CONFLICT = half of all LDS cycles.)
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int P>
__device__ __forceinline__ void mfma_planes(f32x16& acc, const bf16x8 (&a)[P], const bf16x8 (&b)[P]) {
  if constexpr (P == 2) {   // fp16 terms: l.h', h.l', h.h'
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a[1]), __builtin_bit_cast(f16x8_v, b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a[0]), __builtin_bit_cast(f16x8_v, b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a[0]), __builtin_bit_cast(f16x8_v, b[0]), acc, 0, 0, 0);
  } else if constexpr (P == 3) {   // the six products of weight >= 2^-16, smallest first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  }
}

// fp16 terms: the power of two that brings a slab's largest |dY| = m into [8, 16) - 2^11 of head-room below fp16's
// 65504 for the slabs that follow under the same (lagging) scale, and everything down to 2^-4 of m above the
// 2^-1 where the two-term split starts to lose bits against m itself.
__device__ __forceinline__ float f16_scale_for(float m) {
  int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xff) - 127;      // floor(log2 m), m > 0
  e = 3 - e;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}

// CW (2 or 4) consecutive elements of a row stored as fp32 or bf16 -> fp32 vector
template <int CW, bool B16>
__device__ __forceinline__ auto load_row(const float* __restrict__ base, size_t elem) {
  using vecT = float __attribute__((ext_vector_type(CW)));
  if constexpr (!B16) {
    return *reinterpret_cast<const vecT*>(base + elem);
  } else {
    const __bf16* b = reinterpret_cast<const __bf16*>(base) + elem;
    vecT v;
    if constexpr (CW == 4) {
      const uint2 u = *reinterpret_cast<const uint2*>(b);
      v[0] = __builtin_bit_cast(float, u.x << 16); v[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
      v[2] = __builtin_bit_cast(float, u.y << 16); v[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
    } else {
      const unsigned u = *reinterpret_cast<const unsigned*>(b);
      v[0] = __builtin_bit_cast(float, u << 16); v[1] = __builtin_bit_cast(float, u & 0xffff0000u);
    }
    return v;
  }
}

// NTN = N/32 channel slices, KT = K/32, KG = waves that share one channel slice (they split the K tiles,
// KTW each, and the rows of the transform).  A workgroup = NTN*KG waves on one 32-row slab at a time;
// 4-wave workgroups run two per CU.  Wave w -> nt = w % NTN, kg = w / NTN.
// ST (bf16 compute mode, BASELINE configs[3]): rows stored as bf16 in HBM - bit 0: Y_{l-1} (Xp), bit 1:
// Y_l, the dense upstream gradient G and the input gradient dX this launch writes.
template <int NTN, int KT, int KG, bool SPARSE, int CM, int EPI, int ST = 0>   // CM 1 bf16 / 2 three-term ; EPI 0 RED / 1 FIRST
__global__ __launch_bounds__(64 * NTN * KG, 512 / (64 * NTN * KG)) void mlp_bwd_fused_kernel(FusedBwdArgs p) {
  constexpr bool XB = (ST & 1) != 0, YB = (ST & 2) != 0;
  // ST bit 2 (FIRST only): layer 0's raw output is not stored - its 4-float input rows fX and its (K x 4) weight
  // W0 rebuild the values this kernel needs (two channels of four rows per lane for the staged activation,
  // four channels of a row in the epilogue): 268 MB less to read at SA1.  bf16 mode: products of bf16-rounded
  // operands, as the forward kernels compute them.
  constexpr bool XR = (ST & 4) != 0;
  static_assert(!XR || (EPI == 1 && !XB), "recomputed rows: FIRST epilogue, no storage type");
  constexpr int P = CM == 2 ? 3 : (CM == 3 ? 2 : 1);
  // CM 3: fp32 results from two fp16 terms per operand (csrc/common.h).  Activations and weights are split as they
  // are; the gradient operand dY is first multiplied by a power of two, per workgroup and slab: the scale follows the
  // largest |dY| this workgroup has met so far, this slab included ([A] below); dX is scaled back as it leaves the
  // LDS tile, the dW accumulators when the scale changes and at the flush.
  constexpr bool H2 = CM == 3;
  constexpr int N = NTN * 32, K = KT * 32;
  constexpr int NW = NTN * KG;              // waves per workgroup
  constexpr int NT = 64 * NW;               // threads
  constexpr int RS = 32;                    // rows per slab
  constexpr int KTW = KT / KG;              // K tiles per wave
  constexpr int CW = 4 / KG;                // channels per lane patch of the dY transform (4 rows x CW)
  constexpr int LPB = 32 / CW;              // lanes per 4-row block of the transform
  constexpr int QK = K / 4;                 // float4 per dX row
  constexpr int ER = NT / QK;               // rows per epilogue pass (2 passes per slab)
  constexpr int NH = NW / KT;               // dX: waves per K tile = partial tiles to fold (channel halves)
  constexpr int SL = NTN / NH;              // dX: channel slices a wave contracts
  constexpr int CR = 16 / NH;               // accumulator registers per row chunk of the ordered rounds
  static_assert(NW % KT == 0 && NH >= 1 && NTN % NH == 0 && SL >= 1, "dX wave map");
  static_assert(KG >= 1 && KTW >= 1 && KT % KG == 0 && (NW == 4 || NW == 8), "unsupported shape");
  static_assert(2 * ER == RS && 4 * (K / 64) == NW, "epilogue / staging maps");
  static_assert(EPI == 0 || (!SPARSE), "FIRST is a dense layer");
  constexpr int REGB = P * 2 * 2048;        // bytes per channel-slice region: P planes x 2 orientations
  constexpr int XATB = P * K * 64;          // bytes of the act(Y_{l-1}) planes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_reg = smem;                                       // [NTN][2 orientations][P][32 x 64 B]
  char* s_xat = s_reg + NTN * REGB;                         // [P][K x 64 B]
  float* s_dx = reinterpret_cast<float*>(s_xat + XATB);     // [RS][K] fp32
  float* s_vy = s_dx + RS * K;                              // 5N
  float* s_px = s_vy + 5 * N;                               // pss (2K) | pmi (2K)
  float* s_w0 = s_px + 4 * K;                               // XR: W0 (K x 4), bf16-rounded in the bf16 mode
  float* s_mx = s_w0 + 4 * K;                               // H2: [2][8] slab maxima of |dY| per wave (alternating rows)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: scalar branches below
  const int lr = lane & 31, lh = lane >> 5;
  const int nt = wave % NTN, kg = wave / NTN;     // dW: channel slice nt, K tiles kg*KTW ..
  const int dkt = wave % KT, dnh = wave / KT;     // dX: K tile dkt, channel slices dnh*SL ..
  for (int i = tid; i < 5 * N; i += NT) s_vy[i] = p.vec[i];
  for (int i = tid; i < 2 * K; i += NT) {
    const int g = (i < K ? i : p.fld + i - K) + p.fc0;           // [scale | shift], [mean | invstd] of the chunk
    s_px[i] = p.pss[g];
    s_px[2 * K + i] = p.pmi[g];
  }
  if constexpr (XR) {
    for (int i = tid; i < 4 * K; i += NT) {
      const float w = p.W0[i];
      s_w0[i] = CM == 1 ? (float)(__bf16)w : w;
    }
  }

  // ---- the weight as B fragments of dX = dY.W, resident for the whole launch --------------------
  // B^T[col j = k][red = n]: lane (k = 32*dkt + lr, n = 32*(dnh*SL + sl) + 16*s + 8*lh + e)
  bf16x8 wf[SL][2][P];
#pragma unroll
  for (int sl = 0; sl < SL; ++sl)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      __bf16 t[8][P];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        split_planes<P>(p.W[(size_t)(32 * (dnh * SL + sl) + 16 * s + 8 * lh + e) * p.ldw + 32 * dkt + lr], t[e]);
#pragma unroll
      for (int q = 0; q < P; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[sl][s][q][e] = t[e][q];
    }
  f32x16 dwacc[KTW];
#pragma unroll
  for (int kk = 0; kk < KTW; ++kk)
#pragma unroll
    for (int r = 0; r < 16; ++r) dwacc[kk][r] = 0.f;

  // ---- per-lane slots ---------------------------------------------------------------------------
  // dY transform: 4 rows x CW channels; the KG waves of a region split its 32 rows
  const int cl = lane % LPB, rb = lane / LPB;
  const int t_rl = (32 / KG) * kg + 4 * rb;          // first of the lane's 4 rows inside the 32-row group
  const int t_c0 = CW * cl;                          // first of its CW channels inside the slice
  const int t_col = 32 * nt + t_c0;                  // ... inside the layer
  // act(Y_{l-1}) staging: a wave takes 8 rows x 64 channels of the 32 x K tile, a lane 4 rows x 2
  // channels.  Lanes 0-7 / 8-15 of a 16-lane store group take the two 4-row blocks of the same 16
  // channels, so that the [k][row] stores of a group spread over 8 of the 16 bank slots (2-way; with one
  // row block per group they were 4-way conflicted).
  constexpr int XU = K / 64;                         // 64-channel units per row block
  const int x_c2 = 32 * (wave % XU) + ((lane & 7) | ((lane >> 4) << 3));
  const int x_rl = 8 * (wave / XU) + 4 * ((lane >> 3) & 1);
  // epilogue: fixed float4 column of the dX tile, rows e_r0 + i*ER
  const int e_cq = tid % QK, e_r0 = tid / QK;
  using vecT = float __attribute__((ext_vector_type(CW)));
  vecT ry[4], rg[4];                                 // raw Y_l rows, raw upstream rows (dense)
  using ivecT = int __attribute__((ext_vector_type(CW)));
  vecT rdp;                                          // sparse: pooled gradient of the lane's group
  ivecT rarg;                                        //         and its arg-max slots
  float2 rx[4];                                      // raw Y_{l-1}
  float4 rq[XR ? 4 : 1];                             // XR: the 4-float input rows instead
  int r_slot = 0;
  char* reg_base = s_reg + nt * REGB;
  char* xat_base = s_xat;

  // Rows beyond R are read from row R-1 (a valid address, no exec-mask branch per load) and zeroed by
  // the transform.
  auto fetch = [&](int slab) {
    const int row0 = slab * RS;
    const int last = p.R - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = min(row0 + t_rl + j, last);
      ry[j] = load_row<CW, YB>(p.Yl, (size_t)row * N + t_col);
      if constexpr (!SPARSE) rg[j] = load_row<CW, YB>(p.G, (size_t)row * N + t_col);
      const int xrow = min(row0 + x_rl + j, last);
      if constexpr (XR) {
        rq[j] = *reinterpret_cast<const float4*>(p.fX + (size_t)xrow * 4);      // rebuilt in the transform
      } else {
        const auto xv = load_row<2, XB>(p.Xp, (size_t)xrow * p.ldk + 2 * x_c2);
        rx[j] = make_float2(xv[0], xv[1]);
      }
    }
    if constexpr (SPARSE) {
      // the lane's 4 rows start at a multiple of 4 and ns % 4 == 0: one pooled group for all of them
      const int row = min(row0 + t_rl, last);
      const int rp = row / p.ns;
      r_slot = (row0 + t_rl) - rp * p.ns;
      rdp = *reinterpret_cast<const vecT*>(p.dP + (size_t)rp * N + t_col);
      rarg = *reinterpret_cast<const ivecT*>(p.arg + (size_t)rp * N + t_col);
    }
  };

  const int nslab = (p.R + RS - 1) / RS;
  int slab = blockIdx.x;
  constexpr bool PREF = !(NTN == 8 && P == 3);   // (256,128) three-term: no registers to hold a slab ahead
  if (PREF && slab < nslab) fetch(slab);
  // (measured: staggering the start of the second workgroup per CU by 0.5 ... 4 k cycles changes nothing)
  // running column sums of the epilogue
  float es1[4] = {0.f, 0.f, 0.f, 0.f}, es2[4] = {0.f, 0.f, 0.f, 0.f};
  float fs[EPI == 1 ? 4 : 1][8];
  float fcx[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 8; ++q) fs[c][q] = 0.f;
  }
  float sc_cur = 1.f, run_max = 0.f;        // H2: the scale on dY, the largest |dY| met so far
  int par = 0;
  // dY of element (channel e, row j) of the lane's patch, from the rows fetch() holds in registers
  auto dy_at = [&](int e, int j, int row0v, bool tailv) {
    const int c = t_col + e;
    const float sc = s_vy[c], sh = s_vy[N + c], gi = s_vy[2 * N + c], va = s_vy[3 * N + c], vb = s_vy[4 * N + c];
    const float y = ry[j][e];
    const bool on = __builtin_fmaf(y, sc, sh) > 0.f;
    float dz;
    if constexpr (SPARSE) dz = (on && rarg[e] == r_slot + j) ? rdp[e] : 0.f;
    else dz = on ? rg[j][e] : 0.f;
    float d = __builtin_fmaf(gi, dz, __builtin_fmaf(va, y, vb));
    if (tailv && row0v + t_rl + j >= p.R) d = 0.f;
    return d;
  };
  // H2: the largest |dY| of the slab whose rows are in the registers, one value per wave into row `to` of s_mx
  auto publish_max = [&](int row0v, int to) {
    const bool tailv = row0v + RS > p.R;
    float m0 = 0.f;
#pragma unroll
    for (int e = 0; e < CW; ++e)
#pragma unroll
      for (int j = 0; j < 4; ++j) m0 = __builtin_fmaxf(m0, __builtin_fabsf(dy_at(e, j, row0v, tailv)));
    m0 = m0 == m0 ? m0 : 0.f;                                // (a NaN gradient must not poison the scale)
    const float wm0 = wave_allmax(m0);
    if (lane == 0) s_mx[8 * to + wave] = wm0;
  };
  if constexpr (H2) {
    static_assert(!H2 || PREF, "the scale is taken from the prefetched rows");
    __syncthreads();                                         // (the vectors in s_vy)
    if (slab < nslab) publish_max(slab * RS, 0);
  }
  __syncthreads();

  for (; slab < nslab; slab += gridDim.x) {
    const int row0 = slab * RS;
    if constexpr (!PREF) fetch(slab);
    // ---- [A] dY = gi*dZ + a*y + b on the lane's patch, split once, both orientations into LDS -------
    {
      const bool tail = row0 + RS > p.R;                     // (uniform) rows beyond R become zeros
      unsigned pr[CW][2][P];                                 // [channel][row pair][plane]
      if constexpr (H2) {
        // The scale of this slab: the largest |dY| this workgroup has met up to and INCLUDING this slab brought into
        // [8, 16).  The slab's own maximum was taken ONE SLAB AHEAD - at the end of the previous slab, from the rows
        // the prefetch had delivered by then, published in front of that slab's fold barriers (the first slab's: in
        // front of the loop) - so neither a barrier nor a second pass over the patch stands in front of the split.
        // (A scale that simply lagged one slab behind was tried first and is wrong: a workgroup's first slabs may hold
        // nothing but the BatchNorm-backward background terms, 1e5 x below the first row that carries a real gradient -
        // beyond the 2^11 of head-room; the clamp in split2_f16 then returned 10-20 % errors in SA1's first weight
        // gradient, tests/test_gpu_parity_full.py.)  The scale never rises again: the dW accumulators hold products of
        // the earlier slabs, and rows far below the running maximum do not need its last bits.
        float mx = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) mx = __builtin_fmaxf(mx, s_mx[8 * par + w]);
        par ^= 1;                                            // (the next slab's maxima go to the other row)
        if (mx > run_max && mx < 3.0e38f) {
          run_max = mx;
          const float sn = f16_scale_for(mx);
          if (sn != sc_cur) {
            const float f = sn / sc_cur;                     // (powers of two: exact)
#pragma unroll
            for (int kk = 0; kk < KTW; ++kk)
#pragma unroll
              for (int r = 0; r < 16; ++r) dwacc[kk][r] *= f;
            sc_cur = sn;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < CW; ++e) {
        float dy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dy[j] = dy_at(e, j, row0, tail);
          if constexpr (H2) dy[j] *= sc_cur;
        }
        split_pair<P>(dy[0], dy[1], pr[e][0]);
        split_pair<P>(dy[2], dy[3], pr[e][1]);
      }
#pragma unroll
      for (int q = 0; q < P; ++q) {
        // [channel][row]: the lane's 4 consecutive rows of channel t_c0 + e = its two row pairs
#pragma unroll
        for (int e = 0; e < CW; ++e)
          *reinterpret_cast<uint2*>(reg_base + (P + q) * 2048 + swz(t_c0 + e, t_rl >> 3) + 2 * (t_rl & 7)) =
              make_uint2(pr[e][0][q], pr[e][1][q]);
        // [row][channel]: CW consecutive channels of row t_rl + j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          char* d = reg_base + q * 2048 + swz(t_rl + j, t_c0 >> 3) + 2 * (t_c0 & 7);
          const int jp = j >> 1;
          if constexpr (CW == 4) {
            *reinterpret_cast<uint2*>(d) =
                (j & 1) ? make_uint2(pair_hi(pr[0][jp][q], pr[1][jp][q]), pair_hi(pr[2][jp][q], pr[3][jp][q]))
                        : make_uint2(pair_lo(pr[0][jp][q], pr[1][jp][q]), pair_lo(pr[2][jp][q], pr[3][jp][q]));
          } else {
            *reinterpret_cast<unsigned*>(d) = (j & 1) ? pair_hi(pr[0][jp][q], pr[1][jp][q])
                                                      : pair_lo(pr[0][jp][q], pr[1][jp][q]);
          }
        }
      }
      // act(Y_{l-1}) = relu(y*scale + shift) as [k][row] planes
      if constexpr (XR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 r = rq[j];
          if constexpr (CM == 1) {
            r.x = (float)(__bf16)r.x; r.y = (float)(__bf16)r.y; r.z = (float)(__bf16)r.z; r.w = (float)(__bf16)r.w;
          }
          const float4 wa = *reinterpret_cast<const float4*>(s_w0 + 4 * (2 * x_c2));
          const float4 wb = *reinterpret_cast<const float4*>(s_w0 + 4 * (2 * x_c2 + 1));
          rx[j].x = __builtin_fmaf(r.w, wa.w, __builtin_fmaf(r.z, wa.z, __builtin_fmaf(r.y, wa.y, r.x * wa.x)));
          rx[j].y = __builtin_fmaf(r.w, wb.w, __builtin_fmaf(r.z, wb.z, __builtin_fmaf(r.y, wb.y, r.x * wb.x)));
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = 2 * x_c2 + e;
        const float sc = s_px[k], sh = s_px[K + k];
        float a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[j] = fmaxf(0.f, __builtin_fmaf(e == 0 ? rx[j].x : rx[j].y, sc, sh));
          if (tail && row0 + x_rl + j >= p.R) a[j] = 0.f;
        }
        unsigned p0[P], p1[P];
        split_pair<P>(a[0], a[1], p0);
        split_pair<P>(a[2], a[3], p1);
#pragma unroll
        for (int q = 0; q < P; ++q)
          *reinterpret_cast<uint2*>(xat_base + q * (K * 64) + swz(k, x_rl >> 3) + 2 * (x_rl & 7)) =
              make_uint2(p0[q], p1[q]);
      }
    }
    // ---- [B] next slab's rows in flight underneath the MFMA phase ----------------------------------
    if (PREF && slab + (int)gridDim.x < nslab) fetch(slab + (int)gridDim.x);
    lds_barrier();        // planes of every wave in place; the last epilogue is done with the dX tile
    // ---- [C] dW += dY^T.act(Y_{l-1}); dX partial over this wave's channel slice ----------------------
    f32x16 pa;
    {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 at[P];
#pragma unroll
        for (int q = 0; q < P; ++q)
          at[q] = *reinterpret_cast<const bf16x8*>(reg_base + (P + q) * 2048 + swz(lr, 2 * t + lh));
#pragma unroll
        for (int kk = 0; kk < KTW; ++kk) {
          bf16x8 bx[P];
#pragma unroll
          for (int q = 0; q < P; ++q)
            bx[q] = *reinterpret_cast<const bf16x8*>(xat_base + q * (K * 64) +
                                                     swz(32 * (kg * KTW + kk) + lr, 2 * t + lh));
          mfma_planes<P>(dwacc[kk], at, bx);
        }
      }
      // dX tile (32 rows x K tile dkt) over this wave's SL channel slices: A fragments out of those
      // slices' [row][channel] planes, B = the resident weight fragments
#pragma unroll
      for (int r = 0; r < 16; ++r) pa[r] = 0.f;
#pragma unroll
      for (int sl = 0; sl < SL; ++sl)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 a[P];
#pragma unroll
          for (int q = 0; q < P; ++q)
            a[q] = *reinterpret_cast<const bf16x8*>(s_reg + (dnh * SL + sl) * REGB + q * 2048 + swz(lr, 2 * s + lh));
          mfma_planes<P>(pa, a, wf[sl][s]);
        }
    }
    if constexpr (H2) {
      // the next slab's rows have arrived underneath the MFMA phase: its maximum, in front of the fold barriers
      if (slab + (int)gridDim.x < nslab) publish_max((slab + (int)gridDim.x) * RS, par);
    }
    // The NH partial tiles of a K tile meet in the fp32 LDS tile in NH ordered rounds: in round j wave
    // (dkt, dnh) owns row chunk (dnh + j) % NH (32/NH rows = CR accumulator registers) - the first round
    // stores, the others read-add-write; chunks are disjoint within a round, a barrier separates rounds.
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int c = (dnh + j) % NH;
      float* d = s_dx + (size_t)(4 * lh) * K + 32 * dkt + lr;
#pragma unroll
      for (int cc = 0; cc < NH; ++cc) {
        if (cc != c) continue;                 // (static register indices: select the chunk by unrolling)
#pragma unroll
        for (int i = 0; i < CR; ++i) {
          const int r = cc * CR + i;
          float* q = d + ((r & 3) + 8 * (r >> 2)) * K;
          if (j == 0) *q = pa[r];
          else *q += pa[r];
        }
      }
      lds_barrier();
    }
    // ---- [D] dX rows out of the LDS tile: coalesced float4, + the sums of the layer below ------------
    {
      const int srow0 = slab * RS;
      const float4 sc0 = *reinterpret_cast<const float4*>(s_px + 4 * e_cq);
      const float4 sh0 = *reinterpret_cast<const float4*>(s_px + K + 4 * e_cq);
      const float4 mu0 = *reinterpret_cast<const float4*>(s_px + 2 * K + 4 * e_cq);
      const float4 is0 = *reinterpret_cast<const float4*>(s_px + 3 * K + 4 * e_cq);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rl = e_r0 + i * ER, row = srow0 + rl;
        float4 dx = *reinterpret_cast<const float4*>(s_dx + (size_t)rl * K + 4 * e_cq);
        if constexpr (H2) {
          const float inv = 1.0f / sc_cur;                   // (a power of two: exact)
          dx.x *= inv; dx.y *= inv; dx.z *= inv; dx.w *= inv;
        }
        if (row < p.R) {
          float4 y;
          if constexpr (XR) {
            float4 r = *reinterpret_cast<const float4*>(p.fX + (size_t)row * 4);
            if constexpr (CM == 1) {
              r.x = (float)(__bf16)r.x; r.y = (float)(__bf16)r.y; r.z = (float)(__bf16)r.z; r.w = (float)(__bf16)r.w;
            }
            const float4* w4 = reinterpret_cast<const float4*>(s_w0 + 16 * e_cq);
            y.x = __builtin_fmaf(r.w, w4[0].w, __builtin_fmaf(r.z, w4[0].z, __builtin_fmaf(r.y, w4[0].y, r.x * w4[0].x)));
            y.y = __builtin_fmaf(r.w, w4[1].w, __builtin_fmaf(r.z, w4[1].z, __builtin_fmaf(r.y, w4[1].y, r.x * w4[1].x)));
            y.z = __builtin_fmaf(r.w, w4[2].w, __builtin_fmaf(r.z, w4[2].z, __builtin_fmaf(r.y, w4[2].y, r.x * w4[2].x)));
            y.w = __builtin_fmaf(r.w, w4[3].w, __builtin_fmaf(r.z, w4[3].z, __builtin_fmaf(r.y, w4[3].y, r.x * w4[3].x)));
          } else {
            const auto yv = load_row<4, XB>(p.Xp, (size_t)row * p.ldk + 4 * e_cq);
            y = make_float4(yv[0], yv[1], yv[2], yv[3]);
          }
          float dz[4];
          dz[0] = __builtin_fmaf(y.x, sc0.x, sh0.x) > 0.f ? dx.x : 0.f;
          dz[1] = __builtin_fmaf(y.y, sc0.y, sh0.y) > 0.f ? dx.y : 0.f;
          dz[2] = __builtin_fmaf(y.z, sc0.z, sh0.z) > 0.f ? dx.z : 0.f;
          dz[3] = __builtin_fmaf(y.w, sc0.w, sh0.w) > 0.f ? dx.w : 0.f;
          const float xh[4] = {(y.x - mu0.x) * is0.x, (y.y - mu0.y) * is0.y, (y.z - mu0.z) * is0.z,
                               (y.w - mu0.w) * is0.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            es1[c] += dz[c];
            es2[c] = __builtin_fmaf(dz[c], xh[c], es2[c]);
          }
          if constexpr (EPI == 0) {
            if constexpr (YB) {
              unsigned o0[1], o1[1];
              split_pair<1>(dx.x, dx.y, o0);
              split_pair<1>(dx.z, dx.w, o1);
              *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(p.dX) + (size_t)row * p.ldk + 4 * e_cq) = make_uint2(o0[0], o1[0]);
            } else {
              *reinterpret_cast<float4*>(p.dX + (size_t)row * p.ldk + 4 * e_cq) = dx;
            }
          } else {
            const float4 x = *reinterpret_cast<const float4*>(p.fX + (size_t)row * 4);
            const float yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              fs[c][0] = __builtin_fmaf(dz[c], x.x, fs[c][0]);
              fs[c][1] = __builtin_fmaf(dz[c], x.y, fs[c][1]);
              fs[c][2] = __builtin_fmaf(dz[c], x.z, fs[c][2]);
              fs[c][3] = __builtin_fmaf(dz[c], x.w, fs[c][3]);
              fs[c][4] = __builtin_fmaf(yv[c], x.x, fs[c][4]);
              fs[c][5] = __builtin_fmaf(yv[c], x.y, fs[c][5]);
              fs[c][6] = __builtin_fmaf(yv[c], x.z, fs[c][6]);
              fs[c][7] = __builtin_fmaf(yv[c], x.w, fs[c][7]);
            }
            if (e_cq == 0) { fcx[0] += x.x; fcx[1] += x.y; fcx[2] += x.z; fcx[3] += x.w; }
          }
        }
      }
    }
    // (no barrier here: the next [A] writes planes that nobody reads before the next barrier, and the dX
    // tile is only written again after it)
  }

  // ---- flush: dW tiles (row groups folded through LDS first), column sums ---------------------------
  __syncthreads();
  float* s_red = reinterpret_cast<float*>(smem);             // the plane regions are free now
  if (!(p.dbg & 1)) {
    const float inv = H2 ? 1.0f / sc_cur : 1.0f;             // (fp16 terms: the accumulators carry the dY scale)
#pragma unroll
    for (int kk = 0; kk < KTW; ++kk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int k = 32 * (kg * KTW + kk) + lr;
        atomicAdd(p.dW + (size_t)n * p.ldw + k, H2 ? dwacc[kk][r] * inv : dwacc[kk][r]);
      }
  }
  // column sums: lanes with the same float4 column inside a wave first, then the 8 waves through LDS
  constexpr int NV = EPI == 1 ? 8 + 32 + 4 : 8;
  float v[NV];
#pragma unroll
  for (int c = 0; c < 4; ++c) { v[c] = es1[c]; v[4 + c] = es2[c]; }
  if constexpr (EPI == 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 8; ++q) v[8 + c * 8 + q] = fs[c][q];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[40 + c] = fcx[c];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int m = QK; m < 64; m <<= 1) v[i] += __shfl_xor(v[i], m);
  if (lane < QK) {
#pragma unroll
    for (int i = 0; i < NV; ++i) s_red[(wave * QK + lane) * NV + i] = v[i];
  }
  __syncthreads();
  // (a wave's lanes l, l + QK, ... hold the same float4 column at different rows: folded by the shuffles
  // above; NW partials per column remain)
  for (int i = tid; i < QK * NV; i += NT) {
    const int cq = i / NV, q = i % NV;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[(w * QK + cq) * NV + q];
    if (q < 8) {
      const int col = 4 * cq + (q & 3);
      if constexpr (EPI == 0) atomicAdd(p.g12 + (q >> 2) * p.fld + p.fc0 + col, (double)t);
      else atomicAdd(p.fsum + (q >> 2) * K + col, (double)t);
    } else if (q < 40) {
      // FIRST: P (K x 4) at 2K, Q (K x 4) at 6K
      const int c = (q - 8) >> 3, s = (q - 8) & 7, col = 4 * cq + c;
      atomicAdd(p.fsum + (s < 4 ? 2 * K + col * 4 + s : 6 * K + col * 4 + (s - 4)), (double)t);
    } else if (cq == 0) {
      atomicAdd(p.fsum + 10 * K + (q - 40), (double)t);
    }
  }
  if constexpr (EPI == 0) {
    if (p.vfin.ticket != nullptr) {        // layer l-1's backward vectors by the last workgroup (csrc/bn_fin.h)
      __shared__ int s_last;
      sync_drained();
      if (tid == 0) s_last = last_workgroup(p.vfin.ticket, (int)gridDim.x, (int)blockIdx.x);
      __syncthreads();
      if (s_last) bn_vec_finalize(p.vfin, p.fld, p.fc0, K, p.g12, tid, NT);
    }
  }
}

// ---- backward of a POOLED last layer without its output -------------------------------------------
// The last layer of a set-abstraction stack is  Y = A.W^T -> BN -> ReLU -> max over ns rows  with
// A = act(Y_{L-1}) (R x K).  Its upstream gradient is sparse (one selected row per group and channel):
// dZ[r,c] = dP[g,c] at r = arg[g,c] (gated by the ReLU), and the train-mode BN backward is
// dY = gi*dZ + a*y + b per channel.  Everything the layer's backward needs from the dense part is a
// function of A alone, because y = A.W^T:
//     dA = dY.W        = (gi*dZ).W  +  A.M  +  v         M = W^T diag(a) W  (K x K),  v = b^T W
//     dW = dY^T.A      = (gi*dZ)^T.A  +  diag(a) W (A^T A)  +  b (x) colsum(A)
// so the (R x N) output - 537 MB at SA1, written once by the forward and read once here - is neither
// stored nor read: the forward keeps the selected extremum, its row and its raw value per (group,
// channel) only (mlp_fwd_res_kernel ST bit 2), and this kernel walks Y_{L-1} once.  Per 32-row slab:
//   * A = relu(y*scale + shift), split once, as bf16 planes in both orientations ([row][k] for A.M,
//     [k][row] for the Gram matrix A^T A) + an fp32 copy (the sparse products read single rows of it);
//   * the slab's share of the sparse gradient (arg rows that fall into it) is scattered into zeroed
//     [row][channel] planes and contracted with the register-resident W fragments on the MFMA as well
//     (32 x 64 x 128: the same tile shape as before, but with no row of Y behind it); the entries are
//     wiped again after the MFMA phase instead of re-zeroing 24 KB per slab;
//   * S = (gi*dZ)^T.A is 128 row-AXPYs per group: 16 fp32 accumulators per thread, VALU;
//   * half the MFMA work of mlp_bwd_fused_kernel: A.M (K x K) and A^T A (K x K) replace dY.W and dY^T.A
//     (N x K each); the M fragments are formed once per workgroup from W and the vector a;
//   * epilogue as mlp_bwd_fused_kernel's RED: dA rows out of the LDS tile + v, stored, and layer L-1's
//     BN-backward sums taken on the way;
//   * the workgroup's Gram tiles, S and colsum(A) leave as PLAIN stores into its own slot of a workspace
//     (same-address fp32 atomics from 240 workgroups serialise: ~30 G/s chip-wide); pool_bwd_finish_k
//     sums the slots and forms dW.
// Shapes: N = 128, K = 64 (SA1), ns a multiple of 32 (a slab lies inside one group), R % 32 == 0.
struct PoolBwdArgs {
  int R, ns;
  const float* Xp;    // (R x K) pre-BN output of layer L-1
  const float* pss;   // [scale|shift] of layer L-1 (2K)
  const float* pmi;   // [mean|invstd] of layer L-1 (2K)
  const float* W;     // (N x K) weight of layer L
  const float* vec;   // 5 x N backward vectors of layer L (scale, shift, gi, a, b)
  const float* dP;    // (R/ns x N) gradient of the pooled output
  const int* arg;     // (R/ns x N) selected row inside the group
  const float* yraw;  // (R/ns x N) raw output at the selected row
  float* dX;          // (R x K) gradient of layer L-1's activation
  float* part;        // (gridDim.x x PB_NACC) per-workgroup contribution to dW (N x K)
  double* g12;        // layer L-1: sum dZ | sum dZ*xhat (2K), accumulated
  BnVecFin vfin;      // layer L-1's backward vectors by the last workgroup (ticket != null)
  int dbg;            // DEMF_PB_DBG phase-skip bits (measurement only)
};
constexpr int PB_N = 128, PB_K = 64, PB_RS = 64;       // rows per slab = rows per group
constexpr int PB_NACC = PB_N * PB_K;      // floats per workgroup slot
constexpr int PB_LDA = PB_K + 4;      // fp32 A copy: row stride (floats)
constexpr int PB_LDM = PB_K + 4;      // fp32 M in LDS during the prologue

// 128-byte LDS rows (64 bf16): 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7) (as fr_swz<128> in
// csrc/mlp.hip: a ds_read_b128 lane group then touches every bank once)
__device__ __forceinline__ int sw128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// One 8-wave workgroup per CU on 64-row slabs (= one group at ns = 64): half the fold rounds per row and twice
// the bytes in flight of a 32-row slab.  Measured alternatives (tools/pool_bwd_micro.py, SA1 shape, f32 mode):
// 32-row slabs / 8 waves 476 us (one-slab prefetch of 8 KB: latency-bound), this form 384 us, two 4-wave
// workgroups per CU with a two-deep register prefetch 395-431 us (the phases of a slab still add up: with
// the three-term split the transform + fold + epilogue VALU work is as long as the MFMA work).
template <int CM>
__global__ __launch_bounds__(512, 1) void mlp_bwd_pool_kernel(PoolBwdArgs p) {
  constexpr int P = CM == 2 ? 3 : (CM == 3 ? 2 : 1);
  // CM 3: two fp16 terms per operand (csrc/common.h).  The activations (A, both orientations: A.M and the Gram matrix)
  // and W are split as they are.  The two operands that carry the gradient's magnitude get a power-of-two scale each
  // and an accumulator each: M = W^T diag(a) W (scale from its largest entry, once per workgroup) and the slab's sparse
  // gi*dZ entries (scale from the slab's largest entry, taken ONE SLAB AHEAD from the prefetched words, so no barrier
  // is added); the two partial dA tiles are scaled back and added before they meet in the LDS tile.
  constexpr bool H2 = CM == 3;
  constexpr int N = PB_N, K = PB_K, NTN = N / 32, RS = PB_RS, NT = 512, NW = 8;
  constexpr int QK = K / 4;                 // float4 per dX row (16): thread -> (row = tid / 16 (+32), col4 = tid % 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_vy = reinterpret_cast<float*>(smem);              // 5N
  float* s_px = s_vy + 5 * N;                                // pss (2K) | pmi (2K)
  float* s_v = s_px + 4 * K;                                 // v = b^T W (K)
  float* s_mx = s_v + K;                                     // H2: [4][8] per-wave maxima (M | gi*dZ of alternating slabs)
  char* s_base = reinterpret_cast<char*>(s_mx + 32);         // everything below is also prologue / flush scratch
  char* s_dz = s_base;                                       // [NTN][P][64 rows x 64 B]   gi*dZ, [row][channel]
  char* s_ar = s_dz + NTN * P * 4096;                        // [2 k-halves][P][64 rows x 64 B]  A, [row][k]
  char* s_at = s_ar + 2 * P * 4096;                          // [P][64 k x 128 B]          A, [k][row]
  float* s_yf = reinterpret_cast<float*>(s_at + P * 8192);   // [2 buffers][64][PB_LDA] fp32 RAW rows of Y_{L-1}
  float* s_dx = s_yf + 2 * RS * PB_LDA;                      // [64][K] fp32 dA tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int drt = wave & 1, dct = (wave >> 1) & 1, dkh = wave >> 2;   // dA: tile (row half drt, column half dct), reduction half dkh
  const int gti = (wave & 3) >> 1, gtj = wave & 1, gs = wave >> 2;    // Gram: tile (gti, gtj), row half gs of the slab
  for (int i = tid; i < 5 * N; i += NT) s_vy[i] = p.vec[i];
  for (int i = tid; i < 2 * K; i += NT) { s_px[i] = p.pss[i]; s_px[2 * K + i] = p.pmi[i]; }
  // ---- prologue: M = W^T diag(a) W and v = b^T W from an fp32 copy of W in LDS (the plane regions serve
  // as scratch: 32 KB of W + 17 KB of M) ---------------------------------------------------------------
  float* s_w = reinterpret_cast<float*>(s_base);             // [N][K]
  float* s_m = s_w + N * K;                                  // [K][PB_LDM]
  for (int i = tid; i < N * K / 4; i += NT)
    reinterpret_cast<float4*>(s_w)[i] = reinterpret_cast<const float4*>(p.W)[i];
  __syncthreads();
  {
    const int k = tid >> 3, j0 = (tid & 7) * 8;
    float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float vk = 0.f;
    for (int c = 0; c < ((p.dbg & 32) ? 1 : N); ++c) {
      const float wa = s_w[c * K + k] * s_vy[3 * N + c];
      const float4 w0 = *reinterpret_cast<const float4*>(s_w + c * K + j0);
      const float4 w1 = *reinterpret_cast<const float4*>(s_w + c * K + j0 + 4);
      m[0] = __builtin_fmaf(wa, w0.x, m[0]); m[1] = __builtin_fmaf(wa, w0.y, m[1]);
      m[2] = __builtin_fmaf(wa, w0.z, m[2]); m[3] = __builtin_fmaf(wa, w0.w, m[3]);
      m[4] = __builtin_fmaf(wa, w1.x, m[4]); m[5] = __builtin_fmaf(wa, w1.y, m[5]);
      m[6] = __builtin_fmaf(wa, w1.z, m[6]); m[7] = __builtin_fmaf(wa, w1.w, m[7]);
      if ((tid & 7) == 0) vk = __builtin_fmaf(s_vy[4 * N + c], s_w[c * K + k], vk);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_m[k * PB_LDM + j0 + e] = m[e];
    if ((tid & 7) == 0) s_v[k] = vk;
    if constexpr (H2) {
      float mm = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) mm = __builtin_fmaxf(mm, __builtin_fabsf(m[e]));
      mm = mm == mm ? mm : 0.f;
      const float wmm = wave_allmax(mm);
      if (lane == 0) s_mx[wave] = wmm;
    }
  }
  __syncthreads();
  float sM = 1.f;                                            // H2: the scale on M
  if constexpr (H2) {
    float mx = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) mx = __builtin_fmaxf(mx, s_mx[w]);
    if (mx > 0.f && mx < 3.0e38f) sM = f16_scale_for(mx);
  }
  // B fragments, resident for the whole launch.  A.M: B^T[col j][red k] = M[k][j], this wave's k half (two steps)
  bf16x8 mf[2][P];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    __bf16 t[8][P];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      split_planes<P>(s_m[(32 * dkh + 16 * u + 8 * lh + e) * PB_LDM + 32 * dct + lr] * (H2 ? sM : 1.f), t[e]);
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) mf[u][q][e] = t[e][q];
  }
  // (gi*dZ).W: B^T[col j = k][red = channel], this wave's channel half (four 16-channel steps)
  bf16x8 wf[4][P];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    __bf16 t[8][P];
#pragma unroll
    for (int e = 0; e < 8; ++e) split_planes<P>(s_w[(64 * dkh + 16 * u + 8 * lh + e) * K + 32 * dct + lr], t[e]);
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) wf[u][q][e] = t[e][q];
  }
  __syncthreads();
  // the gi*dZ planes start (and are kept) all-zero
  for (int i = tid; i < NTN * P * 4096 / 16; i += NT)
    reinterpret_cast<uint4*>(s_dz)[i] = make_uint4(0u, 0u, 0u, 0u);
  f32x16 gacc;                                               // this wave's share of its Gram tile
#pragma unroll
  for (int r = 0; r < 16; ++r) gacc[r] = 0.f;
  float sacc[16];                                            // S[c][16*part ..): channel c = tid / 4
#pragma unroll
  for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
  const int s_c = tid >> 2, s_part = tid & 3;
  // transform map: thread -> channel pair (2*t_k2, +1) of rows 4*t_rq .. 4*t_rq + 3
  const int t_k2 = tid & 31, t_rq = tid >> 5;
  const int e_cq = tid % QK, e_rl = tid / QK;                // epilogue: float4 column e_cq of rows e_rl, e_rl + 32
  float es1[4] = {0.f, 0.f, 0.f, 0.f}, es2[4] = {0.f, 0.f, 0.f, 0.f}, ecs[4] = {0.f, 0.f, 0.f, 0.f};
  float2 rx[4];
  int r_arg = 0;                                             // channel s_c of the slab's group: selected row,
  float r_y = 0.f, r_dp = 0.f;                               // raw output there, pooled gradient
  // (all of a slab's operands are requested one slab ahead: the sparse entry's three words would
  // otherwise be two dependent global-load latencies in front of every slab's barrier)
  auto fetch = [&](int slab) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      rx[j] = *reinterpret_cast<const float2*>(p.Xp + (size_t)(slab * RS + 4 * t_rq + j) * K + 2 * t_k2);
    const size_t o = (size_t)slab * N + s_c;                 // slab == group (ns == 64)
    r_arg = p.arg[o];
    r_y = p.yraw[o];
    r_dp = p.dP[o];
  };
  const int nslab = p.R / RS;
  int slab = blockIdx.x;
  if (slab < nslab) fetch(slab);
  // H2: the largest |gi*dZ| entry of a slab, from the words fetch() has just requested for it
  auto dz_max_to = [&](float* dst) {
    float d = __builtin_fmaf(r_y, s_vy[s_c], s_vy[N + s_c]) > 0.f ? __builtin_fabsf(s_vy[2 * N + s_c] * r_dp) : 0.f;
    d = d == d ? d : 0.f;
    const float wm = wave_allmax(d);
    if (lane == 0) dst[wave] = wm;
  };
  auto dz_scale_from = [&](const float* src) {
    float mx = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) mx = __builtin_fmaxf(mx, src[w]);
    return (mx > 0.f && mx < 3.0e38f) ? f16_scale_for(mx) : 1.f;
  };
  float sD = 1.f;                                            // H2: the scale on this slab's gi*dZ entries
  if constexpr (H2) {
    if (slab < nslab) dz_max_to(s_mx + 8);
  }
  __syncthreads();
  if constexpr (H2) sD = dz_scale_from(s_mx + 8);

  for (int it = 0; slab < nslab; slab += gridDim.x, ++it) {
    const int row0 = slab * RS;
    float* yf = s_yf + (it & 1) * RS * PB_LDA;               // raw rows: two buffers, so that the next slab's [A]
    //                                                          does not wait for this slab's epilogue
    // ---- [A] raw rows (fp32) + A = relu(y*scale + shift), split ONCE (pairs along rows), planes in both
    // orientations: [k][row] takes the pairs as they are, [row][k] re-pairs two channels with one v_perm --------
    if (!(p.dbg & 1)) {
      const int kk = 2 * t_k2;
      unsigned pr[2][2][P];                                  // [channel e][row pair][plane]
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float sc = s_px[kk + e], sh = s_px[K + kk + e];
        float a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = fmaxf(0.f, __builtin_fmaf(e == 0 ? rx[j].x : rx[j].y, sc, sh));
        split_pair<P>(a[0], a[1], pr[e][0]);
        split_pair<P>(a[2], a[3], pr[e][1]);
        const int r = 4 * t_rq;
#pragma unroll
        for (int q = 0; q < P; ++q)
          *reinterpret_cast<uint2*>(s_at + q * 8192 + sw128(kk + e, r >> 3) + 2 * (r & 7)) = make_uint2(pr[e][0][q], pr[e][1][q]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * t_rq + j, jp = j >> 1;
        *reinterpret_cast<float2*>(yf + r * PB_LDA + kk) = rx[j];
#pragma unroll
        for (int q = 0; q < P; ++q)
          *reinterpret_cast<unsigned*>(s_ar + ((kk >> 5) * P + q) * 4096 + swz(r, (kk & 31) >> 3) + 2 * (kk & 7)) =
              (j & 1) ? pair_hi(pr[0][jp][q], pr[1][jp][q]) : pair_lo(pr[0][jp][q], pr[1][jp][q]);
      }
    }
    // ---- the group's sparse entries: channel s_c's selected row ------------------------------------------
    const int e_row = r_arg;
    const float dzv = __builtin_fmaf(r_y, s_vy[s_c], s_vy[N + s_c]) > 0.f ? s_vy[2 * N + s_c] * r_dp : 0.f;
    if (s_part == 0) {
      __bf16 t[P];
      split_planes<P>(H2 ? dzv * sD : dzv, t);
#pragma unroll
      for (int q = 0; q < P; ++q)
        *reinterpret_cast<__bf16*>(s_dz + ((s_c >> 5) * P + q) * 4096 + swz(e_row, (s_c & 31) >> 3) + 2 * (s_c & 7)) = t[q];
    }
    // ---- [B] next slab's operands in flight underneath the MFMA phase --------------------------------------
    if (slab + (int)gridDim.x < nslab) fetch(slab + (int)gridDim.x);
    lds_barrier();
    // ---- [C] S += (gi*dZ)^T.A (VALU, one raw row -> activation on the fly); dA partial and Gram tile (MFMA)
    if (!(p.dbg & 16)) {
      const float* yr = yf + e_row * PB_LDA + 16 * s_part;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const float4 v = *reinterpret_cast<const float4*>(yr + 4 * i4);
        const float4 sc = *reinterpret_cast<const float4*>(s_px + 16 * s_part + 4 * i4);
        const float4 sh = *reinterpret_cast<const float4*>(s_px + K + 16 * s_part + 4 * i4);
        sacc[4 * i4 + 0] = __builtin_fmaf(dzv, fmaxf(0.f, __builtin_fmaf(v.x, sc.x, sh.x)), sacc[4 * i4 + 0]);
        sacc[4 * i4 + 1] = __builtin_fmaf(dzv, fmaxf(0.f, __builtin_fmaf(v.y, sc.y, sh.y)), sacc[4 * i4 + 1]);
        sacc[4 * i4 + 2] = __builtin_fmaf(dzv, fmaxf(0.f, __builtin_fmaf(v.z, sc.z, sh.z)), sacc[4 * i4 + 2]);
        sacc[4 * i4 + 3] = __builtin_fmaf(dzv, fmaxf(0.f, __builtin_fmaf(v.w, sc.w, sh.w)), sacc[4 * i4 + 3]);
      }
    }
    f32x16 pa;
#pragma unroll
    for (int r = 0; r < 16; ++r) pa[r] = 0.f;
    if (!(p.dbg & 2)) {
      bf16x8 a[P];
#pragma unroll
      for (int u = 0; u < 2; ++u) {                          // A.M: k half dkh, steps 16u: region dkh, chunk 2u + lh
#pragma unroll
        for (int q = 0; q < P; ++q)
          a[q] = *reinterpret_cast<const bf16x8*>(s_ar + (dkh * P + q) * 4096 + swz(32 * drt + lr, 2 * u + lh));
        mfma_planes<P>(pa, a, mf[u]);
      }
      f32x16 pd;                                             // H2: the (gi*dZ).W partial, under its own scale
      if constexpr (H2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pd[r] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {                          // (gi*dZ).W: channel half dkh = slices 2dkh, 2dkh + 1
#pragma unroll
        for (int q = 0; q < P; ++q)
          a[q] = *reinterpret_cast<const bf16x8*>(s_dz + ((2 * dkh + (u >> 1)) * P + q) * 4096 + swz(32 * drt + lr, 2 * (u & 1) + lh));
        if constexpr (H2) mfma_planes<P>(pd, a, wf[u]);
        else mfma_planes<P>(pa, a, wf[u]);
      }
      if constexpr (H2) {
        const float iM = 1.0f / sM, iD = 1.0f / sD;          // (powers of two: exact)
#pragma unroll
        for (int r = 0; r < 16; ++r) pa[r] = __builtin_fmaf(pd[r], iD, pa[r] * iM);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {                          // Gram tile: slab rows 32 * gs + 16u .. + 15
        bf16x8 ga[P], gb[P];
#pragma unroll
        for (int q = 0; q < P; ++q) {
          ga[q] = *reinterpret_cast<const bf16x8*>(s_at + q * 8192 + sw128(32 * gti + lr, 4 * gs + 2 * u + lh));
          gb[q] = *reinterpret_cast<const bf16x8*>(s_at + q * 8192 + sw128(32 * gtj + lr, 4 * gs + 2 * u + lh));
        }
        mfma_planes<P>(gacc, ga, gb);
      }
    }
    const bool has_next = slab + (int)gridDim.x < nslab;
    if constexpr (H2) {
      // the next slab's entries have arrived underneath the MFMA phase: their maximum, in front of the fold barriers
      if (has_next) dz_max_to(s_mx + 16 + 8 * (it & 1));
    }
    // the two partial tiles of a (row half, column half) meet in the fp32 LDS tile in two ordered rounds: in
    // round j wave (.., dkh) owns row chunk (dkh + j) % 2 = 8 accumulator registers
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = (dkh + j) & 1;
      float* d = s_dx + (size_t)(32 * drt + 4 * lh) * K + 32 * dct + lr;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (cc != c) continue;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = cc * 8 + i;
          float* q = d + ((r & 3) + 8 * (r >> 2)) * K;
          if (j == 0) *q = pa[r];
          else *q += pa[r];
        }
      }
      lds_barrier();
    }
    if constexpr (H2) {
      if (has_next) sD = dz_scale_from(s_mx + 16 + 8 * (it & 1));
    }
    // wipe this slab's entries: the gi*dZ planes are all-zero again (nobody reads them before the next barrier)
    if (s_part == 0) {
#pragma unroll
      for (int q = 0; q < P; ++q)
        *reinterpret_cast<unsigned short*>(s_dz + ((s_c >> 5) * P + q) * 4096 + swz(e_row, (s_c & 31) >> 3) + 2 * (s_c & 7)) = 0;
    }
    // ---- [D] dA rows out of the LDS tile (+ v), stored; layer L-1's sums and colsum(A) on the way ---------
    {
      const float4 vv = *reinterpret_cast<const float4*>(s_v + 4 * e_cq);
      const float4 sc0 = *reinterpret_cast<const float4*>(s_px + 4 * e_cq);
      const float4 sh0 = *reinterpret_cast<const float4*>(s_px + K + 4 * e_cq);
      const float4 mu0 = *reinterpret_cast<const float4*>(s_px + 2 * K + 4 * e_cq);
      const float4 is0 = *reinterpret_cast<const float4*>(s_px + 3 * K + 4 * e_cq);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rl = e_rl + 32 * i;
        float4 dx = *reinterpret_cast<const float4*>(s_dx + (size_t)rl * K + 4 * e_cq);
        dx.x += vv.x; dx.y += vv.y; dx.z += vv.z; dx.w += vv.w;
        const float4 y = *reinterpret_cast<const float4*>(yf + rl * PB_LDA + 4 * e_cq);
        const float z[4] = {__builtin_fmaf(y.x, sc0.x, sh0.x), __builtin_fmaf(y.y, sc0.y, sh0.y),
                            __builtin_fmaf(y.z, sc0.z, sh0.z), __builtin_fmaf(y.w, sc0.w, sh0.w)};
        const float dxa[4] = {dx.x, dx.y, dx.z, dx.w};
        const float xh[4] = {(y.x - mu0.x) * is0.x, (y.y - mu0.y) * is0.y, (y.z - mu0.z) * is0.z, (y.w - mu0.w) * is0.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float dz = z[c] > 0.f ? dxa[c] : 0.f;
          es1[c] += dz;
          es2[c] = __builtin_fmaf(dz, xh[c], es2[c]);
          ecs[c] += fmaxf(z[c], 0.f);
        }
        if (!(p.dbg & 8)) *reinterpret_cast<float4*>(p.dX + (size_t)(row0 + rl) * K + 4 * e_cq) = dx;
      }
    }
    // (no barrier here: the next slab writes the OTHER raw-row buffer and planes that nobody reads before its
    // barrier; the dA tile is only written again after that barrier; this slab's raw rows are overwritten two
    // slabs on, behind three more barriers)
  }

  // ---- flush.  The layer's weight gradient is linear in this workgroup's sums:
  //     dW_wg = S + diag(a) W Gm + b (x) cs        (S already carries gi)
  // formed here and stored PLAINLY into the workgroup's slot (same-address fp32 atomics from 240 workgroups
  // serialise: ~30 G/s chip-wide); pool_bwd_finish_k adds the slots up.  Column sums go through LDS.
  __syncthreads();
  float* slot = p.part + (size_t)blockIdx.x * PB_NACC;
  float* s_g = reinterpret_cast<float*>(s_base);             // [2 halves][4 tiles][16][64]: the two row halves of a Gram tile
  float* s_gm = s_g + 8 * 16 * 64;                           // [K][K + 1] Gm
  float* s_cs = s_gm + K * (K + 1);                          // [K] colsum(A)
  float* s_w2 = s_cs + K;                                    // [N][K] W
  {
    float* mine = s_g + ((gs * 4 + (wave & 3)) * 16) * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = gacc[r];
  }
  for (int i = tid; i < N * K / 4; i += NT)
    reinterpret_cast<float4*>(s_w2)[i] = reinterpret_cast<const float4*>(p.W)[i];
  // column sums: lanes l, l + 16, l + 32, l + 48 of a wave hold the same float4 column at different rows
  constexpr int NV = 12;
  float v[NV];
#pragma unroll
  for (int c = 0; c < 4; ++c) { v[c] = es1[c]; v[4 + c] = es2[c]; v[8 + c] = ecs[c]; }
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int m = QK; m < 64; m <<= 1) v[i] += __shfl_xor(v[i], m);
  float* s_red = s_w2 + N * K;                               // [NW][QK][NV]
  if (lane < QK) {
#pragma unroll
    for (int i = 0; i < NV; ++i) s_red[(wave * QK + lane) * NV + i] = v[i];
  }
  __syncthreads();
  if (gs == 0) {
    const float* a0 = s_g + ((wave & 3) * 16) * 64;
    const float* a1 = s_g + ((4 + (wave & 3)) * 16) * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 32 * gti + (r & 3) + 8 * (r >> 2) + 4 * lh, j = 32 * gtj + lr;
      s_gm[i * (K + 1) + j] = a0[r * 64 + lane] + a1[r * 64 + lane];
    }
  }
  for (int i = tid; i < QK * NV; i += NT) {
    const int cq = i / NV, q = i % NV;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[(w * QK + cq) * NV + q];
    const int col = 4 * cq + (q & 3);
    if (q < 8) atomicAdd(p.g12 + (q >> 2) * K + col, (double)t);
    else s_cs[col] = t;
  }
  __syncthreads();
  {
    // thread (c = tid / 4, part): dW_wg[c][16 part ..) = sacc + a_c * sum_j W[c][j] Gm[j][k] + b_c * cs[k]
    const float va = s_vy[3 * N + s_c], vb = s_vy[4 * N + s_c];
    float t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0.f;
    for (int j = 0; j < K; ++j) {
      const float w = s_w2[s_c * K + j];
      const float* gr = s_gm + j * (K + 1) + 16 * s_part;
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = __builtin_fmaf(w, gr[i], t[i]);
    }
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * i4 + e;
        o[e] = sacc[i] + va * t[i] + vb * s_cs[16 * s_part + i];
      }
      *reinterpret_cast<float4*>(slot + (size_t)s_c * K + 16 * s_part + 4 * i4) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  if (p.vfin.ticket != nullptr) {          // layer L-1's backward vectors by the last workgroup (csrc/bn_fin.h)
    __shared__ int s_last;
    sync_drained();
    if (tid == 0) s_last = last_workgroup(p.vfin.ticket, (int)gridDim.x, (int)blockIdx.x);
    __syncthreads();
    if (s_last) bn_vec_finalize(p.vfin, K, 0, K, p.g12, tid, NT);
  }
}

// dW = the sum of the workgroups' slots.  32 consecutive elements x 8 slot lanes per block: a thread adds every
// 8th slot (8 loads in flight), the slot lanes fold through LDS.  (One thread per element walking all the
// slots was a chain of dependent cache misses: ~60 us.)
__global__ __launch_bounds__(256) void pool_bwd_finish_k(int nslots, const float* __restrict__ part,
                                                         float* __restrict__ dW) {
  __shared__ float s_f[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < PB_NACC) {
    int s = sl;
    for (; s + 56 < nslots; s += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] += part[(size_t)(s + 8 * u) * PB_NACC + e];
    }
    for (; s < nslots; s += 8) t[0] += part[(size_t)s * PB_NACC + e];
  }
  s_f[sl][el] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  __syncthreads();
  if (sl == 0 && e < PB_NACC)
    dW[e] = ((s_f[0][el] + s_f[1][el]) + (s_f[2][el] + s_f[3][el])) + ((s_f[4][el] + s_f[5][el]) + (s_f[6][el] + s_f[7][el]));
}

// DEMF_STATIC_TILES=1 (no counter ring): there is no ticket for the last workgroup, so layer l-1's backward vectors
// are formed by a launch of their own behind the pass that produced the sums - the documented contract of the
// *_v entry points ("complete on return") holds either way.  (Found by tests/test_gpu_switches.py: the vectors were
// simply never written in that mode.)
static int vectors_without_ticket(const BnVecFin& vf, int K, double* g12, demf_stream_t stream) {
  if (vf.gamma == nullptr || vf.ticket != nullptr) return DEMF_OK;
  return demf_bn_bwd_vectors(K, (long long)vf.count, g12, vf.gamma, vf.ss, vf.mi, vf.vec, vf.dgamma, vf.dbeta, stream);
}

template <int NTN, int KT, int KG, bool SPARSE, int CM, int EPI, int ST = 0>
static int launch_fused(const FusedBwdArgs& a, hipStream_t s) {
  constexpr int P = CM == 2 ? 3 : (CM == 3 ? 2 : 1);
  constexpr int N = NTN * 32, K = KT * 32, NW = NTN * KG;
  const size_t bytes = (size_t)NTN * P * 2 * 2048 + (size_t)P * K * 64 + sizeof(float) * (32 * K + 5 * N + 4 * K + 4 * K + 24);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_fused_kernel<NTN, KT, KG, SPARSE, CM, EPI, ST>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
      set_error("mlp_bwd_fused: cannot reserve %zu bytes of LDS", bytes);
      return DEMF_ELAUNCH;
    }
    configured = true;
  }
  const int nslab = (a.R + 31) / 32;
  // persistent: 8 waves per CU on DEMF_PERSIST_CUS compute units.  Default 240 of 256: the coordinate
  // pre-pass of the next batch (8 FPS workgroups that each own a CU) is resident underneath the step;
  // a grid sized for all 256 CUs leaves 8 workgroups waiting for a CU and the launch ends with them
  // (measured on the step: 256 -> 6.41 ms, 248 -> 6.35, 240 -> 6.32, 232 -> 6.31, 224 -> 6.33; alone 5.86 either way)
  static const int cus = [] {
    const char* b = getenv("DEMF_PERSIST_CUS_BWD");       // the backward kernels alone (A/B)
    const char* v = b ? b : getenv("DEMF_PERSIST_CUS");
    return v ? atoi(v) : 240;
  }();
  const int cap = cus * (8 / NW);
  const int gx = nslab < cap ? nslab : cap;
  hipLaunchKernelGGL((mlp_bwd_fused_kernel<NTN, KT, KG, SPARSE, CM, EPI, ST>), dim3(gx), dim3(64 * NW), bytes, s, a);
  return check_launch("mlp_bwd_fused");
}

}  // namespace demf

using namespace demf;

// A/B switch: DEMF_F16_TERMS_BWD=0 keeps the fused backward kernels on three bf16 terms while the forward kernels
// take two fp16 terms
static int fused_h2_mode() {
  static const int on = [] { const char* v = getenv("DEMF_F16_TERMS_BWD"); return v ? atoi(v) : 1; }();
  return on;
}
static bool fused_h2_on() { return fused_h2_mode() != 0; }

static int fused_supported(int R, int N, int K, int ns, int sparse, int first) {
  const int cm = compute_mode();
  if (cm != 1 && cm != 2) return 0;
  if (R < 1) return 0;
  if (sparse && (ns < 4 || ns % 4 != 0 || R % ns != 0)) return 0;
  if (first) return (N == 64 && K == 64 && !sparse) ? 1 : 0;
  return ((N == 128 && K == 64) || (N == 128 && K == 128) || (N == 64 && K == 64) || (N == 256 && K == 128)) ? 1 : 0;
}

extern "C" int demf_mlp_bwd_fused(int R, int N, int K, const float* G, const float* dP, const int* arg,
                                  int ns, const float* Y, const float* vec6, const float* W,
                                  const float* Yprev, const float* scale_shift_prev,
                                  const float* mean_invstd_prev, float* dX, float* dW, double* g12_prev,
                                  const float* X0, double* first_sums, const float* gamma_prev,
                                  float* vec6_prev, float* dgamma_prev, float* dbeta_prev,
                                  int store_flags, demf_stream_t stream) {
  const bool sparse = G == nullptr, first = first_sums != nullptr;
  DEMF_REQUIRE(fused_supported(R, N, K, ns, sparse, first),
               "mlp_bwd_fused: unsupported shape / mode R=%d N=%d K=%d ns=%d sparse=%d first=%d mode=%d",
               R, N, K, ns, (int)sparse, (int)first, compute_mode());
  DEMF_REQUIRE(Y && vec6 && W && Yprev && scale_shift_prev && mean_invstd_prev && dW &&
                   (G || (dP && arg)) && (first ? (X0 != nullptr) : (dX && g12_prev)),
               "mlp_bwd_fused: null pointer");
  FusedBwdArgs a{};
  { static const int dbg = getenv("DEMF_BWD_DBG") ? atoi(getenv("DEMF_BWD_DBG")) : 0; a.dbg = dbg; }
  a.R = R; a.N = N; a.K = K; a.ns = ns; a.Yl = Y; a.G = G; a.dP = dP; a.arg = arg; a.vec = vec6;
  a.Xp = Yprev; a.pss = scale_shift_prev; a.pmi = mean_invstd_prev; a.W = W; a.dX = dX; a.dW = dW;
  a.g12 = g12_prev; a.fX = X0; a.fsum = first_sums;
  a.ldk = K; a.ldw = K; a.fld = K; a.fc0 = 0;
  if (!first && gamma_prev != nullptr) {
    DEMF_REQUIRE(vec6_prev && dgamma_prev && dbeta_prev, "mlp_bwd_fused: vectors of layer l-1 need all three outputs");
    a.vfin = BnVecFin{(double)R, gamma_prev, scale_shift_prev, mean_invstd_prev, vec6_prev, dgamma_prev,
                      dbeta_prev, sched_slot()};
  }
  hipStream_t s = (hipStream_t)stream;
  const int cm = compute_mode();
  if (store_flags != 0) {
    // bf16 row storage (bf16 compute mode): the two forms SA1's stack uses
    DEMF_REQUIRE(cm == 1, "mlp_bwd_fused: bf16 row storage needs the bf16 compute mode");
    if (first && store_flags == 2 && N == 64 && K == 64) return launch_fused<2, 2, 2, false, 1, 1, 2>(a, s);
    if (!first && sparse && store_flags == 3 && N == 128 && K == 64) return launch_fused<4, 2, 1, true, 1, 0, 3>(a, s);
    set_error("mlp_bwd_fused: bf16 storage form store_flags=%d N=%d K=%d sparse=%d first=%d not built",
              store_flags, N, K, (int)sparse, (int)first);
    return DEMF_EUNSUPPORTED;
  }
  // two fp16 terms: the (256, 128) and (128, 128) forms - measured 141.6 -> 115.4 and 86.7 -> 81.0 us per launch; SA1's
  // (64, 64) forms gain nothing (232 -> 229: short reductions, the pass over the patch for the scale eats the saving)
  // and stay on three bf16 terms (DEMF_F16_TERMS_BWD=2: every form, 0: none)
  const bool h2 = f16_terms() && fused_h2_on() && (((N == 256 || N == 128) && K == 128) || fused_h2_mode() == 2);
#define FGO(NTNv, KTv, KGv, SPv, EPv)                                                                                     \
  do {                                                                                                                   \
    const int rc_ = cm == 1 ? launch_fused<NTNv, KTv, KGv, SPv, 1, EPv>(a, s)                                            \
                            : (h2 ? launch_fused<NTNv, KTv, KGv, SPv, 3, EPv>(a, s) : launch_fused<NTNv, KTv, KGv, SPv, 2, EPv>(a, s)); \
    return rc_ ? rc_ : vectors_without_ticket(a.vfin, K, g12_prev, stream);                                              \
  } while (0)
  if (first) { FGO(2, 2, 2, false, 1); }
  if (N == 128 && K == 64) { if (sparse) { FGO(4, 2, 1, true, 0); } else { FGO(4, 2, 1, false, 0); } }
  if (N == 128 && K == 128) { if (sparse) { FGO(4, 4, 2, true, 0); } else { FGO(4, 4, 2, false, 0); } }
  if (N == 256 && K == 128) { if (sparse) { FGO(8, 4, 1, true, 0); } else { FGO(8, 4, 1, false, 0); } }
  if (sparse) { FGO(2, 2, 2, true, 0); } else { FGO(2, 2, 2, false, 0); }
#undef FGO
}

// The same pass for a layer whose input is wider than the kernel's K tile: columns c0 .. c0 + Kc of the
// Ktot channels of layer l-1 (Kc a supported K: 64 / 128).  One call per column chunk: each rebuilds the
// dY row tiles (as the dx + dW launches it replaces did) and produces its chunk of dX, dW, the BN-backward
// sums and - by its last workgroup - the vectors of those channels.  Pointers are the WHOLE layer's.
extern "C" int demf_mlp_bwd_fused_cols(int R, int N, int Ktot, int c0, int Kc, const float* G, const float* dP,
                                       const int* arg, int ns, const float* Y, const float* vec6,
                                       const float* W, const float* Yprev, const float* scale_shift_prev,
                                       const float* mean_invstd_prev, float* dX, float* dW, double* g12_prev,
                                       const float* gamma_prev, float* vec6_prev, float* dgamma_prev,
                                       float* dbeta_prev, demf_stream_t stream) {
  const bool sparse = G == nullptr;
  DEMF_REQUIRE(fused_supported(R, N, Kc, ns, sparse, 0) && c0 >= 0 && c0 + Kc <= Ktot && c0 % 4 == 0 && Ktot % 4 == 0,
               "mlp_bwd_fused_cols: unsupported shape / mode R=%d N=%d Ktot=%d c0=%d Kc=%d ns=%d sparse=%d mode=%d",
               R, N, Ktot, c0, Kc, ns, (int)sparse, compute_mode());
  DEMF_REQUIRE(Y && vec6 && W && Yprev && scale_shift_prev && mean_invstd_prev && dW && (G || (dP && arg)) && dX &&
                   g12_prev, "mlp_bwd_fused_cols: null pointer");
  FusedBwdArgs a{};
  { static const int dbg = getenv("DEMF_BWD_DBG") ? atoi(getenv("DEMF_BWD_DBG")) : 0; a.dbg = dbg; }
  a.R = R; a.N = N; a.K = Kc; a.ns = ns; a.Yl = Y; a.G = G; a.dP = dP; a.arg = arg; a.vec = vec6;
  a.Xp = Yprev + c0; a.dX = dX + c0; a.W = W + c0; a.dW = dW + c0;
  a.pss = scale_shift_prev; a.pmi = mean_invstd_prev; a.g12 = g12_prev;
  a.ldk = Ktot; a.ldw = Ktot; a.fld = Ktot; a.fc0 = c0;
  if (gamma_prev != nullptr) {
    DEMF_REQUIRE(vec6_prev && dgamma_prev && dbeta_prev, "mlp_bwd_fused_cols: vectors of layer l-1 need all three outputs");
    a.vfin = BnVecFin{(double)R, gamma_prev, scale_shift_prev, mean_invstd_prev, vec6_prev, dgamma_prev,
                      dbeta_prev, sched_slot()};
  }
  hipStream_t s = (hipStream_t)stream;
  const int cm = compute_mode();
  const bool h2 = f16_terms() && fused_h2_on() && (((N == 256 || N == 128) && Kc == 128) || fused_h2_mode() == 2);
#define FGO(NTNv, KTv, KGv, SPv)                                                                                          \
  do {                                                                                                                   \
    const int rc_ = cm == 1 ? launch_fused<NTNv, KTv, KGv, SPv, 1, 0>(a, s)                                             \
                            : (h2 ? launch_fused<NTNv, KTv, KGv, SPv, 3, 0>(a, s) : launch_fused<NTNv, KTv, KGv, SPv, 2, 0>(a, s)); \
    if (rc_ || c0 + Kc < Ktot) return rc_;       /* (no ticket: the vectors of ALL channels behind the last chunk) */    \
    return vectors_without_ticket(a.vfin, Ktot, g12_prev, stream);                                                       \
  } while (0)
  if (N == 128 && Kc == 64) { if (sparse) { FGO(4, 2, 1, true); } else { FGO(4, 2, 1, false); } }
  if (N == 128 && Kc == 128) { if (sparse) { FGO(4, 4, 2, true); } else { FGO(4, 4, 2, false); } }
  if (N == 256 && Kc == 128) { if (sparse) { FGO(8, 4, 1, true); } else { FGO(8, 4, 1, false); } }
  if (sparse) { FGO(2, 2, 2, true); } else { FGO(2, 2, 2, false); }
#undef FGO
}

static int pool_bwd_grid(int R) {
  static const int cus = [] {
    const char* b = getenv("DEMF_PERSIST_CUS_BWD");
    const char* v = b ? b : getenv("DEMF_PERSIST_CUS");
    return v ? atoi(v) : 240;
  }();
  const int nslab = R / PB_RS;
  return nslab < cus ? (nslab > 0 ? nslab : 1) : cus;
}

// floats of workspace demf_mlp_bwd_pool needs: one slot per workgroup + the totals
extern "C" int demf_mlp_bwd_pool_ws(int R, long long* floats) {
  DEMF_REQUIRE(R >= 0 && floats, "mlp_bwd_pool_ws: bad arguments");
  *floats = (long long)pool_bwd_grid(R) * PB_NACC;
  return DEMF_OK;
}

// Backward of a pooled last layer WITHOUT its (R x N) output (see mlp_bwd_pool_kernel): N = 128, K = 64,
// ns in {32, 64}, R a multiple of ns, compute modes 1 (bf16) / 2 (three-term fp32).  dX (R x K) and dW
// (N x K) are written (not accumulated); g12_prev (2K doubles) is accumulated and - with gamma_prev - turned
// into layer L-1's backward vectors by the last workgroup; ``workspace`` holds demf_mlp_bwd_pool_ws(R)
// floats (scratch, no initialisation needed).
extern "C" int demf_mlp_bwd_pool(int R, int N, int K, int ns, const float* dP, const int* arg,
                                 const float* yraw, const float* vec6, const float* W, const float* Yprev,
                                 const float* scale_shift_prev, const float* mean_invstd_prev, float* dX,
                                 float* dW, double* g12_prev, const float* gamma_prev, float* vec6_prev,
                                 float* dgamma_prev, float* dbeta_prev, float* workspace,
                                 demf_stream_t stream) {
  const int cm = compute_mode();
  DEMF_REQUIRE((cm == 1 || cm == 2) && N == PB_N && K == PB_K && ns == PB_RS && R >= ns && R % ns == 0,
               "mlp_bwd_pool: unsupported shape / mode R=%d N=%d K=%d ns=%d mode=%d", R, N, K, ns, cm);
  DEMF_REQUIRE(dP && arg && yraw && vec6 && W && Yprev && scale_shift_prev && mean_invstd_prev && dX && dW &&
                   g12_prev && workspace, "mlp_bwd_pool: null pointer");
  PoolBwdArgs a{};
  a.R = R; a.ns = ns; a.Xp = Yprev; a.pss = scale_shift_prev; a.pmi = mean_invstd_prev; a.W = W; a.vec = vec6;
  a.dP = dP; a.arg = arg; a.yraw = yraw; a.dX = dX; a.part = workspace; a.g12 = g12_prev;
  { static const int dbg = [] { const char* v = getenv("DEMF_PB_DBG"); return v ? atoi(v) : 0; }(); a.dbg = dbg; }
  if (gamma_prev != nullptr) {
    DEMF_REQUIRE(vec6_prev && dgamma_prev && dbeta_prev, "mlp_bwd_pool: vectors of layer l-1 need all three outputs");
    a.vfin = BnVecFin{(double)R, gamma_prev, scale_shift_prev, mean_invstd_prev, vec6_prev, dgamma_prev,
                      dbeta_prev, sched_slot()};
  }
  hipStream_t s = (hipStream_t)stream;
  const int gx = pool_bwd_grid(R);
  const bool h2 = cm == 2 && f16_terms() && fused_h2_on();
  const int P = h2 ? 2 : (cm == 2 ? 3 : 1);
  const size_t head = sizeof(float) * (5 * PB_N + 4 * PB_K + PB_K + 32);                      // vectors (+ the H2 maxima)
  const size_t main = (size_t)(PB_N / 32) * P * 4096 + 2 * P * 4096 + (size_t)P * 8192 +       // planes
                      sizeof(float) * (2 * PB_RS * PB_LDA + PB_RS * PB_K);                     // 2 x fp32 rows + dA tile
  const size_t scratch = sizeof(float) * ((size_t)PB_N * PB_K + (size_t)PB_K * PB_LDM);      // prologue: W + M
  const size_t fold = sizeof(float) * (8 * 16 * 64 + PB_K * (PB_K + 1) + PB_K + PB_N * PB_K + 8 * 16 * 12);   // flush
  // (the prologue's W + M scratch and the flush's folds overlay the plane regions + the fp32 tiles)
  size_t body = main > scratch ? main : scratch;
  if (body < fold) body = fold;
  const size_t bytes = head + body;
  static bool configured[4] = {false, false, false, false};
  const int var = h2 ? 3 : cm;
  if (!configured[var]) {
    const void* fn = var == 1 ? reinterpret_cast<const void*>(&mlp_bwd_pool_kernel<1>)
                   : (var == 3 ? reinterpret_cast<const void*>(&mlp_bwd_pool_kernel<3>)
                               : reinterpret_cast<const void*>(&mlp_bwd_pool_kernel<2>));
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
      set_error("mlp_bwd_pool: cannot reserve %zu bytes of LDS", bytes);
      return DEMF_ELAUNCH;
    }
    configured[var] = true;
  }
  if (var == 1) hipLaunchKernelGGL(mlp_bwd_pool_kernel<1>, dim3(gx), dim3(512), bytes, s, a);
  else if (var == 3) hipLaunchKernelGGL(mlp_bwd_pool_kernel<3>, dim3(gx), dim3(512), bytes, s, a);
  else hipLaunchKernelGGL(mlp_bwd_pool_kernel<2>, dim3(gx), dim3(512), bytes, s, a);
  if (int e = check_launch("mlp_bwd_pool")) return e;
  hipLaunchKernelGGL(pool_bwd_finish_k, dim3((PB_NACC + 31) / 32), dim3(256), 0, s, gx, workspace, dW);
  if (int e = check_launch("pool_bwd_finish")) return e;
  return vectors_without_ticket(a.vfin, K, g12_prev, stream);
}

// SA1's second layer (64 -> 64, dense upstream gradient G) with the FIRST epilogue, layer 0's raw output NOT
// stored: its 4-float input rows X0 and weight W0 (64 x 4) rebuild what demf_mlp_bwd_fused reads from Yprev.
// Otherwise as demf_mlp_bwd_fused with first_sums != NULL (dW accumulated, first_sums as demf_mlp_first_finish
// reads them).  Compute modes 1 / 2.
extern "C" int demf_mlp_bwd_fused_x4(int R, int N, int K, const float* G, const float* Y, const float* vec6,
                                     const float* W, const float* X0, const float* W0,
                                     const float* scale_shift_prev, const float* mean_invstd_prev, float* dW,
                                     double* first_sums, demf_stream_t stream) {
  DEMF_REQUIRE(fused_supported(R, N, K, 1, 0, 1), "mlp_bwd_fused_x4: unsupported shape / mode R=%d N=%d K=%d mode=%d",
               R, N, K, compute_mode());
  DEMF_REQUIRE(G && Y && vec6 && W && X0 && W0 && scale_shift_prev && mean_invstd_prev && dW && first_sums,
               "mlp_bwd_fused_x4: null pointer");
  FusedBwdArgs a{};
  { static const int dbg = getenv("DEMF_BWD_DBG") ? atoi(getenv("DEMF_BWD_DBG")) : 0; a.dbg = dbg; }
  a.R = R; a.N = N; a.K = K; a.ns = 1; a.Yl = Y; a.G = G; a.vec = vec6; a.Xp = nullptr; a.pss = scale_shift_prev;
  a.pmi = mean_invstd_prev; a.W = W; a.dW = dW; a.fX = X0; a.fsum = first_sums; a.W0 = W0;
  a.ldk = K; a.ldw = K; a.fld = K; a.fc0 = 0;
  hipStream_t s = (hipStream_t)stream;
  if (compute_mode() == 1) return launch_fused<2, 2, 2, false, 1, 1, 4>(a, s);
  return launch_fused<2, 2, 2, false, 2, 1, 4>(a, s);
}
