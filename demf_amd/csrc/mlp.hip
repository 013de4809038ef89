// Fused shared-MLP kernels for gfx950: 1x1 conv (GEMM) + train-mode BatchNorm + ReLU (+ max
// over neighbours), forward and backward, on point-major rows.
//
// This is the dense half of every PointSAModule the reference builds
// (demf/modeling/heads/class_agnostic_vote_head.py:383 via build_sa_module; backbone
// configs/demf/demf_votenet.py:48-62): upstream runs Conv2d -> BN2d -> ReLU as separate
// kernels over (B,C,M,ns) and materialises every intermediate.  Here a layer is ONE pass:
//
//   forward   Y_l = act_{l-1}(Y_{l-1}) @ W_l^T          act(y) = max(0, y*scale + shift)
//             - the previous layer's BN+ReLU is applied in the A-operand prologue (A is never
//               materialised), per-channel sum / sum-of-squares of Y_l are reduced in the
//               epilogue (fp32 per block, fp64 atomics across blocks) -> batch statistics;
//   tail      out = max_s act_L(Y_L)  fused with the last BN+ReLU;
//   backward  dY_l = gamma*invstd*(dZ - mean(dZ) - xhat*mean(dZ*xhat)) is formed on the fly in
//             the prologue of  dA_{l-1} = dY_l @ W_l  and of  dW_l = dY_l^T @ A_{l-1}.
//
// All GEMMs are fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32, 157 TF peak) and at these
// skinny shapes (K,N <= 512, rows up to 1M) the kernels are HBM-bound: algorithmic bytes are
// rows*(K+N)*4 per forward layer.
//
// Fragment trick: for C = A(RxK) * Bt(NxK)^T with both operands K-contiguous, a lane loads
// ONE float4 along K for A (row = lane&31) and one per column tile for Bt (row = lane&31), at
// k = k0 + 4*(lane>>5) .. +3, and feeds component m to MFMA m: MFMA m then contracts over
// k in {k0+m, k0+4+m} on both operands consistently - four MFMAs consume the two float4.
#include "common.h"
#include <vector>
#include "bn_fin.h"
#include <atomic>
#include <type_traits>

namespace demf {

using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));

// Compute mode of the dense kernels (demf_set_compute_dtype):
//   0 = fp32 MFMA (v_mfma_f32_32x32x2_f32, the reference's precision);
//   1 = bf16 MFMA with fp32 accumulate (v_mfma_f32_32x32x16_bf16): operands are rounded to bf16 (RNE,
//       v_cvt_pk_bf16_f32) on their way into LDS, AFTER the fp32 prologue (BN + ReLU / BN-backward);
//       everything stored to memory, every statistic, index and loss stays fp32.  BASELINE configs[3];
//   2 = fp32 operands split into three bf16 terms each (x = h + m + l exactly: 3 x 8 significand
//       bits) and the six products h.h, h.m, m.h, m.m, h.l, l.h accumulated in fp32 on the bf16 MFMA
//       (the dropped m.l, l.m, l.l terms are <= 2^-23 relative - the rounding of one fp32 FMA).
//       On gfx950 the bf16 MFMA runs 16x the fp32 one, so six of them cost 3/8 of the native fp32
//       issue time at fp32-grade results (tests/test_gpu_split.py measures both against fp64).
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Until demf_set_compute_dtype is called the mode is 2, or 0 with DEMF_F32_NATIVE=1 in the environment.
static std::atomic<int> g_compute_mode{-1};
// demf_ctx_push / demf_ctx_pop: the calling THREAD's mode for the calls in between (a stack, so library code
// may nest them); -1 = the process default below.  Two host threads driving two models in different modes do
// not see each other's setting.
constexpr int CTX_DEPTH = 8;
static thread_local int tl_mode_stack[CTX_DEPTH];
static thread_local int tl_mode_top = 0;
int compute_mode() {
  if (tl_mode_top > 0 && tl_mode_stack[tl_mode_top - 1] >= 0) return tl_mode_stack[tl_mode_top - 1];
  int m = g_compute_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* v = getenv("DEMF_F32_NATIVE");
    m = (v && atoi(v)) ? 0 : 2;
    g_compute_mode.store(m);
  }
  return m;
}
bool compute_bf16() { return compute_mode() == 1; }
// (demf_set_f16_terms; until it is called: on, or off with DEMF_F16_TERMS=0 in the environment)
static std::atomic<int> g_f16_terms{-1};
bool f16_terms() {
  int on = g_f16_terms.load(std::memory_order_relaxed);
  if (on < 0) {
    const char* v = getenv("DEMF_F16_TERMS");
    on = v ? (atoi(v) != 0) : 1;
    g_f16_terms.store(on);
  }
  return on && compute_mode() == 2;
}

__device__ __forceinline__ bf16x4 to_bf16x4(const float4& v) {
  bf16x4 r;
  r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
  return r;
}

// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); both differences are exact in
// fp32 and the last one fits bf16's 8 bits, so the three terms carry all 24 significand bits.
__device__ __forceinline__ float4 bf16x4_to_f32(const bf16x4& h) {
  return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ void split3(const float4& v, bf16x4& h, bf16x4& m, bf16x4& l) {
  h = to_bf16x4(v);
  const float4 hf = bf16x4_to_f32(h);
  const float4 r = make_float4(v.x - hf.x, v.y - hf.y, v.z - hf.z, v.w - hf.w);
  m = to_bf16x4(r);
  const float4 mf = bf16x4_to_f32(m);
  l = to_bf16x4(make_float4(r.x - mf.x, r.y - mf.y, r.z - mf.z, r.w - mf.w));
}
__device__ __forceinline__ void split3(const float4& v0, const float4& v1, bf16x8& h, bf16x8& m, bf16x8& l) {
  bf16x4 h0, m0, l0, h1, m1, l1;
  split3(v0, h0, m0, l0);
  split3(v1, h1, m1, l1);
  h = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
  m = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
  l = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}

#ifdef DEMF_MLP_PROFILE   // tools/mlp_phase_prof.py: per-phase shader cycles of block (0,0), thread 0
__device__ long long g_mlp_prof[16];
#define MP_T(k) const long long mp##k = __builtin_readcyclecounter();
#define MP_ACC(i, a, b) if (mp_on) mp_acc[i] += (b) - (a);
#else
#define MP_T(k)
#define MP_ACC(i, a, b)
#endif
constexpr int MLP_BK = 32;        // K step staged per iteration
constexpr int SCHED_GROUPS = 16;  // counters per dynamically scheduled persistent launch
constexpr int DW_CHUNK = 16;      // 32-row slabs per claim of the weight-gradient kernel
constexpr int DW_MAX_SUB = 32;    // blockIdx.y columns with their own counter
constexpr int SCHED_INTS = 2 * DW_MAX_SUB;   // ints per counter set (>= 2*SCHED_GROUPS + 1)
constexpr int MLP_LD = MLP_BK + 4;  // LDS row stride (floats), +16 B pad
constexpr int MLP_LD3 = 3 * (MLP_BK / 2) + 4;   // Bt row stride of compute mode 2: three bf16 terms

enum { PRO_NONE = 0, PRO_BNRELU = 1, PRO_DY_DENSE = 2, PRO_DY_SPARSE = 3 };

// per-channel vectors of the backward prologue ("vec5"), struct-of-arrays of length n each:
//   [0] scale  [1] shift  (z = y*scale + shift, ReLU mask)   [2] gi = gamma*invstd
//   [3] a = -gi*invstd*mean(dZ*xhat)   [4] b = -gi*mean(dZ) - a*mean     (dY = gi*dZ + a*y + b)
// (BnFin, bn_finalize_channel: csrc/bn_fin.h)
struct MlpArgs {
  int R, K, N;                 // rows, reduction length, output columns
  int ldx;                     // row stride of X (floats)
  int ldy;                     // row stride of Y (floats)
  const float* X;              // PRO_NONE/BNRELU: input rows; PRO_DY_*: Y_l (pre-BN output)
  const float* G;              // PRO_DY_DENSE: upstream gradient dA_l (R x K)
  const float* dP;             // PRO_DY_SPARSE: pooled gradient (R/ns x K)
  const int* arg;              //                and arg-max slot (R/ns x K)
  int ns;
  const float* vec;            // BNRELU: [scale|shift] (2K);  DY: 5 vectors of length K
  const float* Bt;             // (N x K) row-major; or, with ldb > 0, the (K x >=N) matrix whose
  int ldb;                     // TRANSPOSE is the B operand (row stride ldb): W itself for dX = dY.W
  float* Y;                    // (R x N) output
  double* stats;               // optional (2N): column sum, column sum of squares
  // optional fused pooling epilogue (forward, last layer): per group of ns consecutive rows and
  // column, the max / min of the raw output and the row offset where each is attained
  float* pmax;                 // (R/ns x N) or null
  float* pmin;
  int* amax;
  int* amin;
  int halves;                  // pooled launches: 2 = two column halves interleaved in a 1-D grid
  int st;                      // bf16 STORAGE of the rows (weight-resident forward only): bit 0 X, bit 1 Y
  // FIRST epilogue (backward of a stack whose first layer has a 4-float input and no input
  // gradient): the output tile IS the gradient of layer 0's activation; instead of storing it, the
  // raw sums of layer 0's whole backward are taken from it (see mlp_first_finish_k)
  const float* W0;             // weight-resident forward, ST bit 3: (K x 4) weight of the layer BELOW, whose raw
                               // output is recomputed from its 4-float input rows X instead of being loaded
  const float* fX;             // (R x 4) input rows of layer 0
  const float* fY;             // (R x N) pre-BN output of layer 0
  const float* fss;            // layer 0 [scale|shift] (2N)
  const float* fmi;            // layer 0 [mean|invstd] (2N)
  double* fsum;                // g1(N) | g2(N) | P(N x 4) | Q(N x 4) | cx(4), accumulated
  // RED epilogue (dx GEMM of layer l): besides storing dX = the gradient of layer l-1's activation,
  // layer l-1's BN-backward sums (sum dZ, sum dZ*xhat) are taken from the tile -> stats; the layer
  // l-1 operands are addressed with row stride / vector length fld and column offset fc0
  int fld, fc0;
  int* sched;                  // persistent launches: SCHED_GROUPS tile counters, 1 + SCHED_GROUPS exit counters, or null
  BnFin fin;                   // STATS launches: in-kernel finalize when fin.ss != null
  BnVecFin vfin;               // RED launches: layer l-1's backward vectors by the last workgroup (ticket != null)
};

// Raw operands of one float4 of A: fetched early (kept in flight across the MFMA phase of the
// previous step) and only transformed when they are written to LDS.
template <int PRO>
struct MlpRaw {
  float4 x;                                                    // X / Y_l
  float4 g;                                                    // DY_DENSE: G ; DY_SPARSE: dP
  int4 a;                                                      // DY_SPARSE: arg
  int s;                                                       // DY_SPARSE: row's slot in its group
};
template <> struct MlpRaw<PRO_NONE> { float4 x; };
template <> struct MlpRaw<PRO_BNRELU> { float4 x; };
template <> struct MlpRaw<PRO_DY_DENSE> { float4 x, g; };

template <int PRO>
__device__ __forceinline__ void mlp_fetch(const MlpArgs& p, int row, int col, bool ok, MlpRaw<PRO>& r) {
  r.x = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PRO == PRO_DY_DENSE) r.g = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PRO == PRO_DY_SPARSE) {
    r.g = make_float4(0.f, 0.f, 0.f, 0.f);
    r.a = make_int4(-1, -1, -1, -1);
    r.s = 0;
  }
  if (!ok) return;
  r.x = *reinterpret_cast<const float4*>(p.X + (size_t)row * p.ldx + col);
  if constexpr (PRO == PRO_DY_DENSE) r.g = *reinterpret_cast<const float4*>(p.G + (size_t)row * p.K + col);
  if constexpr (PRO == PRO_DY_SPARSE) {
    const int rp = row / p.ns;
    r.s = row - rp * p.ns;
    r.g = *reinterpret_cast<const float4*>(p.dP + (size_t)rp * p.K + col);
    r.a = *reinterpret_cast<const int4*>(p.arg + (size_t)rp * p.K + col);
  }
}

// `vec` = the per-channel vectors (global, or the block's LDS copy); `ok` false -> zeros
template <int PRO>
__device__ __forceinline__ float4 mlp_xform(const MlpArgs& p, const float* __restrict__ vec, int col,
                                            bool ok, const MlpRaw<PRO>& r) {
  const float4 x = r.x;
  if constexpr (PRO == PRO_NONE) {
    return x;
  } else if constexpr (PRO == PRO_BNRELU) {
    if (!ok) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 s = *reinterpret_cast<const float4*>(vec + col);
    const float4 t = *reinterpret_cast<const float4*>(vec + p.K + col);
    float4 o;
    o.x = fmaxf(0.f, __builtin_fmaf(x.x, s.x, t.x));
    o.y = fmaxf(0.f, __builtin_fmaf(x.y, s.y, t.y));
    o.z = fmaxf(0.f, __builtin_fmaf(x.z, s.z, t.z));
    o.w = fmaxf(0.f, __builtin_fmaf(x.w, s.w, t.w));
    return o;
  } else {
    if (!ok) return make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g = r.g;
    if constexpr (PRO == PRO_DY_SPARSE) {
      g.x = r.a.x == r.s ? g.x : 0.f;
      g.y = r.a.y == r.s ? g.y : 0.f;
      g.z = r.a.z == r.s ? g.z : 0.f;
      g.w = r.a.w == r.s ? g.w : 0.f;
    }
    // dY = gi*(dZ - c1 - xhat*c2) = gi*dZ + (a*y + b), dZ = g where y*scale+shift > 0
    const int K = p.K;
    const float4 sc = *reinterpret_cast<const float4*>(vec + col);
    const float4 sh = *reinterpret_cast<const float4*>(vec + K + col);
    const float4 gi = *reinterpret_cast<const float4*>(vec + 2 * K + col);
    const float4 va = *reinterpret_cast<const float4*>(vec + 3 * K + col);
    const float4 vb = *reinterpret_cast<const float4*>(vec + 4 * K + col);
    float4 o;
#define MLP_DY(m)                                                                   \
    {                                                                               \
      const float dz = __builtin_fmaf(x.m, sc.m, sh.m) > 0.f ? g.m : 0.f;           \
      o.m = __builtin_fmaf(gi.m, dz, __builtin_fmaf(va.m, x.m, vb.m));              \
    }
    MLP_DY(x) MLP_DY(y) MLP_DY(z) MLP_DY(w)
#undef MLP_DY
    return o;
  }
}

template <int PRO>
__device__ __forceinline__ float4 mlp_load_a(const MlpArgs& p, const float* __restrict__ vec,
                                             int row, int col) {
  MlpRaw<PRO> r;
  mlp_fetch<PRO>(p, row, col, true, r);
  return mlp_xform<PRO>(p, vec, col, true, r);
}

// C(R x N) = pro(A)(R x K) @ Bt(N x K)^T ; NT = ceil(N/32) column tiles per wave (all of N).
// The (tile, k-step) sequence of a block is flattened so the next 32 x BK slab of A (per wave,
// prologue applied on arrival) and the next N x BK slab of Bt (shared by the 4 waves) are already
// in flight while the current ones feed the MFMAs.  A lives in a wave-private LDS region; Bt is
// staged once per step for the whole block as full 128-byte rows (fragment-shaped loads straight
// from global would cost 32 cache lines per wave-instruction).
// RT = 32-row tiles per wave (2 when the accumulators fit: twice the MFMA work per Bt fragment
// and per barrier).  The prologue's per-channel vectors are copied to LDS once per block.
constexpr int MLP_MAXK = 512;
// Fused max-pool epilogue.  max_s relu(sc*y_s + sh) = relu(sc*y* + sh) with y* = max_s y_s when
// sc >= 0 and min_s y_s otherwise (sc*y + sh is monotone in y and so is its rounding), so the GEMM can
// reduce the RAW output over each group of NS rows before the batch statistics exist; the
// consumer picks max or min once the scale is known (pool_select_k).  C/D layout: a lane owns one
// column and the rows (r&3) + 8*(r>>2) + 4*(lane>>5) of each 32-row tile; the partner lane^32 owns
// the other 16 rows.  Ties resolve to the smallest row offset (first maximum, as max_pool2d).
template <int RT, int NT, int NS>
__device__ __forceinline__ void pool_epilogue(const MlpArgs& p, const f32x16 (&acc)[RT][NT], int nt,
                                              int row0, int col, int lh) {
  constexpr int GROUPS = (32 * RT) / NS;           // groups inside this wave's rows
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    float mx = -__builtin_inff(), mn = __builtin_inff();
    int ax = 0, an = 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho_w = rt * 32 + (r & 3) + 8 * (r >> 2);      // + 4*lh, added below
        if (rho_w / NS != g) continue;                            // compile-time after unrolling
        const int rho = rho_w % NS + 4 * lh;
        const float v = acc[rt][nt][r];
        // a lane visits its rows in ascending order, so "first extremum wins" is the strict compare
        // (branch-free: one v_cmp + two v_cndmask each)
        const bool up = v > mx, dn = v < mn;
        mx = up ? v : mx; ax = up ? rho : ax;
        mn = dn ? v : mn; an = dn ? rho : an;
      }
    const float omx = __shfl_xor(mx, 32), omn = __shfl_xor(mn, 32);
    const int oax = __shfl_xor(ax, 32), oan = __shfl_xor(an, 32);
    if (omx > mx || (omx == mx && oax < ax)) { mx = omx; ax = oax; }
    if (omn < mn || (omn == mn && oan < an)) { mn = omn; an = oan; }
    const int row = row0 + g * NS;
    if (lh == 0 && row < p.R && col < p.N) {
      const size_t o = (size_t)(row / NS) * p.N + col;
      p.pmax[o] = mx; p.pmin[o] = mn; p.amax[o] = ax; p.amin[o] = an;
    }
  }
}

// NS = 64 with 32-row wave tiles: a group spans the row tiles of two neighbouring waves (2w, 2w+1).
// Each wave reduces its 32 rows, both publish through LDS, the even wave merges (its offsets are the
// smaller ones, so it wins ties) and writes.  s_pool: [wave][nt][lane&31] x {max, min, amax, amin}.
template <int NT>
__device__ __forceinline__ void pool_half_reduce(const f32x16 (&acc)[1][NT], int nt, int wave, int lr,
                                                 int lh, float4* __restrict__ s_pool) {
  float mx = -__builtin_inff(), mn = __builtin_inff();
  int ax = 0, an = 0;
  const int base = (wave & 1) * 32 + 4 * lh;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rho = base + (r & 3) + 8 * (r >> 2);
    const float v = acc[0][nt][r];
    const bool up = v > mx, dn = v < mn;          // rows ascend: the strict compare keeps the first
    mx = up ? v : mx; ax = up ? rho : ax;
    mn = dn ? v : mn; an = dn ? rho : an;
  }
  const float omx = __shfl_xor(mx, 32), omn = __shfl_xor(mn, 32);
  const int oax = __shfl_xor(ax, 32), oan = __shfl_xor(an, 32);
  if (omx > mx || (omx == mx && oax < ax)) { mx = omx; ax = oax; }
  if (omn < mn || (omn == mn && oan < an)) { mn = omn; an = oan; }
  if (lh == 0)
    s_pool[(wave * NT + nt) * 32 + lr] =
        make_float4(mx, mn, __builtin_bit_cast(float, ax), __builtin_bit_cast(float, an));
}

// SEL: the sign of the BN scale is the sign of gamma, which IS known before the statistics: only the
// extremum the consumer will pick is tracked - the maximum of v for gamma >= 0, of -v otherwise
// (``flip`` = the lane's sign-bit mask) - at half the compare / select work and half the pooled stores.
__device__ __forceinline__ float flip_sign(float v, unsigned flip) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ flip);
}

template <int RT, int NT, int NS>
__device__ __forceinline__ void pool_epilogue_sel(const MlpArgs& p, const f32x16 (&acc)[RT][NT], int nt,
                                                  int row0, int col, int lh, unsigned flip) {
  constexpr int GROUPS = (32 * RT) / NS;
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    float mx = -__builtin_inff();
    int ax = 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho_w = rt * 32 + (r & 3) + 8 * (r >> 2);
        if (rho_w / NS != g) continue;
        const int rho = rho_w % NS + 4 * lh;
        const float v = flip_sign(acc[rt][nt][r], flip);
        const bool up = v > mx;
        mx = up ? v : mx; ax = up ? rho : ax;
      }
    const float omx = __shfl_xor(mx, 32);
    const int oax = __shfl_xor(ax, 32);
    if (omx > mx || (omx == mx && oax < ax)) { mx = omx; ax = oax; }
    const int row = row0 + g * NS;
    if (lh == 0 && row < p.R && col < p.N) {
      const size_t o = (size_t)(row / NS) * p.N + col;
      p.pmax[o] = flip_sign(mx, flip); p.amax[o] = ax;
    }
  }
}

template <int NT>
__device__ __forceinline__ void pool_half_reduce_sel(const f32x16 (&acc)[1][NT], int nt, int wave, int lr,
                                                     int lh, float4* __restrict__ s_pool, unsigned flip) {
  float mx = -__builtin_inff();
  int ax = 0;
  const int base = (wave & 1) * 32 + 4 * lh;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rho = base + (r & 3) + 8 * (r >> 2);
    const float v = flip_sign(acc[0][nt][r], flip);
    const bool up = v > mx;
    mx = up ? v : mx; ax = up ? rho : ax;
  }
  const float omx = __shfl_xor(mx, 32);
  const int oax = __shfl_xor(ax, 32);
  if (omx > mx || (omx == mx && oax < ax)) { mx = omx; ax = oax; }
  if (lh == 0) {
    float2* sp = reinterpret_cast<float2*>(s_pool);
    sp[(wave * NT + nt) * 32 + lr] = make_float2(mx, __builtin_bit_cast(float, ax));
  }
}

// CM = 1 (bf16): the A / B slabs hold bf16 (same byte row stride as the fp32 layout, so the slab
// doubles as fp32 staging for the FIRST / RED epilogues unchanged) and a K step of 32 is two
// v_mfma_f32_32x32x16_bf16 per tile instead of sixteen v_mfma_f32_32x32x2_f32.
// CM = 2 (three-term split): the A slab stays fp32 - each wave splits the 8 floats of its own
// fragment in registers after the LDS read (its rows are private to it, so nothing is split twice) -
// and the Bt slab, which all four waves share, is split once on its way into LDS: row stride
// MLP_LD3 floats = [32 h | 32 m | 32 l] bf16 + 16 B pad (52 dwords: 8 consecutive rows still cover
// the 32 banks with their 16-byte reads).  Six bf16 MFMAs per tile and 16 columns of K.
template <int NT, int RT, int PRO, bool STATS, bool POOL = false, bool FIRST = false, bool RED = false,
          int CM = 0, bool SEL = false>
__global__ __launch_bounds__(256, 2) void mlp_gemm_kernel(MlpArgs p) {
  static_assert(!SEL || POOL, "SEL is a mode of the pooled epilogue");
  static_assert(!(STATS && RED) && !(FIRST && RED), "one column-sum epilogue at a time");
  constexpr bool BF16 = CM == 1, X3 = CM == 2;
  constexpr int LDB = X3 ? MLP_LD3 : MLP_LD;
  constexpr int WROWS = 32 * RT;          // rows per wave
  constexpr int BROWS = 4 * WROWS;        // rows per block tile
  constexpr int NVEC = PRO == PRO_NONE ? 0 : (PRO == PRO_BNRELU ? 2 : 5);
  __shared__ __attribute__((aligned(16))) float s_a[4][WROWS * MLP_LD];
  __shared__ __attribute__((aligned(16))) float s_b[NT * 32 * LDB];
  __shared__ __attribute__((aligned(16))) float s_vec[NVEC ? NVEC * MLP_MAXK : 4];
  __shared__ float s_red[(STATS || RED) ? 4 * NT * 32 * 2 : (FIRST ? 4 * NT * 32 * 10 + 16 : 1)];
  __shared__ float4 s_pool[(POOL && RT == 1) ? 4 * NT * 32 : 1];
  if constexpr (NVEC > 0) {
    // compact copy: vector v of length K lives at s_vec + v*K (same addressing as global)
    for (int i = threadIdx.x; i < NVEC * p.K; i += 256) s_vec[i] = p.vec[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  float* sa = s_a[wave];
  // few-row launches split the output columns over blockIdx.y so the chip is still filled
  // Pooled launches with two column halves use a 1-D grid in which blocks b and b+8 - the same
  // XCD under round-robin dispatch - take the two halves of the same row tiles at the same time,
  // so the second read of the A rows is an L2 hit instead of a second trip to HBM.
  int bx = blockIdx.x, gx = gridDim.x, by = blockIdx.y;
  if constexpr (POOL) {
    if (p.halves == 2) {
      by = (bx >> 3) & 1;
      bx = (bx & 7) | ((bx >> 4) << 3);
      gx >>= 1;
    }
  }
  const int cofs = by * NT * 32;
  const int ntiles = (p.R + BROWS - 1) / BROWS;
  const int ksteps = (p.K + MLP_BK - 1) / MLP_BK;
  // Persistent launches take their first tile by block index and every further one from a device
  // counter, claimed one tile ahead (the atomic's latency hides behind a whole tile): blocks that
  // run slower - e.g. while the furthest-point chain of the next batch is resident on the chip -
  // simply take fewer tiles instead of holding the launch back.  sched == null: static striding.
  // Same-address atomics serialise (~60 ns each), so the blocks are split into SCHED_GROUPS
  // groups that interleave over the XCDs and over the tile range in runs of 8; a group owns every
  // SCHED_GROUPS-th run of 8 tiles and has its own counter (32 claimants instead of 512).
  __shared__ int s_next;
  const bool dyn = p.sched != nullptr;
  const int grp = (bx >> 3) & (SCHED_GROUPS - 1);
  const int grp_first = gx / SCHED_GROUPS;          // tiles per group taken by block index
  auto group_tile = [&](int j) { return (((j >> 3) * SCHED_GROUPS + grp) << 3) + (j & 7); };
  int claimed = 0;
  float cs1[NT], cs2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) cs1[nt] = cs2[nt] = 0.f;
  float fs[FIRST ? NT : 1][10];
  float4 fcx = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (FIRST) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 10; ++q) fs[nt][q] = 0.f;
  }
  f32x16 acc[RT][NT];
  unsigned selbits = 0;                            // SEL: bit nt = this lane's column of tile nt has gamma < 0
  if constexpr (SEL) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = cofs + nt * 32 + lr;
      if (col < p.N && p.fin.gamma[col] < 0.f) selbits |= 1u << nt;
    }
  }

  const int pr = lane >> 3, pc = (lane & 7) * 4;   // this lane's slot in a 32 x BK slab
  const int br = threadIdx.x >> 3;                  // Bt slab: row br + 32*i, cols pc..pc+3
  MlpRaw<PRO> pre[4 * RT];
  bool pok[4 * RT];
  float4 preb[NT];
  auto prefetch = [&](int tile, int ks) {
    const int k0 = ks * MLP_BK;
    const int row0 = tile * BROWS + wave * WROWS;
#pragma unroll
    for (int it = 0; it < 4 * RT; ++it) {
      const int row = row0 + it * 8 + pr, col = k0 + pc;
      pok[it] = row < p.R && col < p.K;
      mlp_fetch<PRO>(p, row, col, pok[it], pre[it]);   // raw: stays in flight until the LDS write
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = cofs + br + 32 * i, col = k0 + pc;
      preb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.ldb > 0) {
        // transposed source: this thread's float4 runs along the OUTPUT columns of reduction row br
        const int kk = k0 + br, nn = cofs + 32 * i + pc;
        if (kk < p.K && nn < p.N) preb[i] = *reinterpret_cast<const float4*>(p.Bt + (size_t)kk * p.ldb + nn);
      } else if (n < p.N && col < p.K) {
        preb[i] = *reinterpret_cast<const float4*>(p.Bt + (size_t)n * p.K + col);
      }
    }
  };
#ifdef DEMF_MLP_PROFILE
  long long mp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool mp_on = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
#endif
  int tile = bx;
  if (tile < ntiles) prefetch(tile, 0);

  int ks = 0;
  while (tile < ntiles) {
    MP_T(0)
    const int k0 = ks * MLP_BK;
    const int row0 = tile * BROWS + wave * WROWS;
    const bool last_ks = ks == ksteps - 1;
    if (dyn && threadIdx.x == 0) {
      if (ks == 0) claimed = group_tile(grp_first + atomicAdd(p.sched + grp, 1));
      if (last_ks) s_next = claimed;
    }
    float* sb = s_b;
    if (ks == 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][nt][r] = 0.f;
    }
    lds_barrier();                                    // everyone is done reading the old Bt slab (LDS-only
                                                      // wait: the previous tile's output stores stay in flight)
    MP_T(1) MP_ACC(0, mp0, mp1)
    if constexpr (BF16) {
      // bf16 element (row, k) lives at byte row * 4*MLP_LD + 2*k of the slab
      char* sab = reinterpret_cast<char*>(sa);
      char* sbb = reinterpret_cast<char*>(sb);
#pragma unroll
      for (int it = 0; it < 4 * RT; ++it)
        *reinterpret_cast<bf16x4*>(sab + (it * 8 + pr) * (4 * MLP_LD) + 2 * pc) =
            to_bf16x4(mlp_xform<PRO>(p, s_vec, k0 + pc, pok[it], pre[it]));
#pragma unroll
      for (int i = 0; i < NT; ++i)
        if (p.ldb > 0) {
          char* d = sbb + (32 * i + pc) * (4 * MLP_LD) + 2 * br;
          *reinterpret_cast<__bf16*>(d) = (__bf16)preb[i].x;
          *reinterpret_cast<__bf16*>(d + 4 * MLP_LD) = (__bf16)preb[i].y;
          *reinterpret_cast<__bf16*>(d + 8 * MLP_LD) = (__bf16)preb[i].z;
          *reinterpret_cast<__bf16*>(d + 12 * MLP_LD) = (__bf16)preb[i].w;
        } else {
          *reinterpret_cast<bf16x4*>(sbb + (br + 32 * i) * (4 * MLP_LD) + 2 * pc) = to_bf16x4(preb[i]);
        }
    } else if constexpr (X3) {
#pragma unroll
      for (int it = 0; it < 4 * RT; ++it)
        *reinterpret_cast<float4*>(sa + (it * 8 + pr) * MLP_LD + pc) =
            mlp_xform<PRO>(p, s_vec, k0 + pc, pok[it], pre[it]);
      // term t of element (n, k) lives at byte n * 4*MLP_LD3 + 64*t + 2*k of the Bt slab
      char* sbb = reinterpret_cast<char*>(sb);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        bf16x4 h, m, l;
        split3(preb[i], h, m, l);
        if (p.ldb > 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            char* d = sbb + (32 * i + pc + e) * (4 * MLP_LD3) + 2 * br;
            *reinterpret_cast<__bf16*>(d) = h[e];
            *reinterpret_cast<__bf16*>(d + 64) = m[e];
            *reinterpret_cast<__bf16*>(d + 128) = l[e];
          }
        } else {
          char* d = sbb + (br + 32 * i) * (4 * MLP_LD3) + 2 * pc;
          *reinterpret_cast<bf16x4*>(d) = h;
          *reinterpret_cast<bf16x4*>(d + 64) = m;
          *reinterpret_cast<bf16x4*>(d + 128) = l;
        }
      }
    } else {
#pragma unroll
    for (int it = 0; it < 4 * RT; ++it)
      *reinterpret_cast<float4*>(sa + (it * 8 + pr) * MLP_LD + pc) =
          mlp_xform<PRO>(p, s_vec, k0 + pc, pok[it], pre[it]);
#pragma unroll
    for (int i = 0; i < NT; ++i)
      if (p.ldb > 0) {
        float* d = sb + (32 * i + pc) * MLP_LD + br;
        d[0] = preb[i].x; d[MLP_LD] = preb[i].y; d[2 * MLP_LD] = preb[i].z; d[3 * MLP_LD] = preb[i].w;
      } else {
        *reinterpret_cast<float4*>(sb + (br + 32 * i) * MLP_LD + pc) = preb[i];
      }
    }
    MP_T(2) MP_ACC(1, mp1, mp2)
    lds_barrier();
    MP_T(3) MP_ACC(2, mp2, mp3)
    // the pooling epilogue needs registers: on the last k-step of a tile the next tile's
    // operands are fetched after it instead of being held in flight across it
    constexpr bool DEFER = (POOL && RT == 2) || FIRST;   // epilogues that need the prefetch registers
    const bool defer_prefetch = DEFER && last_ks;
    const int next_tile = last_ks ? (dyn ? s_next : tile + gx) : tile;
    const int next_ks = last_ks ? 0 : ks + 1;
    if (next_tile < ntiles && !defer_prefetch) prefetch(next_tile, next_ks);
    MP_T(4) MP_ACC(3, mp3, mp4)
    // FIRST: the wave's 64 x 32 half tile of layer 0's output (coalesced float4 rows) and its input
    // rows are fetched before the MFMAs of the last k-step and staged through the wave's A slab
    float4 yq[(FIRST || RED) ? 2 * RT * 2 : 1];
    float4 xq = make_float4(0.f, 0.f, 0.f, 0.f);
    auto first_load = [&](int half) {
#pragma unroll
      for (int j = 0; j < 2 * RT * 2; ++j) {
        const int id = lane + 64 * j;
        const int row = row0 + (id >> 3);
        yq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < p.R && cofs + 32 * half + 4 * (id & 7) < p.N)
          yq[j] = *reinterpret_cast<const float4*>(p.fY + (size_t)row * p.fld + p.fc0 + cofs + 32 * half + 4 * (id & 7));
      }
    };
    auto first_store = [&]() {
#pragma unroll
      for (int j = 0; j < 2 * RT * 2; ++j) {
        const int id = lane + 64 * j;
        *reinterpret_cast<float4*>(sa + (id >> 3) * MLP_LD + 4 * (id & 7)) = yq[j];
      }
    };
    if constexpr (RED) {
      if (last_ks) first_load(0);
    }
    if constexpr (FIRST) {
      if (last_ks) {
        first_load(0);
        if (lane < WROWS && row0 + lane < p.R)
          xq = *reinterpret_cast<const float4*>(p.fX + (size_t)(row0 + lane) * 4);
      }
    }
    if constexpr (BF16) {
      const char* sab = reinterpret_cast<const char*>(sa);
      const char* sbb = reinterpret_cast<const char*>(sb);
      const int kch = min(MLP_BK / 16, (p.K - k0 + 15) / 16);
      for (int c16 = 0; c16 < kch; ++c16) {
        bf16x8 a8[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          a8[rt] = *reinterpret_cast<const bf16x8*>(sab + (rt * 32 + lr) * (4 * MLP_LD) + 2 * (c16 * 16 + 8 * lh));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(sbb + (nt * 32 + lr) * (4 * MLP_LD) + 2 * (c16 * 16 + 8 * lh));
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[rt], b8, acc[rt][nt], 0, 0, 0);
        }
      }
    } else if constexpr (X3) {
      const char* sbb = reinterpret_cast<const char*>(sb);
      const int kch = min(MLP_BK / 16, (p.K - k0 + 15) / 16);
      for (int c16 = 0; c16 < kch; ++c16) {
        bf16x8 ah[RT], am[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float* src = sa + (rt * 32 + lr) * MLP_LD + c16 * 16 + 8 * lh;
          split3(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4),
                 ah[rt], am[rt], al[rt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const char* src = sbb + (nt * 32 + lr) * (4 * MLP_LD3) + 2 * (c16 * 16 + 8 * lh);
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(src);
          const bf16x8 bm = *reinterpret_cast<const bf16x8*>(src + 64);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(src + 128);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            // small terms first
            acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rt], bh, acc[rt][nt], 0, 0, 0);
            acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bl, acc[rt][nt], 0, 0, 0);
            acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[rt], bm, acc[rt][nt], 0, 0, 0);
            acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[rt], bh, acc[rt][nt], 0, 0, 0);
            acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bm, acc[rt][nt], 0, 0, 0);
            acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bh, acc[rt][nt], 0, 0, 0);
          }
        }
      }
    } else {
    const int kchunks = min(MLP_BK / 8, (p.K - k0 + 7) / 8);
    for (int c8 = 0; c8 < kchunks; ++c8) {
      float4 a4[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        a4[rt] = *reinterpret_cast<const float4*>(sa + (rt * 32 + lr) * MLP_LD + c8 * 8 + 4 * lh);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 b4 = *reinterpret_cast<const float4*>(sb + (nt * 32 + lr) * MLP_LD + c8 * 8 + 4 * lh);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rt].x, b4.x, acc[rt][nt], 0, 0, 0);
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rt].y, b4.y, acc[rt][nt], 0, 0, 0);
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rt].z, b4.z, acc[rt][nt], 0, 0, 0);
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rt].w, b4.w, acc[rt][nt], 0, 0, 0);
        }
      }
    }
    }
    MP_T(5) MP_ACC(4, mp4, mp5)
    if (last_ks) {
      // epilogue: C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float s1 = 0.f, s2 = 0.f;
        if constexpr (FIRST) {
          const int col = cofs + nt * 32 + lr;
          const bool cok = col < p.N;
          const float sc0 = cok ? p.fss[p.fc0 + col] : 0.f, sh0 = cok ? p.fss[p.fld + p.fc0 + col] : 0.f;
          const float mu0 = cok ? p.fmi[p.fc0 + col] : 0.f, is0 = cok ? p.fmi[p.fld + p.fc0 + col] : 0.f;
          // stage this half of Y0 (and, once, the input rows in the 4 pad columns) in the A slab
          first_store();
          if (nt == 0 && lane < WROWS) *reinterpret_cast<float4*>(sa + lane * MLP_LD + MLP_BK) = xq;
          if (nt + 1 < NT) first_load(nt + 1);       // next half in flight during this one's sums
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rl = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads from piling up
              if (row0 + rl < p.R && cok) {
                const float y = sa[rl * MLP_LD + lr];
                const float4 x = *reinterpret_cast<const float4*>(sa + rl * MLP_LD + MLP_BK);
                const float dz = __builtin_fmaf(y, sc0, sh0) > 0.f ? acc[rt][nt][r] : 0.f;
                fs[nt][0] += dz;
                fs[nt][1] = __builtin_fmaf(dz, (y - mu0) * is0, fs[nt][1]);
                fs[nt][2] = __builtin_fmaf(dz, x.x, fs[nt][2]);
                fs[nt][3] = __builtin_fmaf(dz, x.y, fs[nt][3]);
                fs[nt][4] = __builtin_fmaf(dz, x.z, fs[nt][4]);
                fs[nt][5] = __builtin_fmaf(dz, x.w, fs[nt][5]);
                fs[nt][6] = __builtin_fmaf(y, x.x, fs[nt][6]);
                fs[nt][7] = __builtin_fmaf(y, x.y, fs[nt][7]);
                fs[nt][8] = __builtin_fmaf(y, x.z, fs[nt][8]);
                fs[nt][9] = __builtin_fmaf(y, x.w, fs[nt][9]);
                if (nt == 0 && lr == 0 && by == 0) { fcx.x += x.x; fcx.y += x.y; fcx.z += x.z; fcx.w += x.w; }
              }
            }
        } else {
        float sc0 = 0.f, sh0 = 0.f, mu0 = 0.f, is0 = 0.f;
        if constexpr (RED) {
          const int col = cofs + nt * 32 + lr;
          if (col < p.N) {
            sc0 = p.fss[p.fc0 + col]; sh0 = p.fss[p.fld + p.fc0 + col];
            mu0 = p.fmi[p.fc0 + col]; is0 = p.fmi[p.fld + p.fc0 + col];
          }
          first_store();                             // this half of Y_{l-1} through the wave's A slab
          if (nt + 1 < NT) first_load(nt + 1);
        }
        // interior tiles of the pooled launches store unpredicated: a per-store bounds test costs an
        // exec-mask branch each; only the last row tile / a ragged column tile takes the checked path
        auto store_tile = [&](auto full_c) {
          constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              const float v = acc[rt][nt][r];
              const bool ok = FULL || (row < p.R && cofs + nt * 32 + lr < p.N);
              if (ok) p.Y[(size_t)row * p.ldy + cofs + nt * 32 + lr] = v;
              if constexpr (RED) {
                if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                if (ok) {
                  const float y = sa[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * MLP_LD + lr];
                  const float dz = __builtin_fmaf(y, sc0, sh0) > 0.f ? v : 0.f;
                  s1 += dz;
                  s2 = __builtin_fmaf(dz, (y - mu0) * is0, s2);
                }
              }
            }
          if constexpr (STATS) {
            // column statistics on register pairs: v_pk_add_f32 / v_pk_fma_f32 take two accumulators
            // per instruction (rows >= R are exact zeros: their A rows are zero)
            using f32x2 = float __attribute__((ext_vector_type(2)));
            f32x2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                const f32x2 v = {acc[rt][nt][r], acc[rt][nt][r + 1]};
                a1 += v;
                a2 = __builtin_elementwise_fma(v, v, a2);
              }
            s1 += a1.x + a1.y;
            s2 += a2.x + a2.y;
          }
        };
        // (only in the pooled instantiations: duplicating the loop in the wide plain ones - up to
        // 128 accumulators live - made them 30-90 % slower)
        if constexpr (POOL) {
          if (tile * BROWS + BROWS <= p.R && cofs + NT * 32 <= p.N) store_tile(std::true_type{});
          else store_tile(std::false_type{});
        } else {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              const float v = acc[rt][nt][r];
              if (row < p.R && cofs + nt * 32 + lr < p.N)
                p.Y[(size_t)row * p.ldy + cofs + nt * 32 + lr] = v;
              if constexpr (RED) {
                if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                if (row < p.R && cofs + nt * 32 + lr < p.N) {
                  const float y = sa[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * MLP_LD + lr];
                  const float dz = __builtin_fmaf(y, sc0, sh0) > 0.f ? v : 0.f;
                  s1 += dz;
                  s2 = __builtin_fmaf(dz, (y - mu0) * is0, s2);
                }
              }
              if constexpr (STATS) {
                s1 += v;                   // rows >= R are exact zeros (their A rows are zero)
                s2 = __builtin_fmaf(v, v, s2);
              }
            }
        }
        }
        cs1[nt] += s1;
        cs2[nt] += s2;
        if constexpr (POOL) {
          {
            const int col = cofs + nt * 32 + lr;
            if constexpr (SEL) {
              const unsigned flip = ((selbits >> nt) & 1u) << 31;
              if (p.ns == 16) pool_epilogue_sel<RT, NT, 16>(p, acc, nt, row0, col, lh, flip);
              else if (p.ns == 32) pool_epilogue_sel<RT, NT, 32>(p, acc, nt, row0, col, lh, flip);
              else if constexpr (RT == 2) pool_epilogue_sel<RT, NT, 64>(p, acc, nt, row0, col, lh, flip);
              else pool_half_reduce_sel<NT>(acc, nt, wave, lr, lh, s_pool, flip);
            } else {
            if (p.ns == 16) pool_epilogue<RT, NT, 16>(p, acc, nt, row0, col, lh);
            else if (p.ns == 32) pool_epilogue<RT, NT, 32>(p, acc, nt, row0, col, lh);
            else if constexpr (RT == 2) pool_epilogue<RT, NT, 64>(p, acc, nt, row0, col, lh);
            else pool_half_reduce<NT>(acc, nt, wave, lr, lh, s_pool);
            }
          }
        }
      }
      if constexpr (POOL && RT == 1) {
        if (p.ns == 64) {                       // merge the two half groups of each wave pair
          lds_barrier();                        // (LDS-only: the tile's output stores keep draining)
          if ((wave & 1) == 0 && lh == 0) {
            const int grow = row0;              // first row of this wave pair's group
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int col = cofs + nt * 32 + lr;
              if constexpr (SEL) {
                const float2* sp = reinterpret_cast<const float2*>(s_pool);
                const float2 a = sp[(wave * NT + nt) * 32 + lr];
                const float2 b = sp[((wave + 1) * NT + nt) * 32 + lr];
                const bool take_b = b.x > a.x;          // the even wave's offsets are the smaller ones
                if (grow < p.R && col < p.N) {
                  const unsigned flip = ((selbits >> nt) & 1u) << 31;
                  const size_t o = (size_t)(grow / 64) * p.N + col;
                  p.pmax[o] = flip_sign(take_b ? b.x : a.x, flip);
                  p.amax[o] = __builtin_bit_cast(int, take_b ? b.y : a.y);
                }
                continue;
              }
              const float4 a = s_pool[(wave * NT + nt) * 32 + lr];
              const float4 b = s_pool[((wave + 1) * NT + nt) * 32 + lr];
              float mx = a.x, mn = a.y;
              int ax = __builtin_bit_cast(int, a.z), an = __builtin_bit_cast(int, a.w);
              if (b.x > mx) { mx = b.x; ax = __builtin_bit_cast(int, b.z); }
              if (b.y < mn) { mn = b.y; an = __builtin_bit_cast(int, b.w); }
              if (grow < p.R && col < p.N) {
                const size_t o = (size_t)(grow / 64) * p.N + col;
                p.pmax[o] = mx; p.pmin[o] = mn; p.amax[o] = ax; p.amin[o] = an;
              }
            }
          }
        }
      }
      if constexpr ((POOL && RT == 2) || FIRST) {
        if (next_tile < ntiles) prefetch(next_tile, next_ks);
      }
    }
    MP_T(6) MP_ACC(5, mp5, mp6)
    tile = next_tile;
    ks = next_ks;
  }
#ifdef DEMF_MLP_PROFILE
  if (mp_on) { for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)&g_mlp_prof[i], (unsigned long long)mp_acc[i]); atomicAdd((unsigned long long*)&g_mlp_prof[15], 1ull); }
#endif
  if (dyn) {
    // the last block out re-arms the counter pair for the next launch that uses this slot
    if (threadIdx.x == 0) {
      // (no fence: only atomics touch the counters, and an agent-scope release here would write the
      // XCD's whole L2 back once per block)
      // two-level count (512 same-address atomics with a return value would hold the last block
      // back by ~30 us): group members first, then the 16 group-lasts
      int* gdone = p.sched + SCHED_GROUPS;
      const int members = (int)(gridDim.x * gridDim.y) / SCHED_GROUPS;
      if (atomicAdd(gdone + 1 + grp, 1) == members - 1) {
        atomicExch(gdone + 1 + grp, 0);
        atomicExch(p.sched + grp, 0);
        if (atomicAdd(gdone, 1) == SCHED_GROUPS - 1) atomicExch(gdone, 0);
      }
    }
  }
  if constexpr (FIRST) {
    // lanes l and l+32 hold the same column: fold, then the 4 waves through LDS, then fp64 atomics
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        const float v = fs[nt][q] + __shfl_xor(fs[nt][q], 32);
        if (lh == 0) s_red[((wave * NT + nt) * 10 + q) * 32 + lr] = v;
      }
    fcx.x += __shfl_xor(fcx.x, 32); fcx.y += __shfl_xor(fcx.y, 32);
    fcx.z += __shfl_xor(fcx.z, 32); fcx.w += __shfl_xor(fcx.w, 32);
    if (lane == 0) {
      float* c = s_red + 4 * NT * 32 * 10 + wave * 4;
      c[0] = fcx.x; c[1] = fcx.y; c[2] = fcx.z; c[3] = fcx.w;
    }
    __syncthreads();
    constexpr int PER = NT * 10 * 32;
    for (int i = threadIdx.x; i < PER; i += 256) {
      const float v = s_red[i] + s_red[PER + i] + s_red[2 * PER + i] + s_red[3 * PER + i];
      const int nt = i / 320, q = (i / 32) % 10, c = i & 31;
      const int col = cofs + nt * 32 + c;
      if (col < p.N) {
        // layout: g1 | g2 | P (N x 4) | Q (N x 4) | cx(4)
        const int dst = q < 2 ? q * p.N + col : (q < 6 ? 2 * p.N + col * 4 + (q - 2) : 6 * p.N + col * 4 + (q - 6));
        atomicAdd(p.fsum + dst, (double)v);
      }
    }
    if (threadIdx.x < 4 && by == 0) {
      const float* c = s_red + 4 * NT * 32 * 10;
      atomicAdd(p.fsum + 10 * p.N + threadIdx.x,
                (double)(c[threadIdx.x] + c[4 + threadIdx.x] + c[8 + threadIdx.x] + c[12 + threadIdx.x]));
    }
  }
  if constexpr (STATS || RED) {
    // lanes l and l+32 hold the same column: fold, then the 4 waves, then one fp64 atomic
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      cs1[nt] += __shfl_xor(cs1[nt], 32);
      cs2[nt] += __shfl_xor(cs2[nt], 32);
      if (lh == 0) {
        s_red[(wave * NT + nt) * 64 + lr] = cs1[nt];
        s_red[(wave * NT + nt) * 64 + 32 + lr] = cs2[nt];
      }
    }
    __syncthreads();
    // (the launch's last workgroup consumes the sums: they go to this workgroup's copy of the accumulator, bn_fin.h)
    const int lin_s = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    double* sdst = p.stats;
    if constexpr (RED) {
      if (p.vfin.ticket != nullptr && p.vfin.racc != nullptr) sdst = repl_copy(p.vfin.racc, p.fld, lin_s);
    } else {
      if (p.fin.ss != nullptr && p.fin.racc != nullptr) sdst = repl_copy(p.fin.racc, p.N, lin_s);
    }
    for (int i = threadIdx.x; i < NT * 64; i += 256) {
      const float v = s_red[i] + s_red[NT * 64 + i] + s_red[2 * NT * 64 + i] + s_red[3 * NT * 64 + i];
      const int nt = i >> 6, which = (i >> 5) & 1, c = i & 31;
      if (cofs + nt * 32 + c < p.N)
        atomicAdd(sdst + which * (RED ? p.fld : p.N) + (RED ? p.fc0 : 0) + cofs + nt * 32 + c, (double)v);
    }
  }
  if constexpr (RED) {
    if (p.vfin.ticket != nullptr) {
      // layer l-1's backward vectors for this launch's columns, by its last workgroup (csrc/bn_fin.h)
      sync_drained();
      if (threadIdx.x == 0)
        s_next = last_workgroup(p.vfin.ticket, (int)(gridDim.x * gridDim.y), (int)(blockIdx.y * gridDim.x + blockIdx.x));
      __syncthreads();
      if (s_next) bn_vec_finalize(p.vfin, p.fld, p.fc0, p.N, p.stats, threadIdx.x, 256);
    }
  }
  if constexpr (STATS) {
    if (p.fin.ss != nullptr) {
      // Only atomics touch the sums and the counters, and sync_drained() waits for every wave's own
      // to be acknowledged, so no fence is needed.  Two-level exit count as above.
      sync_drained();
      if (threadIdx.x == 0) {
        const int total = (int)(gridDim.x * gridDim.y);
        const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int ngroups = total < SCHED_GROUPS ? total : SCHED_GROUPS;
        const int g = lin % SCHED_GROUPS;
        const int members = total / SCHED_GROUPS + (g < total % SCHED_GROUPS ? 1 : 0);
        int* t = p.fin.ticket + FIN_OFF;
        int last = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own sum atomics acknowledged first (csrc/bn_fin.h)
        if (atomicAdd(t + 1 + g, 1) == members - 1) {
          atomicExch(t + 1 + g, 0);
          if (atomicAdd(t, 1) == ngroups - 1) { atomicExch(t, 0); last = 1; }
        }
        s_next = last;
      }
      __syncthreads();
      if (s_next) {
        const BnFin& f = p.fin;
        if (threadIdx.x == 0 && f.nbt != nullptr) *f.nbt += 1;
        for (int c = threadIdx.x; c < p.N; c += 256) {
          // read through the atomic unit (the sums were produced by device-scope atomics) and leave
          // the accumulator zeroed, in one exchange each
          double s1 = __builtin_bit_cast(
              double, atomicExch(reinterpret_cast<unsigned long long*>(p.stats + c), 0ull));
          double s2 = __builtin_bit_cast(
              double, atomicExch(reinterpret_cast<unsigned long long*>(p.stats + p.N + c), 0ull));
          if (f.racc != nullptr) { s1 += repl_take(f.racc, p.N, c); s2 += repl_take(f.racc, p.N, p.N + c); }
          bn_finalize_channel(c, p.N, f.count, s1, s2, f.gamma, f.beta, f.eps, f.momentum, f.rmean,
                              f.rvar, f.ss, f.mi, f.conv_bias);
        }
      }
    }
  }
}

// ---- forward GEMM with the WEIGHT RESIDENT in LDS and free-running waves ---------------------------
// mlp_gemm_kernel above restages the weight slab from L2 for every K step of every row tile and
// separates the steps with workgroup barriers: at SA1's last layer (R = 1 M rows, 64 -> 128) every
// 128-row tile costs 2 restagings + 4 barriers, the waves of a workgroup move in lockstep and the kernel
// sits at 0.50 of the HBM roof with its VALU, MFMA and memory phases in series.
// For layers whose whole weight fits LDS as split bf16 planes (N*K*2*P bytes: 48 KB for 128 x 64 in the
// three-term mode) it is staged ONCE per workgroup; after that a wave needs nobody: it walks its own
// 64-row units (two 32-row tiles), loads them as 4-channel register patches (full 256-byte rows), applies
// the previous layer's BN + ReLU, splits the result once into bf16 planes in its PRIVATE LDS slab, and
// runs the MFMAs against the resident weight planes - no workgroup barrier anywhere in the main loop, so
// the eight waves of a CU drift apart and one's loads / transform / stores overlap another's MFMAs.
// Epilogue as in mlp_gemm_kernel: raw output stored, column statistics (-> the in-kernel BN finalize by
// the last workgroup), and the max-pool over ns = 16 / 32 / 64 rows on the extremum the sign of gamma
// selects (a 64-row group = the wave's two tiles, folded in registers).
// LDS rows are K bf16 = 128 B (K = 64); 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7): a
// ds_read_b128 lane group {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} then touches every bank once.
constexpr int FR_NW = 8;
template <int KB>   // KB = bytes per row (K * 2)
__device__ __forceinline__ int fr_swz(int row, int chunk) {
  if constexpr (KB == 128) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
  else return row * KB + ((chunk ^ (row & 15)) << 4);
}
using f32x2_t = float __attribute__((ext_vector_type(2)));
using bf16x2_t = __bf16 __attribute__((ext_vector_type(2)));
template <int P>
__device__ __forceinline__ void fr_split_pair(float a, float b, unsigned (&o)[P]) {
  if constexpr (P == 2) {       // two fp16 terms (csrc/common.h)
    split2_f16(a, b, o[0], o[1]);
    return;
  }
  const f32x2_t x = {a, b};
  const bf16x2_t h = __builtin_convertvector(x, bf16x2_t);
  o[0] = __builtin_bit_cast(unsigned, h);
  if constexpr (P == 3) {
    const f32x2_t r = x - __builtin_convertvector(h, f32x2_t);
    const bf16x2_t m = __builtin_convertvector(r, bf16x2_t);
    o[1] = __builtin_bit_cast(unsigned, m);
    const f32x2_t l = r - __builtin_convertvector(m, f32x2_t);
    o[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2_t));
  }
}
template <int P>
__device__ __forceinline__ void fr_mfma(f32x16& acc, const bf16x8 (&a)[P], const bf16x8 (&b)[P]) {
  if constexpr (P == 2) {   // fp16 terms: l.h', h.l', h.h'
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a[1]), __builtin_bit_cast(f16x8_v, b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a[0]), __builtin_bit_cast(f16x8_v, b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a[0]), __builtin_bit_cast(f16x8_v, b[0]), acc, 0, 0, 0);
  } else if constexpr (P == 3) {   // six products of weight >= 2^-16, smallest first (as the CM = 2 path above)
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

// ST (bf16 compute mode only, BASELINE configs[3]): the rows themselves live in HBM as bf16 - bit 0: the
// input rows X, bit 1: the raw output Y (statistics and pooling still see the fp32 accumulators).  A
// lane pair (columns 2c, 2c+1) swaps one register per row pair through DPP so that every lane stores one
// packed dword: half the bytes AND half the store instructions of the fp32 form.
template <int NTN, int KT, bool POOL, int CM, int ST = 0>   // N = 32*NTN, K = 32*KT (KT = 2); CM 1 bf16 / 2 three bf16 terms / 3 two fp16 terms
__global__ __launch_bounds__(64 * FR_NW, 1) void mlp_fwd_res_kernel(MlpArgs p) {
  constexpr bool XB = (ST & 1) != 0, YB = (ST & 2) != 0;
  // ST bit 2 (pooled launches): the raw output is NOT stored at all - the backward of a pooled last
  // layer can be written without it (csrc/mlp_bwd.hip mlp_bwd_pool_kernel): what leaves the kernel is the
  // column statistics and, per group and channel, the selected extremum + its row.  Channels with
  // gamma == 0 (constant activation: upstream's max-pool picks slot 0) then report slot 0 and ITS raw
  // value, which the BN backward needs for dgamma and could otherwise only gather from Y.
  constexpr bool NOY = (ST & 4) != 0;
  static_assert(!NOY || (POOL && !YB), "no-store form: pooled launches, fp32 bookkeeping");
  // ST bit 3: X holds the 4-float input rows of the layer BELOW (SA1's first layer, 4 -> K channels): that
  // layer's raw output - 256 bytes per row - is never materialised; the 16 fma per lane and row that rebuild
  // the lane's four channels of it are free next to a 16-fold cut of the bytes read.  In the bf16 mode the
  // products are taken on bf16-rounded operands, as the MFMA kernel that used to produce the rows did.
  constexpr bool XR = (ST & 8) != 0;
  static_assert(!XR || !XB, "recomputed rows have no storage type");
  constexpr int P = CM == 2 ? 3 : (CM == 3 ? 2 : 1);
  constexpr int N = NTN * 32, K = KT * 32, KB = K * 2;
  constexpr int KS = K / 16;                       // MFMA K steps
  constexpr int LPR = K / 4;                       // lanes per row of the patch load (float4 each)
  constexpr int RPI = 64 / LPR;                    // rows per load instruction
  constexpr int NLD = 32 / RPI;                    // loads per 32-row tile
  static_assert(KT == 2, "rows of 64 channels");
  extern __shared__ __attribute__((aligned(16))) char fr_smem[];
  char* s_w = fr_smem;                                       // [P][N rows x KB]
  char* s_a = s_w + P * N * KB;                              // [8 waves][P][32 rows x KB]
  float* s_vec = reinterpret_cast<float*>(s_a + FR_NW * P * 32 * KB);   // scale | shift (2K)
  float* s_red = s_vec + 2 * K;                              // [8][NTN][64]
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  // ---- the weight (N x K) as split planes, once ------------------------------------------------------
  for (int i = tid; i < N * (K / 4); i += 64 * FR_NW) {
    const int n = i / (K / 4), k = 4 * (i % (K / 4));
    const float4 w = *reinterpret_cast<const float4*>(p.Bt + (size_t)n * K + k);
    unsigned p0[P], p1[P];
    fr_split_pair<P>(w.x, w.y, p0);
    fr_split_pair<P>(w.z, w.w, p1);
#pragma unroll
    for (int q = 0; q < P; ++q)
      *reinterpret_cast<uint2*>(s_w + q * N * KB + fr_swz<KB>(n, k >> 3) + 2 * (k & 7)) = make_uint2(p0[q], p1[q]);
  }
  for (int i = tid; i < 2 * K; i += 64 * FR_NW) s_vec[i] = p.vec[i];
  __syncthreads();
  // lane's patch: channels 4*c4 .. +3 of rows rq + RPI*j
  const int c4 = lane % LPR, rq = lane / LPR;
  const float4 sc = *reinterpret_cast<const float4*>(s_vec + 4 * c4);
  const float4 sh = *reinterpret_cast<const float4*>(s_vec + K + 4 * c4);
  float4 w0r[XR ? 4 : 1];                              // XR: rows 4*c4 .. +3 of the lower layer's (K x 4) weight
  if constexpr (XR) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float4 w = *reinterpret_cast<const float4*>(p.W0 + (size_t)(4 * c4 + e) * 4);
      if constexpr (CM == 1) {
        w.x = (float)(__bf16)w.x; w.y = (float)(__bf16)w.y; w.z = (float)(__bf16)w.z; w.w = (float)(__bf16)w.w;
      }
      w0r[e] = w;
    }
  }
  char* my_a = s_a + wave * P * 32 * KB;
  unsigned selbits = 0, zbits = 0;
  if constexpr (POOL) {
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
      const float gm = p.fin.gamma[nt * 32 + lr];
      if (gm < 0.f) selbits |= 1u << nt;
      if (NOY && gm == 0.f) zbits |= 1u << nt;
    }
  }
  float cs1[NTN], cs2[NTN];
#pragma unroll
  for (int nt = 0; nt < NTN; ++nt) cs1[nt] = cs2[nt] = 0.f;
  const int ntile = (p.R + 31) / 32;
  const int nunit = (ntile + 1) / 2;
  float4 raw[NLD];
  auto fetch = [&](int tile) {
    const int row0 = tile * 32;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      int row = row0 + rq + RPI * j;
      row = row < p.R ? row : p.R - 1;                     // valid address; zeroed by the transform
      if constexpr (XR) {
        raw[j] = *reinterpret_cast<const float4*>(p.X + (size_t)row * 4);       // the 4-float INPUT row
      } else if constexpr (XB) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(p.X) + (size_t)row * p.ldx + 4 * c4);
        raw[j] = make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                             __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
      } else {
        raw[j] = *reinterpret_cast<const float4*>(p.X + (size_t)row * p.ldx + 4 * c4);
      }
    }
  };
  int unit = blockIdx.x * FR_NW + wave;
  const int ustride = gridDim.x * FR_NW;
  if (unit < nunit) fetch(2 * unit);
  for (; unit < nunit; unit += ustride) {
    float pmx[NTN];
    int pax[NTN];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int tile = 2 * unit + half;
      if (tile >= ntile) break;
      const int row0 = tile * 32;
      // ---- BN + ReLU of the previous layer, split once, into the wave's private planes ---------------
      const bool tail = row0 + 32 > p.R;
#pragma unroll
      for (int j = 0; j < NLD; ++j) {
        const int rl = rq + RPI * j;
        float4 x = raw[j];
        if constexpr (XR) {                                  // the lane's four channels of the lower layer's output
          float4 r = x;
          if constexpr (CM == 1) {
            r.x = (float)(__bf16)r.x; r.y = (float)(__bf16)r.y; r.z = (float)(__bf16)r.z; r.w = (float)(__bf16)r.w;
          }
          x.x = __builtin_fmaf(r.w, w0r[0].w, __builtin_fmaf(r.z, w0r[0].z, __builtin_fmaf(r.y, w0r[0].y, r.x * w0r[0].x)));
          x.y = __builtin_fmaf(r.w, w0r[1].w, __builtin_fmaf(r.z, w0r[1].z, __builtin_fmaf(r.y, w0r[1].y, r.x * w0r[1].x)));
          x.z = __builtin_fmaf(r.w, w0r[2].w, __builtin_fmaf(r.z, w0r[2].z, __builtin_fmaf(r.y, w0r[2].y, r.x * w0r[2].x)));
          x.w = __builtin_fmaf(r.w, w0r[3].w, __builtin_fmaf(r.z, w0r[3].z, __builtin_fmaf(r.y, w0r[3].y, r.x * w0r[3].x)));
        }
        x.x = fmaxf(0.f, __builtin_fmaf(x.x, sc.x, sh.x));
        x.y = fmaxf(0.f, __builtin_fmaf(x.y, sc.y, sh.y));
        x.z = fmaxf(0.f, __builtin_fmaf(x.z, sc.z, sh.z));
        x.w = fmaxf(0.f, __builtin_fmaf(x.w, sc.w, sh.w));
        if (tail && row0 + rl >= p.R) x = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned p0[P], p1[P];
        fr_split_pair<P>(x.x, x.y, p0);
        fr_split_pair<P>(x.z, x.w, p1);
#pragma unroll
        for (int q = 0; q < P; ++q)
          *reinterpret_cast<uint2*>(my_a + q * 32 * KB + fr_swz<KB>(rl, c4 >> 1) + 8 * (c4 & 1)) =
              make_uint2(p0[q], p1[q]);
      }
      // next tile's rows in flight underneath this tile's MFMAs and epilogue
      {
        const int nxt = half == 0 ? tile + 1 : 2 * (unit + ustride);
        if (nxt < ntile && (half == 0 || unit + ustride < nunit)) fetch(nxt);
      }
      f32x16 acc[1][NTN];
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nt][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 a[P];
#pragma unroll
        for (int q = 0; q < P; ++q)
          a[q] = *reinterpret_cast<const bf16x8*>(my_a + q * 32 * KB + fr_swz<KB>(lr, 2 * s + lh));
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
          bf16x8 b[P];
#pragma unroll
          for (int q = 0; q < P; ++q)
            b[q] = *reinterpret_cast<const bf16x8*>(s_w + q * N * KB + fr_swz<KB>(nt * 32 + lr, 2 * s + lh));
          fr_mfma<P>(acc[0][nt], a, b);
        }
      }
      // ---- epilogue: raw output, column statistics, pooled extremum ---------------------------------
      const bool full = row0 + 32 <= p.R;
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) {
        const int col = nt * 32 + lr;
        if constexpr (NOY) {
          // (nothing to store)
        } else if constexpr (YB) {
          // lanes (2c, 2c+1): the even lane stores row r, the odd lane row r+1 of the register pair
          const bool odd = lane & 1;
          unsigned* yp = reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(p.Y) +
                                                      (size_t)(row0 + 4 * lh) * N + (col & ~1));
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float send = odd ? acc[0][nt][r] : acc[0][nt][r + 1];
            const float recv = dpp_f32<0xB1>(send, send);              // quad_perm [1,0,3,2]
            const f32x2_t v = {odd ? recv : acc[0][nt][r], odd ? acc[0][nt][r + 1] : recv};
            const int rr = r + (odd ? 1 : 0);
            const int rl = (rr & 3) + 8 * (rr >> 2);
            if (full || row0 + 4 * lh + rl < p.R)
              yp[(size_t)rl * (N / 2)] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
          }
        } else {
        float* yp = p.Y + (size_t)(row0 + 4 * lh) * N + col;
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * N] = acc[0][nt][r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (row0 + 4 * lh + (r & 3) + 8 * (r >> 2) < p.R) yp[(size_t)((r & 3) + 8 * (r >> 2)) * N] = acc[0][nt][r];
        }
        }
        f32x2_t a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                    // rows >= R are exact zeros
          const f32x2_t v = {acc[0][nt][r], acc[0][nt][r + 1]};
          a1 += v;
          a2 = __builtin_elementwise_fma(v, v, a2);
        }
        cs1[nt] += a1.x + a1.y;
        cs2[nt] += a2.x + a2.y;
        if constexpr (POOL) {
          const unsigned flip = ((selbits >> nt) & 1u) << 31;
          if (p.ns == 16) pool_epilogue_sel<1, NTN, 16>(p, acc, nt, row0, col, lh, flip);
          else if (p.ns == 32) pool_epilogue_sel<1, NTN, 32>(p, acc, nt, row0, col, lh, flip);
          else {
            // ns = 64: this tile's extremum (first maximum wins: rows ascend), folded with the other half
            float mx = -__builtin_inff();
            int ax = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rho = 32 * half + 4 * lh + (r & 3) + 8 * (r >> 2);
              const float v = flip_sign(acc[0][nt][r], flip);
              const bool up = v > mx;
              mx = up ? v : mx; ax = up ? rho : ax;
            }
            if constexpr (NOY) {
              if ((zbits >> nt) & 1u) {              // gamma == 0: slot 0 of the group and its raw value
                const bool has0 = half == 0 && lh == 0;
                mx = has0 ? flip_sign(acc[0][nt][0], flip) : -__builtin_inff();
                ax = has0 ? 0 : 64;
              }
            }
            const float omx = __shfl_xor(mx, 32);
            const int oax = __shfl_xor(ax, 32);
            if (omx > mx || (omx == mx && oax < ax)) { mx = omx; ax = oax; }
            if (half == 0) { pmx[nt] = mx; pax[nt] = ax; }
            else {
              if (!(mx > pmx[nt])) { mx = pmx[nt]; ax = pax[nt]; }     // the earlier half wins ties
              if (lh == 0) {
                const size_t o = (size_t)unit * N + col;
                p.pmax[o] = flip_sign(mx, flip);
                p.amax[o] = ax;
              }
            }
          }
        }
      }
    }
  }
  // ---- column statistics: lanes l / l+32 share a column, then the 8 waves, one fp64 atomic each -------
#pragma unroll
  for (int nt = 0; nt < NTN; ++nt) {
    cs1[nt] += __shfl_xor(cs1[nt], 32);
    cs2[nt] += __shfl_xor(cs2[nt], 32);
    if (lh == 0) {
      s_red[(wave * NTN + nt) * 64 + lr] = cs1[nt];
      s_red[(wave * NTN + nt) * 64 + 32 + lr] = cs2[nt];
    }
  }
  __syncthreads();
  for (int i = tid; i < NTN * 64; i += 64 * FR_NW) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < FR_NW; ++w) v += s_red[w * NTN * 64 + i];
    const int nt = i >> 6, which = (i >> 5) & 1, c = i & 31;
    double* sdst = p.fin.ss != nullptr && p.fin.racc != nullptr ? repl_copy(p.fin.racc, N, (int)blockIdx.x) : p.stats;
    atomicAdd(sdst + which * N + nt * 32 + c, (double)v);
  }
  if (p.fin.ss != nullptr) {
    // train-mode BN bookkeeping by the last workgroup (as mlp_gemm_kernel: only atomics touch the sums)
    sync_drained();
    if (tid == 0) {
      const int total = (int)gridDim.x;
      const int ngroups = total < SCHED_GROUPS ? total : SCHED_GROUPS;
      const int g = (int)blockIdx.x % SCHED_GROUPS;
      const int members = total / SCHED_GROUPS + (g < total % SCHED_GROUPS ? 1 : 0);
      int* t = p.fin.ticket + FIN_OFF;
      int last = 0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own sum atomics acknowledged first (csrc/bn_fin.h)
      if (atomicAdd(t + 1 + g, 1) == members - 1) {
        atomicExch(t + 1 + g, 0);
        if (atomicAdd(t, 1) == ngroups - 1) { atomicExch(t, 0); last = 1; }
      }
      s_last = last;
    }
    __syncthreads();
    if (s_last) {
      const BnFin& f = p.fin;
      if (tid == 0 && f.nbt != nullptr) *f.nbt += 1;
      for (int c = tid; c < N; c += 64 * FR_NW) {
        double s1 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(p.stats + c), 0ull));
        double s2 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(p.stats + N + c), 0ull));
        if (f.racc != nullptr) { s1 += repl_take(f.racc, N, c); s2 += repl_take(f.racc, N, N + c); }
        bn_finalize_channel(c, N, f.count, s1, s2, f.gamma, f.beta, f.eps, f.momentum, f.rmean, f.rvar, f.ss,
                            f.mi, f.conv_bias);
      }
    }
  }
}

// ---- forward GEMM for 128-channel inputs: producer and consumer waves ------------------------------------
// The layers behind SA1 (SA2-4, the vote aggregation: 128 -> 128 and 128 -> 256, 32 k ... 262 k rows) ran on
// mlp_gemm_kernel at ~82 TF/s = 20 % of what the three-term mode can issue: every K step of 32 restages both
// operands behind two workgroup barriers, and the loads of a step are in flight for one step's MFMAs only.
// Their weight does not fit LDS as split planes next to per-wave slabs (256 x 128 x 6 B = 196 KB), so the
// barrier-free form of mlp_fwd_res_kernel does not carry over.  Here a workgroup is 8 waves with two ROLES:
//   waves 0-3  PRODUCERS: stream 64-row slabs of the input (full 512-byte rows, three slabs in flight in
//              registers), apply the previous layer's BN + ReLU, split ONCE into bf16 planes in LDS (two stages);
//   waves 4-7  CONSUMERS: each owns 32 output columns for the whole launch with ITS weight fragments in
//              registers (8 K-steps x P planes x 4 VGPRs = 96), runs the slab's MFMAs straight from the planes,
//              stores the raw output, accumulates the column statistics and the pooled extremum.
// One LDS-only barrier per slab; each SIMD hosts one producer and one consumer, so VALU / memory work and the
// matrix pipe overlap by construction.  128 columns per workgroup: a 256-column layer runs as two column halves
// whose workgroups are co-scheduled on one XCD (blocks b and b + 8), so the second read of a slab is an L2 hit.
// Requires K == 128, N % 128 == 0, R % 64 == 0, contiguous rows.
constexpr int PC_K = 128, PC_KB = 256, PC_ROWS = 64;
template <int CM, bool POOL>
__global__ __launch_bounds__(512, 1) void mlp_fwd_pc_kernel(MlpArgs p) {
  constexpr int P = CM == 2 ? 3 : (CM == 3 ? 2 : 1);
  constexpr int K = PC_K, KB = PC_KB, KS = K / 16, SR = PC_ROWS, NL = SR / 8;
  constexpr int PF = 3;                                      // slabs in flight per producer (3 x 8 float4)
  extern __shared__ __attribute__((aligned(16))) char pc_smem[];
  char* s_a = pc_smem;                                       // [2 stages][P][64 rows x KB]
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const bool producer = wave < 4;
  int bx = (int)blockIdx.x, gx = (int)gridDim.x, h = 0;
  if (p.N == 256) {                                          // two column halves, interleaved in runs of 8 blocks
    h = (bx >> 3) & 1;
    bx = (bx & 7) | ((bx >> 4) << 3);
    gx >>= 1;
  }
  const int nslab = p.R / SR;
  const int N = p.N;
  // ---- producer state: lane's patch = channels 4*k4 .. +3 of rows prow + 8 i ---------------------------
  const int k4 = tid & 31, prow = (tid >> 5) & 7;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 rq[PF][NL];
  auto fetch = [&](int slab, float4 (&r)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      r[i] = *reinterpret_cast<const float4*>(p.X + (size_t)(slab * SR + prow + 8 * i) * p.ldx + 4 * k4);
  };
  auto commit = [&](int stage, const float4 (&r)[NL]) {
    char* base = s_a + (size_t)stage * P * SR * KB;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float4 x = r[i];
      x.x = fmaxf(0.f, __builtin_fmaf(x.x, sc.x, sh.x));
      x.y = fmaxf(0.f, __builtin_fmaf(x.y, sc.y, sh.y));
      x.z = fmaxf(0.f, __builtin_fmaf(x.z, sc.z, sh.z));
      x.w = fmaxf(0.f, __builtin_fmaf(x.w, sc.w, sh.w));
      unsigned p0[P], p1[P];
      fr_split_pair<P>(x.x, x.y, p0);
      fr_split_pair<P>(x.z, x.w, p1);
#pragma unroll
      for (int q = 0; q < P; ++q)
        *reinterpret_cast<uint2*>(base + q * SR * KB + fr_swz<KB>(prow + 8 * i, k4 >> 1) + 8 * (k4 & 1)) =
            make_uint2(p0[q], p1[q]);
    }
  };
  const int col = h * 128 + 32 * (wave & 3) + lr;             // consumers: this lane's output column
  float cs1 = 0.f, cs2 = 0.f;
  // (the two roles are two separate loops with the same barrier count, so that the register allocation is the
  // larger of the two states, not their sum)
  if (producer) {
    sc = *reinterpret_cast<const float4*>(p.vec + 4 * k4);
    sh = *reinterpret_cast<const float4*>(p.vec + K + 4 * k4);
    // rq[d] holds slab j + 1 + d of this workgroup's sequence; slab 0 goes through rq[PF - 1] first
    if (bx < nslab) { fetch(bx, rq[PF - 1]); }
#pragma unroll
    for (int d = 0; d < PF - 1; ++d)
      if (bx + (d + 1) * gx < nslab) fetch(bx + (d + 1) * gx, rq[d]);
    if (bx < nslab) commit(0, rq[PF - 1]);
    if (bx + PF * gx < nslab) fetch(bx + PF * gx, rq[PF - 1]);
    lds_barrier();
    int j = 0;
    for (int slab = bx; slab < nslab; slab += gx, ++j) {
      // slab j + 1 into the other stage (its rows were requested PF slabs ago), then slab j + 1 + PF into flight
      if (slab + gx < nslab) commit((j + 1) & 1, rq[0]);
#pragma unroll
      for (int d = 0; d + 1 < PF; ++d)
#pragma unroll
        for (int i = 0; i < NL; ++i) rq[d][i] = rq[d + 1][i];
      if (slab + (PF + 1) * gx < nslab) fetch(slab + (PF + 1) * gx, rq[PF - 1]);
      lds_barrier();
    }
  } else {
    // ---- consumer: its weight row as MFMA B fragments, resident for the whole launch ------------------------
    bf16x8 bfrag[KS][P];
    unsigned flip = 0;
    const float* wrow = p.Bt + (size_t)col * K + 8 * lh;
#pragma unroll
    for (int sidx = 0; sidx < KS; ++sidx) {
      const float4 w0 = *reinterpret_cast<const float4*>(wrow + 16 * sidx);
      const float4 w1 = *reinterpret_cast<const float4*>(wrow + 16 * sidx + 4);
      unsigned q0[P], q1[P], q2[P], q3[P];
      fr_split_pair<P>(w0.x, w0.y, q0);
      fr_split_pair<P>(w0.z, w0.w, q1);
      fr_split_pair<P>(w1.x, w1.y, q2);
      fr_split_pair<P>(w1.z, w1.w, q3);
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const uint4 u = make_uint4(q0[q], q1[q], q2[q], q3[q]);
        bfrag[sidx][q] = __builtin_bit_cast(bf16x8, u);
      }
    }
    if constexpr (POOL) flip = p.fin.gamma[col] < 0.f ? 0x80000000u : 0u;
    lds_barrier();
    int j = 0;
    for (int slab = bx; slab < nslab; slab += gx, ++j) {
      const char* base = s_a + (size_t)(j & 1) * P * SR * KB;
      // two row tiles = two independent accumulator chains
      f32x16 acc[2][1];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][0][r] = 0.f;
#pragma unroll
      for (int sidx = 0; sidx < KS; ++sidx) {
        bf16x8 a0[P], a1[P];
#pragma unroll
        for (int q = 0; q < P; ++q) {
          a0[q] = *reinterpret_cast<const bf16x8*>(base + q * SR * KB + fr_swz<KB>(lr, 2 * sidx + lh));
          a1[q] = *reinterpret_cast<const bf16x8*>(base + q * SR * KB + fr_swz<KB>(32 + lr, 2 * sidx + lh));
        }
        fr_mfma<P>(acc[0][0], a0, bfrag[sidx]);
        fr_mfma<P>(acc[1][0], a1, bfrag[sidx]);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row0 = slab * SR + 32 * rt;
        float* yp = p.Y + (size_t)(row0 + 4 * lh) * p.ldy + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldy] = acc[rt][0][r];
        f32x2_t a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_t v = {acc[rt][0][r], acc[rt][0][r + 1]};
          a1 += v;
          a2 = __builtin_elementwise_fma(v, v, a2);
        }
        cs1 += a1.x + a1.y;
        cs2 += a2.x + a2.y;
        if constexpr (POOL) {
          const f32x16 (&one)[1][1] = *reinterpret_cast<const f32x16 (*)[1][1]>(&acc[rt]);
          if (p.ns == 16) pool_epilogue_sel<1, 1, 16>(p, one, 0, row0, col, lh, flip);
          else pool_epilogue_sel<1, 1, 32>(p, one, 0, row0, col, lh, flip);
        }
      }
      lds_barrier();
    }
  }
  // ---- column statistics: lanes l / l + 32 share a column; a consumer wave owns its columns alone --------
  if (!producer) {
    cs1 += __shfl_xor(cs1, 32);
    cs2 += __shfl_xor(cs2, 32);
    if (lh == 0) {
      double* sdst = p.fin.ss != nullptr && p.fin.racc != nullptr ? repl_copy(p.fin.racc, N, (int)blockIdx.x) : p.stats;
      atomicAdd(sdst + col, (double)cs1);
      atomicAdd(sdst + N + col, (double)cs2);
    }
  }
  if (p.fin.ss != nullptr) {
    // train-mode BN bookkeeping by the last workgroup (as mlp_gemm_kernel: only atomics touch the sums)
    sync_drained();
    if (tid == 0) {
      const int total = (int)gridDim.x;
      const int ngroups = total < SCHED_GROUPS ? total : SCHED_GROUPS;
      const int g = (int)blockIdx.x % SCHED_GROUPS;
      const int members = total / SCHED_GROUPS + (g < total % SCHED_GROUPS ? 1 : 0);
      int* t = p.fin.ticket + FIN_OFF;
      int last = 0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own sum atomics acknowledged first (csrc/bn_fin.h)
      if (atomicAdd(t + 1 + g, 1) == members - 1) {
        atomicExch(t + 1 + g, 0);
        if (atomicAdd(t, 1) == ngroups - 1) { atomicExch(t, 0); last = 1; }
      }
      s_last = last;
    }
    __syncthreads();
    if (s_last) {
      const BnFin& f = p.fin;
      if (tid == 0 && f.nbt != nullptr) *f.nbt += 1;
      for (int c = tid; c < N; c += 512) {
        double s1 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(p.stats + c), 0ull));
        double s2 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(p.stats + N + c), 0ull));
        if (f.racc != nullptr) { s1 += repl_take(f.racc, N, c); s2 += repl_take(f.racc, N, N + c); }
        bn_finalize_channel(c, N, f.count, s1, s2, f.gamma, f.beta, f.eps, f.momentum, f.rmean, f.rvar, f.ss,
                            f.mi, f.conv_bias);
      }
    }
  }
}

// ---- BN statistics -> per-channel scale/shift (+ running stats, saved mean/invstd) -----------
__global__ void bn_finalize_kernel(int N, double count, double* __restrict__ stats,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var,
                                   long long* __restrict__ num_batches_tracked,
                                   float* __restrict__ ss, float* __restrict__ mi,
                                   const float* __restrict__ conv_bias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
  if (c >= N) return;
  const double s1 = stats[c], s2 = stats[N + c];
  stats[c] = 0.0;                                   // consumed: the accumulator is left zeroed
  stats[N + c] = 0.0;
  bn_finalize_channel(c, N, count, s1, s2, gamma, beta, eps, momentum, running_mean, running_var, ss,
                      mi, conv_bias);
}

// ---- tail: out[r,c] = max_s max(0, y[r,s,c]*scale+shift), first maximum wins -----------------
__global__ __launch_bounds__(256) void bnrelu_maxpool_fwd_k(long long RC, int ns, int C,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ ss,
                                                            float* __restrict__ out,
                                                            int* __restrict__ arg) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; t < RC; t += stride) {
    const long long r = t / C;
    const int c = (int)(t - r * C);
    const float sc = ss[c], sh = ss[C + c];
    const float* p = y + r * ns * C + c;
    float best = fmaxf(0.f, __builtin_fmaf(p[0], sc, sh));
    int bi = 0;
    for (int s = 1; s < ns; ++s) {
      const float v = fmaxf(0.f, __builtin_fmaf(p[(size_t)s * C], sc, sh));
      if (v > best) {
        best = v;
        bi = s;
      }
    }
    out[t] = best;
    arg[t] = bi;
  }
}

// ---- tail for the fused pooling epilogue: pick max or min by the sign of the BN scale -----------
__global__ __launch_bounds__(256) void pool_select_k(long long RC, int C,
                                                     const float* __restrict__ pmax,
                                                     const float* __restrict__ pmin,
                                                     const int* __restrict__ amax,
                                                     const int* __restrict__ amin,
                                                     const float* __restrict__ ss,
                                                     float* __restrict__ out,
                                                     int* __restrict__ arg,
                                                     float* __restrict__ yraw, int slot0) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; t < RC; t += stride) {
    const int c = (int)(t % C);
    const float sc = ss[c], sh = ss[C + c];
    const bool up = sc > 0.f || (slot0 && sc == 0.f);     // slot0: zero-scale channels carry slot 0 in pmax
    const float y = up ? pmax[t] : pmin[t];
    out[t] = fmaxf(0.f, __builtin_fmaf(y, sc, sh));
    const int a = sc == 0.f ? 0 : (up ? amax[t] : amin[t]);  // constant activation: first slot wins
    arg[t] = a;
    // the raw output AT the selected row (what the sparse BN-backward reduce would otherwise gather
    // from Y).  Zero scale: the selected row is slot 0, whose value is not among the extrema - NaN
    // tells the consumer to gather it (dgamma = sum dZ*xhat needs it even though gamma is 0).
    // (``slot0``: the producer tracked slot 0 and its raw value for zero-gamma channels itself - the no-store
    // forward, which has no Y to gather from)
    if (yraw) yraw[t] = (sc == 0.f && !slot0) ? __builtin_nanf("") : y;
  }
}

// ---- backward reductions: g1 = sum dZ, g2 = sum dZ*xhat per channel ---------------------------
// dense upstream gradient G (R x N); one block strides over rows, 256 threads = 64 col4 x 4 rows
template <bool SPARSE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_k(int R, int N, int ns,
                                                       const float* __restrict__ G,
                                                       const float* __restrict__ dP,
                                                       const int* __restrict__ arg,
                                                       const float* __restrict__ Y,
                                                       const float* __restrict__ ss,
                                                       const float* __restrict__ mi,
                                                       double* __restrict__ g12,
                                                       const float* __restrict__ yraw, BnVecFin fin,
                                                       int y_bf16) {
  // thread -> one column, strided rows; columns are the fast index so reads coalesce
  __shared__ float red[2][256];
  __shared__ int s_last;
  const int cols_per_pass = N < 256 ? N : 256;
  const int rows_par = 256 / cols_per_pass;           // row lanes per block pass
  const int c_in = threadIdx.x % cols_per_pass, r_in = threadIdx.x / cols_per_pass;
  for (int cb = 0; cb < N; cb += cols_per_pass) {
    const int c = cb + c_in;
    float a1 = 0.f, a2 = 0.f;
    if (r_in < rows_par && c < N) {
      const float sc = ss[c], sh = ss[N + c], mu = mi[c], is = mi[N + c];
      if constexpr (SPARSE) {
        const int Rp = R / ns;
        for (int rp = blockIdx.x * rows_par + r_in; rp < Rp; rp += gridDim.x * rows_par) {
          // y at the arg-max row: handed over by demf_pool_select, else gathered from Y
          float y = yraw ? yraw[(size_t)rp * N + c] : __builtin_nanf("");
          if (y != y) {
            const size_t o = ((size_t)rp * ns + arg[(size_t)rp * N + c]) * N + c;
            y = y_bf16 ? (float)reinterpret_cast<const __bf16*>(Y)[o] : Y[o];
          }
          const float dz = __builtin_fmaf(y, sc, sh) > 0.f ? dP[(size_t)rp * N + c] : 0.f;
          a1 += dz;
          a2 = __builtin_fmaf(dz, (y - mu) * is, a2);
        }
      } else {
        for (int r = blockIdx.x * rows_par + r_in; r < R; r += gridDim.x * rows_par) {
          const float y = Y[(size_t)r * N + c];
          const float dz = __builtin_fmaf(y, sc, sh) > 0.f ? G[(size_t)r * N + c] : 0.f;
          a1 += dz;
          a2 = __builtin_fmaf(dz, (y - mu) * is, a2);
        }
      }
    }
    red[0][threadIdx.x] = a1;
    red[1][threadIdx.x] = a2;
    __syncthreads();
    if (threadIdx.x < cols_per_pass && cb + threadIdx.x < N) {
      float t1 = 0.f, t2 = 0.f;
      for (int j = 0; j < rows_par; ++j) {
        t1 += red[0][j * cols_per_pass + threadIdx.x];
        t2 += red[1][j * cols_per_pass + threadIdx.x];
      }
      atomicAdd(g12 + cb + threadIdx.x, (double)t1);
      atomicAdd(g12 + N + cb + threadIdx.x, (double)t2);
    }
    __syncthreads();
  }
  if (fin.ticket != nullptr) {     // this layer's backward vectors by the last workgroup (csrc/bn_fin.h)
    sync_drained();
    if (threadIdx.x == 0) s_last = last_workgroup(fin.ticket, (int)gridDim.x, (int)blockIdx.x);
    __syncthreads();
    if (s_last) bn_vec_finalize(fin, N, 0, N, g12, threadIdx.x, 256);
  }
}

// dense G, N % 4 == 0, N <= 1024: a thread owns 4 consecutive channels (float4 loads of G and Y),
// row lanes fill the rest of the block; 4 rows in flight per thread.  `rev`: rows are visited last to
// first - G has just been written front to back by the dx GEMM, so its tail is what the 256 MB
// memory-side cache still holds (measured -0.03 ms/step).
// GATHER: the pooled (sparse) form on its R/ns pooled rows - G = dP, Y = yraw (the raw output at the
// selected row, handed over by demf_pool_select); a NaN there (zero BN scale: no extremum was selected)
// sends that one value to the arg-max row of the full output Yfull, as bn_bwd_reduce_k<true> does.
template <bool GATHER>
__global__ __launch_bounds__(256) void bn_bwd_reduce_dense4_k(int R, int N,
                                                              const float* __restrict__ G,
                                                              const float* __restrict__ Y,
                                                              const float* __restrict__ ss,
                                                              const float* __restrict__ mi,
                                                              double* __restrict__ g12, int rev,
                                                              BnVecFin fin, const float* __restrict__ Yfull,
                                                              const int* __restrict__ arg, int ns, int y_bf16) {
  __shared__ float4 red[2][256];
  __shared__ int s_last;
  const int cg = N >> 2;
  const int rows_par = 256 / cg;
  const int c4 = threadIdx.x % cg, r_in = threadIdx.x / cg;
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
  if (r_in < rows_par) {
    const float4 sc = reinterpret_cast<const float4*>(ss)[c4];
    const float4 sh = reinterpret_cast<const float4*>(ss + N)[c4];
    const float4 mu = reinterpret_cast<const float4*>(mi)[c4];
    const float4 is = reinterpret_cast<const float4*>(mi + N)[c4];
    const int step = gridDim.x * rows_par;
    auto acc = [&](const float4& g, const float4& y) {
      const float d0 = __builtin_fmaf(y.x, sc.x, sh.x) > 0.f ? g.x : 0.f;
      const float d1 = __builtin_fmaf(y.y, sc.y, sh.y) > 0.f ? g.y : 0.f;
      const float d2 = __builtin_fmaf(y.z, sc.z, sh.z) > 0.f ? g.z : 0.f;
      const float d3 = __builtin_fmaf(y.w, sc.w, sh.w) > 0.f ? g.w : 0.f;
      a1.x += d0; a1.y += d1; a1.z += d2; a1.w += d3;
      a2.x = __builtin_fmaf(d0, (y.x - mu.x) * is.x, a2.x);
      a2.y = __builtin_fmaf(d1, (y.y - mu.y) * is.y, a2.y);
      a2.z = __builtin_fmaf(d2, (y.z - mu.z) * is.z, a2.z);
      a2.w = __builtin_fmaf(d3, (y.w - mu.w) * is.w, a2.w);
    };
    auto fix = [&](float4& y, int rr) {          // GATHER: NaN -> the value at the arg-max row
      if constexpr (GATHER) {
        auto one = [&](float& v, int c) {
          if (v != v) {
            const size_t o = ((size_t)rr * ns + arg[(size_t)rr * N + c]) * N + c;
            v = y_bf16 ? (float)reinterpret_cast<const __bf16*>(Yfull)[o] : Yfull[o];
          }
        };
        one(y.x, 4 * c4); one(y.y, 4 * c4 + 1); one(y.z, 4 * c4 + 2); one(y.w, 4 * c4 + 3);
      }
    };
    int r = blockIdx.x * rows_par + r_in;
    for (; r + 3 * step < R; r += 4 * step) {
      float4 g[4], y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = rev ? R - 1 - (r + u * step) : r + u * step;
        const size_t o = (size_t)rr * N + 4 * c4;
        g[u] = *reinterpret_cast<const float4*>(G + o);
        y[u] = *reinterpret_cast<const float4*>(Y + o);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        fix(y[u], rev ? R - 1 - (r + u * step) : r + u * step);
        acc(g[u], y[u]);
      }
    }
    for (; r < R; r += step) {
      const int rr = rev ? R - 1 - r : r;
      const size_t o = (size_t)rr * N + 4 * c4;
      float4 y = *reinterpret_cast<const float4*>(Y + o);
      fix(y, rr);
      acc(*reinterpret_cast<const float4*>(G + o), y);
    }
  }
  red[0][threadIdx.x] = a1;
  red[1][threadIdx.x] = a2;
  __syncthreads();
  if (threadIdx.x < cg) {
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
    for (int j = 0; j < rows_par; ++j) {
      const float4 u = red[0][j * cg + threadIdx.x], v = red[1][j * cg + threadIdx.x];
      t1.x += u.x; t1.y += u.y; t1.z += u.z; t1.w += u.w;
      t2.x += v.x; t2.y += v.y; t2.z += v.z; t2.w += v.w;
    }
    double* dst = fin.ticket != nullptr && fin.racc != nullptr ? repl_copy(fin.racc, N, (int)blockIdx.x) : g12;
    double* o1 = dst + 4 * threadIdx.x;
    double* o2 = dst + N + 4 * threadIdx.x;
    atomicAdd(o1, (double)t1.x); atomicAdd(o1 + 1, (double)t1.y);
    atomicAdd(o1 + 2, (double)t1.z); atomicAdd(o1 + 3, (double)t1.w);
    atomicAdd(o2, (double)t2.x); atomicAdd(o2 + 1, (double)t2.y);
    atomicAdd(o2 + 2, (double)t2.z); atomicAdd(o2 + 3, (double)t2.w);
  }
  if (fin.ticket != nullptr) {     // this layer's backward vectors by the last workgroup (csrc/bn_fin.h)
    sync_drained();
    if (threadIdx.x == 0) s_last = last_workgroup(fin.ticket, (int)gridDim.x, (int)blockIdx.x);
    __syncthreads();
    if (s_last) bn_vec_finalize(fin, N, 0, N, g12, threadIdx.x, 256);
  }
}

// g1,g2 -> the 5 backward vectors + dgamma, dbeta
__global__ void bn_bwd_vectors_k(int N, double count, double* __restrict__ g12,
                                 const float* __restrict__ gamma, const float* __restrict__ ss,
                                 const float* __restrict__ mi, float* __restrict__ vec,
                                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const double g1 = g12[c], g2 = g12[N + c];
  g12[c] = 0.0;                                     // consumed: left zeroed for the next user
  g12[N + c] = 0.0;
  const double mean = mi[c], is = mi[N + c];
  const double gi = (double)gamma[c] * is;
  const double a = -gi * is * (g2 / count);
  vec[c] = ss[c];
  vec[N + c] = ss[N + c];
  vec[2 * N + c] = (float)gi;
  vec[3 * N + c] = (float)a;
  vec[4 * N + c] = (float)(-gi * (g1 / count) - a * mean);
  dgamma[c] = (float)g2;
  dbeta[c] = (float)g1;
}

// ---- dW(N x K) += dY(R x N)^T @ A(R x K): reduction over rows --------------------------------
// Block = 4 waves sharing a 32-row slab of dY and A in LDS per step; the NxK output is split into
// 32x32 MFMA tiles dealt round-robin to the waves (TPW tiles per wave).  Each block reduces its
// row range in registers and adds the partial with fp32 atomics.
struct DwArgs {
  int R, N, K, ldx;
  const float* Yl;     // (R x N) pre-BN output of this layer
  const float* G;      // dense upstream gradient or null
  const float* dP;     // sparse upstream gradient
  const int* arg;
  int ns;
  const float* vec;    // 5 x N backward vectors of this layer
  const float* Xp;     // (R x K) previous layer's pre-BN output, or the raw input
  const float* pvec;   // [scale|shift] of the previous layer (2K) or null (raw input)
  float* dW;           // (N x K), accumulated
  int lddw;            // row stride of dW (>= K)
  int n0, k0, NTn, NTk;  // output sub-block of a block (derived from blockIdx.y inside the kernel)
  int nsub_k;            // sub-blocks along K
  int chunk;             // 32-row slabs per claim
  int* sched;            // dynamic slab chunks: per-blockIdx.y {next chunk}, {finished}, + total; or null
  int dbg;               // DEMF_DW_DBG phase-skip bits (builds with -DDEMF_DW_PROFILE only): 1 MFMAs, 2 gather + split,
                         // 4 transform + stage, 8 flush, 16 loads
};
#ifdef DEMF_DW_PROFILE
#define DW_DBG(p, bit) ((p).dbg & (bit))
#else
#define DW_DBG(p, bit) false
#endif

#ifndef DEMF_DW_TR
#define DEMF_DW_TR 1      // transpose-read fragments from bf16 planes (0: the per-wave fp32 column gather - A/B builds)
#endif

// Waves form a 2 x 2 grid over the (<= 4 x 4) output tiles of the launch: wave (wn, wk) owns
// tiles tn = wn + 2i (i < TN), tk = wk + 2j (j < TK), so per row pair it reads TN + TK LDS
// values for TN*TK MFMAs.
// X3 (compute mode 2): the slab stays fp32 in LDS; per 16 rows a lane gathers its 8 rows of one column
// (8 ds_read_b32, conflict-free: consecutive lanes read consecutive columns), splits them into three
// bf16 terms and issues the six significant products on v_mfma_f32_32x32x16_bf16.
template <int TN, int TK, bool SPARSE, int X3 = 0>   // 0 fp32 MFMA, 1 three-term split, 2 one bf16 term
__device__ __forceinline__ void mlp_dw_body(DwArgs p, const int bx, const int by, const int gx) {
  constexpr int PROY = SPARSE ? PRO_DY_SPARSE : PRO_DY_DENSE;
  {  // this block's (<= 2TN x 2TK tiles) corner of the N x K output
    const int TNt = (p.N + 31) / 32, TKt = (p.K + 31) / 32;
    p.n0 = (by / p.nsub_k) * 2 * TN;
    p.k0 = (by % p.nsub_k) * 2 * TK;
    p.NTn = TNt - p.n0 < 2 * TN ? TNt - p.n0 : 2 * TN;
    p.NTk = TKt - p.k0 < 2 * TK ? TKt - p.k0 : 2 * TK;
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // staged widths are padded to the wave grid (2*TN, 2*TK tiles) so idle tiles read zeros
  constexpr int WN = 2 * TN * 32, WK = 2 * TK * 32;  // columns of dY / A staged per slab
  // TR (round 5, the bf16-MFMA modes): the slab is split ONCE by the staging threads and kept as row-major bf16
  // planes [PL][32][W + 8]; a wave's fragment - 8 consecutive ROWS of one column - is two ds_read_b64_tr_b16 per
  // plane (the 16 lanes of a group pass the addresses of a 4-row x 16-column block and receive one column each:
  // tools/ubench/tr16_probe.cpp).  Before, every wave gathered its columns from an fp32 slab with 8 ds_read_b32 and
  // split them itself - each element twice over (the two waves that share an operand), 250 of the loop's 650
  // instructions.  Same planes, same MFMA order: bit-identical results.
  constexpr bool TR = DEMF_DW_TR && X3 != 0;
  constexpr int PL = X3 == 1 ? 3 : 1;
  constexpr int ldn = TR ? WN + 8 : WN + 4, ldk = TR ? WK + 8 : WK + 4;     // row strides (bf16 / float elements)
  constexpr int PBN = 32 * ldn * 2, PBK = 32 * ldk * 2;                     // TR: bytes per plane
  float* s_dy = smem;                                // [32][WN + 4] fp32, or the dY planes
  float* s_a = TR ? reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + PL * PBN) : s_dy + 32 * ldn;
  float* s_vy = TR ? reinterpret_cast<float*>(reinterpret_cast<char*>(s_a) + PL * PBK) : s_a + 32 * ldk;   // 5 x N vectors
  float* s_vx = s_vy + 5 * p.N;                      // [scale|shift] of the previous layer (2K)
  char* const p_dy = reinterpret_cast<char*>(s_dy);
  char* const p_a = reinterpret_cast<char*>(s_a);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wn = wave >> 1, wk = wave & 1;
  for (int i = threadIdx.x; i < 5 * p.N; i += 256) s_vy[i] = p.vec[i];
  if (p.pvec)
    for (int i = threadIdx.x; i < 2 * p.K; i += 256) s_vx[i] = p.pvec[i];
  f32x16 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  MlpArgs ay;  // reuse the GEMM prologue helpers
  ay.K = p.N; ay.ldx = p.N; ay.X = p.Yl; ay.G = p.G; ay.dP = p.dP; ay.arg = p.arg; ay.ns = p.ns;
  MlpArgs ax;
  ax.K = p.K; ax.ldx = p.ldx; ax.X = p.Xp;

  // per-thread slots of the 32-row slab: float4 index f = threadIdx.x + 256*j
  constexpr int qn = WN / 4, qk = WK / 4;
  MlpRaw<PROY> ry[4];
  MlpRaw<PRO_NONE> ra[4];
  // TR: a float4 (4 consecutive columns of a row) -> its PL planes, one ds_write_b64 each
  auto stage_planes = [&](char* dst, int plane_bytes, const float4& v) {
    if constexpr (X3 == 1) {
      bf16x4 h, m, l;
      split3(v, h, m, l);
      *reinterpret_cast<bf16x4*>(dst) = h;
      *reinterpret_cast<bf16x4*>(dst + plane_bytes) = m;
      *reinterpret_cast<bf16x4*>(dst + 2 * plane_bytes) = l;
    } else {
      *reinterpret_cast<bf16x4*>(dst) = to_bf16x4(v);
    }
  };
  // TR: rows 16c + 8lh .. + 7 of column col0 + lr of one plane (lane t of a 16-lane group addresses row t >> 2,
  // columns 4 (t & 3) .. + 3 of the group's 4 x 16 block)
  const int tr_off = ((lane & 15) >> 2) * 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  auto frag = [&](const char* plane, int ld, int col0, int c) {
    using v4s = short __attribute__((ext_vector_type(4)));
    using lds_v4s = v4s __attribute__((address_space(3)));
    const char* q = plane + ((16 * c + 8 * lh) * ld + col0 + tr_col) * 2 + tr_off * ld;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(q));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(q + 8 * ld));
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  bool oky[4], oka[4];
  auto prefetch = [&](int slab) {
    const int row0 = slab * 32;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int f = threadIdx.x + 256 * jj;
      const int r = f / qn, c = (f - r * qn) * 4;
      oky[jj] = f < 32 * qn && row0 + r < p.R && p.n0 * 32 + c < p.N && !DW_DBG(p, 16);
      mlp_fetch<PROY>(ay, row0 + r, p.n0 * 32 + c, oky[jj], ry[jj]);
      const int r2 = f / qk, c2 = (f - r2 * qk) * 4;
      oka[jj] = f < 32 * qk && row0 + r2 < p.R && p.k0 * 32 + c2 < p.K && !DW_DBG(p, 16);
      mlp_fetch<PRO_NONE>(ax, row0 + r2, p.k0 * 32 + c2, oka[jj], ra[jj]);
    }
  };
  // Rows are handed out in chunks of p.chunk 32-row slabs: the first chunk by block index, every
  // further one from a device counter per sub-block column (by), claimed a chunk ahead.
  const int nslab = (p.R + 31) / 32;
  const int nchunk = (nslab + p.chunk - 1) / p.chunk;
  __shared__ int s_next;
  const bool dyn = p.sched != nullptr;
  int claimed = 0;
  int chunk = bx, si = 0;
  if (chunk < nchunk) prefetch(chunk * p.chunk);
  while (chunk < nchunk) {
    const int slab = chunk * p.chunk + si;
    const bool last = si == p.chunk - 1 || slab + 1 >= nslab;
    if (dyn && threadIdx.x == 0) {
      if (si == 0) claimed = gx + atomicAdd(p.sched + by, 1);
      if (last) s_next = claimed;
    }
    __syncthreads();                                 // previous slab fully consumed (and s_v* ready)
    if (!DW_DBG(p, 4))
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int f = threadIdx.x + 256 * jj;
      if (f < 32 * qn) {
        const int r = f / qn, c = (f - r * qn) * 4;
        const float4 v = mlp_xform<PROY>(ay, s_vy, p.n0 * 32 + c, oky[jj], ry[jj]);
        if constexpr (TR) stage_planes(p_dy + (r * ldn + c) * 2, PBN, v);
        else *reinterpret_cast<float4*>(s_dy + r * ldn + c) = v;
      }
      if (f < 32 * qk) {
        const int r = f / qk, c = (f - r * qk) * 4;
        float4 v = ra[jj].x;
        if (p.pvec) {
          MlpRaw<PRO_BNRELU> rb;
          rb.x = v;
          v = mlp_xform<PRO_BNRELU>(ax, s_vx, p.k0 * 32 + c, oka[jj], rb);
        }
        if constexpr (TR) stage_planes(p_a + (r * ldk + c) * 2, PBK, v);
        else *reinterpret_cast<float4*>(s_a + r * ldk + c) = v;
      }
    }
    __syncthreads();
    const int next_chunk = last ? (dyn ? s_next : chunk + gx) : chunk;
    const int next_si = last ? 0 : si + 1;
    if (next_chunk < nchunk) prefetch(next_chunk * p.chunk + next_si);
    if constexpr (TR && X3 == 2) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 a8[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) a8[i] = frag(p_dy, ldn, (wn + 2 * i) * 32, c);
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          const bf16x8 b8 = frag(p_a, ldk, (wk + 2 * j) * 32, c);
#pragma unroll
          for (int i = 0; i < TN; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8, acc[i][j], 0, 0, 0);
        }
      }
    } else if constexpr (TR && X3 == 1) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 ah[TN], am[TN], al[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          ah[i] = frag(p_dy, ldn, (wn + 2 * i) * 32, c);
          am[i] = frag(p_dy + PBN, ldn, (wn + 2 * i) * 32, c);
          al[i] = frag(p_dy + 2 * PBN, ldn, (wn + 2 * i) * 32, c);
        }
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          const bf16x8 bh = frag(p_a, ldk, (wk + 2 * j) * 32, c);
          const bf16x8 bm = frag(p_a + PBK, ldk, (wk + 2 * j) * 32, c);
          const bf16x8 bl = frag(p_a + 2 * PBK, ldk, (wk + 2 * j) * 32, c);
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            if (DW_DBG(p, 1)) { acc[i][j][0] += (float)al[i][0] + (float)bh[0]; continue; }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
          }
        }
      }
    } else if constexpr (X3 == 2) {
      // compute dtype bf16: the same column gather, operands rounded to bf16, one MFMA per tile pair
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        auto gather1 = [&](const float* col, int ld) {
          const float* q = col + (16 * c + 8 * lh) * ld;
          const bf16x4 lo = to_bf16x4(make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]));
          const bf16x4 hi = to_bf16x4(make_float4(q[4 * ld], q[5 * ld], q[6 * ld], q[7 * ld]));
          return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        bf16x8 a8[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) a8[i] = gather1(s_dy + (wn + 2 * i) * 32 + lr, ldn);
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          const bf16x8 b8 = gather1(s_a + (wk + 2 * j) * 32 + lr, ldk);
#pragma unroll
          for (int i = 0; i < TN; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8, acc[i][j], 0, 0, 0);
        }
      }
    } else if constexpr (X3 == 1) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        // lane supplies rows 16c + 8lh .. +7 of column lr of its tiles
        bf16x8 ah[TN], am[TN], al[TN];
        auto gather = [&](const float* col, int ld, bf16x8& h, bf16x8& m, bf16x8& l) {
          if (DW_DBG(p, 2)) { h = m = l = bf16x8{}; return; }
          const float* q = col + (16 * c + 8 * lh) * ld;
          split3(make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]),
                 make_float4(q[4 * ld], q[5 * ld], q[6 * ld], q[7 * ld]), h, m, l);
        };
#pragma unroll
        for (int i = 0; i < TN; ++i) gather(s_dy + (wn + 2 * i) * 32 + lr, ldn, ah[i], am[i], al[i]);
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          bf16x8 bh, bm, bl;
          gather(s_a + (wk + 2 * j) * 32 + lr, ldk, bh, bm, bl);
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            if (DW_DBG(p, 1)) { acc[i][j][0] += (float)al[i][0] + (float)bh[0]; continue; }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
          }
        }
      }
    } else
    // A-op: dY^T -> lane supplies dY[r = 2m + lh][n = tn*32 + lr]; B-op: A[r][k = tk*32 + lr]
#pragma unroll 4
    for (int m = 0; m < 16; ++m) {
      float a[TN], b[TK];
#pragma unroll
      for (int i = 0; i < TN; ++i) a[i] = s_dy[(2 * m + lh) * ldn + (wn + 2 * i) * 32 + lr];
#pragma unroll
      for (int j = 0; j < TK; ++j) b[j] = s_a[(2 * m + lh) * ldk + (wk + 2 * j) * 32 + lr];
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    chunk = next_chunk;
    si = next_si;
  }
  if (dyn && threadIdx.x == 0) {
    // re-arm the counters for the next launch that is handed this set: column-lasts, then the last
    int* done = p.sched + DW_MAX_SUB;
    if (atomicAdd(done + by, 1) == gx - 1) {
      atomicExch(done + by, 0);
      atomicExch(p.sched + by, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int tn = wn + 2 * i, tk = wk + 2 * j;
      if (tn < p.NTn && tk < p.NTk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = (p.n0 + tn) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int k = (p.k0 + tk) * 32 + lr;
          if (n < p.N && k < p.K && (!DW_DBG(p, 8) || acc[i][j][r] == 12345.f)) atomicAdd(p.dW + (size_t)n * p.lddw + k, acc[i][j][r]);
        }
      }
    }
}

template <int TN, int TK, bool SPARSE, int X3 = 0>
__global__ __launch_bounds__(256, 2) void mlp_dw_kernel(DwArgs p) {
  mlp_dw_body<TN, TK, SPARSE, X3>(p, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// Several weight-gradient launches of ONE template variant as one grid: nothing in a backward depends on a
// layer's dW, so the few-row stacks (FP modules, vote module, prediction heads: 15-20 launches of ~20 us)
// queue theirs and issue them together at the end (demf_mlp_gemm_bwd_dw_group) - the launches' ramps and tails
// overlap and the graph has that many fewer dependent nodes.
constexpr int DW_GROUP_MAX = 12;
struct DwGroup {
  int n;
  int start[DW_GROUP_MAX + 1];     // first block of job j
  int gx[DW_GROUP_MAX];            // its grid.x (row chunks); grid.y = (start[j+1] - start[j]) / gx
  DwArgs job[DW_GROUP_MAX];
};
template <int TN, int TK, bool SPARSE, int X3 = 0>
__global__ __launch_bounds__(256, 2) void mlp_dw_group_kernel(DwGroup g) {
  const int b = (int)blockIdx.x;
  int j = 0;
  while (j + 1 < g.n && b >= g.start[j + 1]) ++j;
  const int local = b - g.start[j], gx = g.gx[j];
  mlp_dw_body<TN, TK, SPARSE, X3>(g.job[j], local % gx, local / gx, gx);
}

// Counter sets {next tile per group, finished blocks} of the dynamically scheduled persistent
// launches.  A launch takes the next set of a ring; the last block of the launch leaves it zeroed, so
// a captured launch can be replayed with the set it was given.  Launches that share a set are
// 1024 launches apart on the calling thread's stream order (the MLP path is single-stream).
constexpr int SCHED_SLOTS = 1024;
__device__ int g_tile_sched[SCHED_INTS * SCHED_SLOTS];
int* sched_slot() {
  static int* base = nullptr;
  static std::atomic<unsigned> next{0};
  if (base == nullptr) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_tile_sched)) != hipSuccess) return nullptr;
    base = (int*)ptr;
  }
  if (env_int("DEMF_STATIC_TILES", 0)) return nullptr;     // A/B switch
  return base + SCHED_INTS * (next.fetch_add(1) % SCHED_SLOTS);
}

// Blocks of zeroed doubles for the replicated accumulators (csrc/bn_fin.h): as the counter sets, the next block of a
// ring; the launch's last workgroup leaves it zeroed.
constexpr int ACC_SLOTS = 128;
__device__ double g_accum_ring[(size_t)ACC_SLOT_DOUBLES * ACC_SLOTS];
double* accum_slot() {
  static double* base = nullptr;
  static std::atomic<unsigned> next{0};
  if (base == nullptr) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_accum_ring)) != hipSuccess) return nullptr;
    base = (double*)ptr;
  }
  if (!env_int("DEMF_ACC_REPL", 1)) return nullptr;     // A/B switch
  return base + (size_t)ACC_SLOT_DOUBLES * (next.fetch_add(1) % ACC_SLOTS);
}

static int mlp_grid(int R, int brows) {
  const int tiles = (R + brows - 1) / brows;
  const int cap = env_int("DEMF_GEMM_GRID", 256 * 2);   // persistent: two 256-thread blocks per CU
  return tiles < cap ? (tiles > 0 ? tiles : 1) : cap;
}

template <int PRO, bool STATS, bool POOL, bool RED, int CM>
static int launch_gemm_t(const MlpArgs& a, hipStream_t s);


// ---- few-row forward layers: 64 x 64 tiles, operands split ONCE at staging, double-buffered planes (round 6) ---------
// The FP modules, the vote module, the vote aggregation's second layer and the prediction heads run this file's
// forward GEMM on 2 048 ... 32 768 rows.  mlp_gemm_kernel walks a 32-column K step in ~4 200 cycles there (DESIGN_LOG
// section 3.2: the fp32 slab goes to LDS as it is and EVERY wave splits its own fragments, two barriers per step, 24
// MFMAs = 768 issue cycles): a 2 048 x 256 -> 128 layer is 8 dependent steps = 15 us for 0.07 GFLOP.  Here a
// workgroup of 4 waves owns a 64 x 64 output tile; per 32-wide K step every thread takes 8 consecutive k of one A
// row and of one W row (two float4 each, a full 128-byte line per 4 threads), applies the BN + ReLU prologue, splits
// them into the P bf16 planes and writes ONE 16-byte chunk per plane ([row][k] planes, XOR-swizzled chunks:
// fragment reads are conflict-free); the planes are double-buffered, so a step has ONE barrier, and the global rows
// of step i + 2 / i + 3 are in flight in registers.  A wave contracts 32 x 32 x 32 per step: 12 MFMAs (P = 3) on
// fragments read as ds_read_b128.  Epilogue: column statistics (fp64 atomics, one per column per workgroup), the raw
// output, the BatchNorm bookkeeping in the last workgroup.  Column tiles of a row block run on one XCD (A read once).
__device__ __forceinline__ int ft_swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int P>
__device__ __forceinline__ void ft_split8(const float4& a, const float4& b, uint4 (&o)[P]) {
  using v2f = float __attribute__((ext_vector_type(2)));
  using v2b = __bf16 __attribute__((ext_vector_type(2)));
  const v2f x[4] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
  unsigned w[4][P];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const v2b h = __builtin_convertvector(x[i], v2b);
    w[i][0] = __builtin_bit_cast(unsigned, h);
    if constexpr (P == 3) {
      const v2f r = x[i] - __builtin_convertvector(h, v2f);
      const v2b m = __builtin_convertvector(r, v2b);
      w[i][1] = __builtin_bit_cast(unsigned, m);
      const v2f l = r - __builtin_convertvector(m, v2f);
      w[i][2] = __builtin_bit_cast(unsigned, __builtin_convertvector(l, v2b));
    }
  }
#pragma unroll
  for (int q = 0; q < P; ++q) o[q] = make_uint4(w[0][q], w[1][q], w[2][q], w[3][q]);
}

template <int P, bool PRO>
__global__ __launch_bounds__(256, 2) void mlp_fwd_tile_kernel(MlpArgs p) {
  constexpr int PLANE = 64 * 64;                    // bytes: 64 rows x 32 k bf16
  constexpr int BUF = 2 * P * PLANE;                // A planes | W planes of one K step
  extern __shared__ __attribute__((aligned(16))) char ft_smem[];
  char* s_buf = ft_smem;                            // [2][BUF]
  float* s_vec = reinterpret_cast<float*>(ft_smem + 2 * BUF);   // [scale | shift] of the prologue (2K)
  float* s_red = s_vec + (PRO ? 2 * p.K : 0);       // [2 waves][2 stats][64 columns]
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    // all column tiles of a row block on ONE XCD, back to back (workgroups are dealt to the 8 XCDs round-robin)
    const int gx = gridDim.x, gy = gridDim.y;
    if ((gy & 7) == 0) {
      const int L = bx + gx * by;
      by = (L & 7) + 8 * (L / (8 * gx));
      bx = (L >> 3) % gx;
    }
  }
  const int m0 = by * 64, n0 = bx * 64;
  if constexpr (PRO) {
    for (int i = tid; i < 2 * p.K; i += 256) s_vec[i] = p.vec[i];
  }
  // staging map: row (tid >> 2) of the tile, k chunk (tid & 3) = 8 consecutive k
  const int srow = tid >> 2, sch = tid & 3;
  const int arow = min(m0 + srow, p.R - 1);         // rows beyond R: a valid address, zeroed in the transform
  const bool arow_ok = m0 + srow < p.R;
  const float* ga = p.X + (size_t)arow * p.ldx + 8 * sch;
  const float* gw = p.Bt + (size_t)(n0 + srow) * p.K + 8 * sch;
  const int nk = p.K / 32;
  // Register ring of FOUR K steps: a row that the layer below has just written is ~2 us away (another XCD's L2
  // or HBM), a step is ~0.5 us - with two stages every step waited for its rows.  (Stages are addressed
  // statically - the K loop is unrolled by four - or they would live in scratch.)
  float4 ra0[2], rw0[2], ra1[2], rw1[2], ra2[2], rw2[2], ra3[2], rw3[2];
  auto fetch = [&](float4 (&a4)[2], float4 (&w4)[2], int ks) {
    const float* a = ga + 32 * ks;
    const float* w = gw + 32 * ks;
    a4[0] = *reinterpret_cast<const float4*>(a);
    a4[1] = *reinterpret_cast<const float4*>(a + 4);
    w4[0] = *reinterpret_cast<const float4*>(w);
    w4[1] = *reinterpret_cast<const float4*>(w + 4);
  };
  auto commit = [&](const float4 (&a4)[2], const float4 (&w4)[2], int ks, char* buf) {
    float4 a0 = a4[0], a1 = a4[1];
    if constexpr (PRO) {
      const float4 c0 = *reinterpret_cast<const float4*>(s_vec + 32 * ks + 8 * sch);
      const float4 c1 = *reinterpret_cast<const float4*>(s_vec + 32 * ks + 8 * sch + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(s_vec + p.K + 32 * ks + 8 * sch);
      const float4 h1 = *reinterpret_cast<const float4*>(s_vec + p.K + 32 * ks + 8 * sch + 4);
      a0.x = fmaxf(0.f, __builtin_fmaf(a0.x, c0.x, h0.x)); a0.y = fmaxf(0.f, __builtin_fmaf(a0.y, c0.y, h0.y));
      a0.z = fmaxf(0.f, __builtin_fmaf(a0.z, c0.z, h0.z)); a0.w = fmaxf(0.f, __builtin_fmaf(a0.w, c0.w, h0.w));
      a1.x = fmaxf(0.f, __builtin_fmaf(a1.x, c1.x, h1.x)); a1.y = fmaxf(0.f, __builtin_fmaf(a1.y, c1.y, h1.y));
      a1.z = fmaxf(0.f, __builtin_fmaf(a1.z, c1.z, h1.z)); a1.w = fmaxf(0.f, __builtin_fmaf(a1.w, c1.w, h1.w));
    }
    if (!arow_ok) { a0 = make_float4(0.f, 0.f, 0.f, 0.f); a1 = a0; }
    uint4 pa[P], pw[P];
    ft_split8<P>(a0, a1, pa);
    ft_split8<P>(w4[0], w4[1], pw);
    const int off = ft_swz(srow, sch);
#pragma unroll
    for (int q = 0; q < P; ++q) {
      *reinterpret_cast<uint4*>(buf + q * PLANE + off) = pa[q];
      *reinterpret_cast<uint4*>(buf + (P + q) * PLANE + off) = pw[q];
    }
  };
  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  fetch(ra0, rw0, 0);
  if (nk > 1) fetch(ra1, rw1, 1);
  if (nk > 2) fetch(ra2, rw2, 2);
  if (nk > 3) fetch(ra3, rw3, 3);
  __syncthreads();                                  // the prologue vectors
  commit(ra0, rw0, 0, s_buf);
  if (nk > 4) fetch(ra0, rw0, 4);
  __syncthreads();
  // one K step: fragments of step ks out of buffer `cur`, step ks + 1 (registers an / wn) into buffer `nxt`,
  // the rows of step ks + 5 requested into the registers just freed
  auto step = [&](int ks, const char* cur, char* nxt, float4 (&an)[2], float4 (&wn4)[2]) {
    bf16x8 fa[2][P], fb[2][P];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int q = 0; q < P; ++q) {
        fa[s2][q] = *reinterpret_cast<const bf16x8*>(cur + q * PLANE + ft_swz(wm * 32 + lr, 2 * s2 + lh));
        fb[s2][q] = *reinterpret_cast<const bf16x8*>(cur + (P + q) * PLANE + ft_swz(wn * 32 + lr, 2 * s2 + lh));
      }
    // two accumulators (one per 16-wide half of the step): consecutive MFMAs are independent, and the staging
    // arithmetic of the NEXT step (issued below) runs on the vector ALU underneath them
    if constexpr (P == 3) {   // the six products of weight >= 2^-16, smallest first (as every three-term kernel)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][2], fb[0][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][2], fb[1][0], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][2], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][2], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][1], fb[0][1], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][1], fb[1][1], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][1], fb[0][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][1], fb[1][0], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][1], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][1], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][0], acc2, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][0], acc2, 0, 0, 0);
    }
    if (ks + 1 < nk) {
      commit(an, wn4, ks + 1, nxt);
      if (ks + 5 < nk) fetch(an, wn4, ks + 5);
    }
    if constexpr (P == 3) {
      // interleave: one MFMA, then a group of vector-ALU instructions of the staging code
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
      }
    }
    lds_barrier();
  };
  for (int ks = 0; ks < nk; ks += 4) {
    step(ks, s_buf, s_buf + BUF, ra1, rw1);                       // step 4j: stages step 4j + 1 out of ring slot 1
    if (ks + 1 < nk) step(ks + 1, s_buf + BUF, s_buf, ra2, rw2);
    if (ks + 2 < nk) step(ks + 2, s_buf, s_buf + BUF, ra3, rw3);
    if (ks + 3 < nk) step(ks + 3, s_buf + BUF, s_buf, ra0, rw0);
  }
  // ---- epilogue: C layout of 32x32: col = lr, row = (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
  const int n = n0 + wn * 32 + lr;
  float cs1 = 0.f, cs2 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < p.R) {
      p.Y[(size_t)m * p.ldy + n] = acc[r];
      cs1 += acc[r];
      cs2 = __builtin_fmaf(acc[r], acc[r], cs2);
    }
  }
  cs1 += __shfl_xor(cs1, 32);
  cs2 += __shfl_xor(cs2, 32);
  if (lh == 0) {
    s_red[(wm * 2 + 0) * 64 + wn * 32 + lr] = cs1;
    s_red[(wm * 2 + 1) * 64 + wn * 32 + lr] = cs2;
  }
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6, c = tid & 63;
    const float v = s_red[(0 * 2 + which) * 64 + c] + s_red[(1 * 2 + which) * 64 + c];
    double* sdst = p.fin.racc != nullptr ? repl_copy(p.fin.racc, p.N, (int)blockIdx.y) : p.stats;
    atomicAdd(sdst + (size_t)which * p.N + n0 + c, (double)v);
    // EVERY wave that issued sum atomics drains them before the barrier in front of the ticket: the barrier itself
    // does not wait for another wave's vector-memory counter (workgroup-scope release omits vmcnt(0) outside
    // threadgroup-split mode), and wave 1 carries all of this workgroup's sum of squares - taken late, the last
    // workgroup would finalise without its own 64 rows (measured: a 1e-3 shift of every gradient below the layer in
    // one run out of six) and the stragglers would land in the accumulator the finalize has just left zeroed.
  }
  // BatchNorm bookkeeping by the last workgroup (atomics only: see mlp_gemm_kernel)
  sync_drained();
  if (tid == 0) {
    const int total = (int)(gridDim.x * gridDim.y);
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int ngroups = total < SCHED_GROUPS ? total : SCHED_GROUPS;
    const int g = lin % SCHED_GROUPS;
    const int members = total / SCHED_GROUPS + (g < total % SCHED_GROUPS ? 1 : 0);
    int* t = p.fin.ticket + FIN_OFF;
    int last = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (atomicAdd(t + 1 + g, 1) == members - 1) {
      atomicExch(t + 1 + g, 0);
      if (atomicAdd(t, 1) == ngroups - 1) { atomicExch(t, 0); last = 1; }
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last) {
    const BnFin& f = p.fin;
    if (tid == 0 && f.nbt != nullptr) *f.nbt += 1;
    for (int c = tid; c < p.N; c += 256) {
      double s1 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(p.stats + c), 0ull));
      double s2 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(p.stats + p.N + c), 0ull));
      if (f.racc != nullptr) { s1 += repl_take(f.racc, p.N, c); s2 += repl_take(f.racc, p.N, p.N + c); }
      bn_finalize_channel(c, p.N, f.count, s1, s2, f.gamma, f.beta, f.eps, f.momentum, f.rmean, f.rvar, f.ss, f.mi,
                          f.conv_bias);
    }
  }
}

template <int P, bool PRO>
static int launch_fwd_tile(const MlpArgs& a, hipStream_t s) {
  const int lds = 2 * 2 * P * 64 * 64 + (PRO ? 2 * a.K * 4 : 0) + 2 * 2 * 64 * 4;
  static unsigned long long reserved = 0;
  if (lds > 64 * 1024 && !reserve_lds(reinterpret_cast<const void*>(&mlp_fwd_tile_kernel<P, PRO>), lds, &reserved))
    return -1000;
  const dim3 grid(a.N / 64, (a.R + 63) / 64);
  hipLaunchKernelGGL((mlp_fwd_tile_kernel<P, PRO>), grid, dim3(256), lds, s, a);
  return check_launch("mlp_fwd_tile");
}


// ---- few-row INPUT-GRADIENT launches on the same tile form -------------------------------------------------------
// dX (R x Kout) = dY (R x N) . W (N x Kout) with dY = gi*dZ + a*y + b built in the staging of the A operand (dZ: the
// dense upstream gradient G, or the pooled gradient dP routed by the arg-max slot; masked by the ReLU of THIS layer),
// W read as it is: a lane takes ONE output column and 8 consecutive reduction rows of a step - 8 coalesced dword
// loads - so its split values are exactly one 16-byte chunk of the [column][reduction] planes.  RED: the output tile
// is also layer l-1's activation gradient - its BN-backward sums are taken on the way out (Y_{l-1} read per
// element), the vectors by the last workgroup (csrc/bn_fin.h).  Reduction length p.K = this layer's channels N,
// output columns p.N; p.ldb = row stride of W.
template <int P, bool SPARSE, bool RED>
__global__ __launch_bounds__(256, 2) void mlp_dx_tile_kernel(MlpArgs p) {
  constexpr int PLANE = 64 * 64;
  constexpr int BUF = 2 * P * PLANE;
  extern __shared__ __attribute__((aligned(16))) char ft_smem[];
  char* s_buf = ft_smem;                            // [2][BUF]
  float* s_vec = reinterpret_cast<float*>(ft_smem + 2 * BUF);   // scale | shift | gi | a | b of THIS layer (5 x p.K)
  float* s_red = s_vec + 5 * p.K;                   // [2 waves][2 sums][64 columns]
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    if ((gy & 7) == 0) {
      const int L = bx + gx * by;
      by = (L & 7) + 8 * (L / (8 * gx));
      bx = (L >> 3) % gx;
    }
  }
  const int m0 = by * 64, n0 = bx * 64;
  for (int i = tid; i < 5 * p.K; i += 256) s_vec[i] = p.vec[i];
  // A staging: row (tid >> 2) of the tile, 8 consecutive channels of the step
  const int srow = tid >> 2, sch = tid & 3;
  const int arow = min(m0 + srow, p.R - 1);
  const bool arow_ok = m0 + srow < p.R;
  const float* gy = p.X + (size_t)arow * p.ldx + 8 * sch;                     // Y_l
  const float* gg = SPARSE ? nullptr : p.G + (size_t)arow * p.ldx + 8 * sch;   // dense upstream gradient
  const int rp = SPARSE ? arow / p.ns : 0;
  const int slot = SPARSE ? arow - rp * p.ns : 0;
  const float* gdp = SPARSE ? p.dP + (size_t)rp * p.K + 8 * sch : nullptr;
  const int* garg = SPARSE ? p.arg + (size_t)rp * p.K + 8 * sch : nullptr;
  // B staging: output column (tid & 63), reduction rows 8 (tid >> 6) .. + 7 of the step
  const int bcol = tid & 63, bq = tid >> 6;
  const float* gw = p.Bt + (size_t)(8 * bq) * p.ldb + n0 + bcol;
  const int nk = p.K / 32;
  struct Stage { float4 y[2], g[2]; int4 a[2]; float w[8]; };
  Stage st0, st1;
  auto fetch = [&](Stage& st, int ks) {
    st.y[0] = *reinterpret_cast<const float4*>(gy + 32 * ks);
    st.y[1] = *reinterpret_cast<const float4*>(gy + 32 * ks + 4);
    if constexpr (SPARSE) {
      st.g[0] = *reinterpret_cast<const float4*>(gdp + 32 * ks);
      st.g[1] = *reinterpret_cast<const float4*>(gdp + 32 * ks + 4);
      st.a[0] = *reinterpret_cast<const int4*>(garg + 32 * ks);
      st.a[1] = *reinterpret_cast<const int4*>(garg + 32 * ks + 4);
    } else {
      st.g[0] = *reinterpret_cast<const float4*>(gg + 32 * ks);
      st.g[1] = *reinterpret_cast<const float4*>(gg + 32 * ks + 4);
    }
    const float* w = gw + (size_t)(32 * ks) * p.ldb;
#pragma unroll
    for (int e = 0; e < 8; ++e) st.w[e] = w[(size_t)e * p.ldb];
  };
  auto commit = [&](const Stage& st, int ks, char* buf) {
    const int c0 = 32 * ks + 8 * sch;
    const float y[8] = {st.y[0].x, st.y[0].y, st.y[0].z, st.y[0].w, st.y[1].x, st.y[1].y, st.y[1].z, st.y[1].w};
    const float g[8] = {st.g[0].x, st.g[0].y, st.g[0].z, st.g[0].w, st.g[1].x, st.g[1].y, st.g[1].z, st.g[1].w};
    const int ar[8] = {st.a[0].x, st.a[0].y, st.a[0].z, st.a[0].w, st.a[1].x, st.a[1].y, st.a[1].z, st.a[1].w};
    float dy[8];
    float v5[5][8];                                   // scale, shift, gi, a, b of the thread's 8 channels: float4 reads
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const float4 lo = *reinterpret_cast<const float4*>(s_vec + q * p.K + c0);
      const float4 hi = *reinterpret_cast<const float4*>(s_vec + q * p.K + c0 + 4);
      v5[q][0] = lo.x; v5[q][1] = lo.y; v5[q][2] = lo.z; v5[q][3] = lo.w;
      v5[q][4] = hi.x; v5[q][5] = hi.y; v5[q][6] = hi.z; v5[q][7] = hi.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool on = __builtin_fmaf(y[e], v5[0][e], v5[1][e]) > 0.f;
      float dz;
      if constexpr (SPARSE) dz = (on && ar[e] == slot) ? g[e] : 0.f;
      else dz = on ? g[e] : 0.f;
      dy[e] = arow_ok ? __builtin_fmaf(v5[2][e], dz, __builtin_fmaf(v5[3][e], y[e], v5[4][e])) : 0.f;
    }
    uint4 pa[P], pw[P];
    ft_split8<P>(make_float4(dy[0], dy[1], dy[2], dy[3]), make_float4(dy[4], dy[5], dy[6], dy[7]), pa);
    ft_split8<P>(make_float4(st.w[0], st.w[1], st.w[2], st.w[3]), make_float4(st.w[4], st.w[5], st.w[6], st.w[7]), pw);
    const int offa = ft_swz(srow, sch), offb = ft_swz(bcol, bq);
#pragma unroll
    for (int q = 0; q < P; ++q) {
      *reinterpret_cast<uint4*>(buf + q * PLANE + offa) = pa[q];
      *reinterpret_cast<uint4*>(buf + (P + q) * PLANE + offb) = pw[q];
    }
  };
  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  fetch(st0, 0);
  if (nk > 1) fetch(st1, 1);
  __syncthreads();                                  // the layer's vectors
  commit(st0, 0, s_buf);
  if (nk > 2) fetch(st0, 2);
  __syncthreads();
  auto step = [&](int ks, const char* cur, char* nxt, Stage& sn) {
    bf16x8 fa[2][P], fb[2][P];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int q = 0; q < P; ++q) {
        fa[s2][q] = *reinterpret_cast<const bf16x8*>(cur + q * PLANE + ft_swz(wm * 32 + lr, 2 * s2 + lh));
        fb[s2][q] = *reinterpret_cast<const bf16x8*>(cur + (P + q) * PLANE + ft_swz(wn * 32 + lr, 2 * s2 + lh));
      }
    if constexpr (P == 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][2], fb[0][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][2], fb[1][0], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][2], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][2], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][1], fb[0][1], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][1], fb[1][1], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][1], fb[0][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][1], fb[1][0], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][1], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][1], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][0], acc2, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][0], acc2, 0, 0, 0);
    }
    if (ks + 1 < nk) {
      commit(sn, ks + 1, nxt);
      if (ks + 3 < nk) fetch(sn, ks + 3);
    }
    lds_barrier();
  };
  // RED: this lane's 16 values of Y_{l-1} (the rows / column of its accumulator registers) are requested now and
  // arrive underneath the K loop instead of as 16 exposed loads behind it
  const int n = n0 + wn * 32 + lr;                  // output column of this launch
  float yprev[16];
  if constexpr (RED) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, p.R - 1);
      yprev[r] = p.fY[(size_t)m * p.fld + p.fc0 + n];
    }
  }
  for (int ks = 0; ks < nk; ks += 2) {
    step(ks, s_buf, s_buf + BUF, st1);
    if (ks + 1 < nk) step(ks + 1, s_buf + BUF, s_buf, st0);
  }
  // ---- epilogue
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
  float sc0 = 0.f, sh0 = 0.f, mu0 = 0.f, is0 = 0.f;
  if constexpr (RED) {
    sc0 = p.fss[p.fc0 + n]; sh0 = p.fss[p.fld + p.fc0 + n];
    mu0 = p.fmi[p.fc0 + n]; is0 = p.fmi[p.fld + p.fc0 + n];
  }
  float es1 = 0.f, es2 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < p.R) {
      p.Y[(size_t)m * p.ldy + n] = acc[r];
      if constexpr (RED) {
        const float y = yprev[r];
        const float dz = __builtin_fmaf(y, sc0, sh0) > 0.f ? acc[r] : 0.f;
        es1 += dz;
        es2 = __builtin_fmaf(dz, (y - mu0) * is0, es2);
      }
    }
  }
  if constexpr (RED) {
    es1 += __shfl_xor(es1, 32);
    es2 += __shfl_xor(es2, 32);
    if (lh == 0) {
      s_red[(wm * 2 + 0) * 64 + wn * 32 + lr] = es1;
      s_red[(wm * 2 + 1) * 64 + wn * 32 + lr] = es2;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const float v = s_red[(0 * 2 + which) * 64 + c] + s_red[(1 * 2 + which) * 64 + c];
      double* sdst = p.vfin.ticket != nullptr && p.vfin.racc != nullptr ? repl_copy(p.vfin.racc, p.fld, (int)blockIdx.y) : p.stats;
      atomicAdd(sdst + (size_t)which * p.fld + p.fc0 + n0 + c, (double)v);
    }
    if (p.vfin.ticket != nullptr) {
      sync_drained();
      if (tid == 0)
        s_last = last_workgroup(p.vfin.ticket, (int)(gridDim.x * gridDim.y), (int)(blockIdx.y * gridDim.x + blockIdx.x));
      __syncthreads();
      if (s_last) bn_vec_finalize(p.vfin, p.fld, p.fc0, p.N, p.stats, tid, 256);
    }
  }
}

template <int P, bool SPARSE, bool RED>
static int launch_dx_tile(const MlpArgs& a, hipStream_t s) {
  const int lds = 2 * 2 * P * 64 * 64 + 5 * a.K * 4 + 2 * 2 * 64 * 4;
  static unsigned long long reserved = 0;
  if (lds > 64 * 1024 && !reserve_lds(reinterpret_cast<const void*>(&mlp_dx_tile_kernel<P, SPARSE, RED>), lds, &reserved))
    return -1000;
  const dim3 grid(a.N / 64, (a.R + 63) / 64);
  hipLaunchKernelGGL((mlp_dx_tile_kernel<P, SPARSE, RED>), grid, dim3(256), lds, s, a);
  return check_launch("mlp_dx_tile");
}

template <int PRO, bool STATS, bool POOL = false, bool RED = false>
static int launch_gemm(const MlpArgs& a, hipStream_t s) {
  switch (compute_mode()) {
    case 1: return launch_gemm_t<PRO, STATS, POOL, RED, 1>(a, s);
    case 2: {
      // A/B switch: DEMF_X3_MASK bit 0 = forward launches, bit 1 = input-gradient, bit 2 = weight-gradient
      static const int mask = env_int("DEMF_X3_MASK", 7);
      if (mask & (PRO >= PRO_DY_DENSE ? 2 : 1)) return launch_gemm_t<PRO, STATS, POOL, RED, 2>(a, s);
      return launch_gemm_t<PRO, STATS, POOL, RED, 0>(a, s);
    }
    default: return launch_gemm_t<PRO, STATS, POOL, RED, 0>(a, s);
  }
}

template <int PRO, bool STATS, bool POOL, bool RED, int BF16>
static int launch_gemm_t(const MlpArgs& a, hipStream_t s) {
  if constexpr ((BF16 == 1 || BF16 == 2) && !RED && STATS && PRO == PRO_BNRELU) {
    // producer / consumer forward for 128-channel inputs (mlp_fwd_pc_kernel): SA2-4 and the vote aggregation
    static const int pc_on = env_int("DEMF_FWD_PC", 1);
    static const int pc_min_r = env_int("DEMF_FWD_PC_MIN_R", 16384);
    const bool pc_sel = !POOL || (a.pmin == nullptr && a.fin.gamma != nullptr && a.pmax && a.amax);
    if (pc_on && a.st == 0 && a.K == PC_K && (a.N == 128 || a.N == 256) && a.ldx == PC_K && a.ldy == a.N &&
        a.ldb == 0 && pc_sel && a.R >= pc_min_r && a.R % PC_ROWS == 0 && a.vec != nullptr && a.stats != nullptr &&
        (!POOL || a.ns == 16 || a.ns == 32) && (a.fin.ss == nullptr || a.fin.ticket != nullptr)) {
      static const int cus = [] { const char* v = getenv("DEMF_PERSIST_CUS"); return v ? atoi(v) : 240; }();
      const int nslab = a.R / PC_ROWS, halves = a.N / 128;
      int gh = cus / halves;                       // workgroups per column half
      if (gh > nslab) gh = nslab;
      if (halves == 2) gh = gh / 8 * 8 > 0 ? gh / 8 * 8 : 8;   // interleaved halves: runs of 8 blocks
#define PCGO(CMv)                                                                                              \
      do {                                                                                                     \
        constexpr int P = CMv == 2 ? 3 : (CMv == 3 ? 2 : 1);                                                   \
        const size_t bytes = (size_t)2 * P * PC_ROWS * PC_KB;                                                  \
        static bool configured = false;                                                                        \
        if (!configured) {                                                                                     \
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_pc_kernel<CMv, POOL>),                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {     \
            set_error("mlp_fwd_pc: cannot reserve %zu bytes of LDS", bytes);                                   \
            return DEMF_ELAUNCH;                                                                               \
          }                                                                                                    \
          configured = true;                                                                                   \
        }                                                                                                      \
        hipLaunchKernelGGL((mlp_fwd_pc_kernel<CMv, POOL>), dim3(gh * halves), dim3(512), bytes, s, a);         \
      } while (0)
      // (fp32 results: two fp16 terms, three products - csrc/common.h - unless DEMF_F16_TERMS=0: three bf16 terms, six)
      if constexpr (BF16 == 2) { if (f16_terms()) PCGO(3); else PCGO(2); }
      else PCGO(BF16);
#undef PCGO
      return check_launch("mlp_fwd_pc");
    }
    // weight-resident, barrier-free forward (mlp_fwd_res_kernel): 64-channel inputs, N = 64 / 128
    static const int fr_on = env_int("DEMF_FWD_RES", 1);
    const bool sel = !POOL || (a.pmin == nullptr && a.fin.gamma != nullptr);
    if ((fr_on || a.st) && a.K == 64 && (a.N == 64 || a.N == 128) && (a.ldx == 64 || (a.st == 8 && a.ldx == 4)) &&
        a.ldy == a.N && a.ldb == 0 && sel &&
        a.R >= 64 * 256 && (!POOL || ((a.ns == 16 || a.ns == 32 || a.ns == 64) && a.R % 64 == 0)) &&
        (a.fin.ss == nullptr || a.fin.ticket != nullptr)) {
      const bool h2 = BF16 == 2 && f16_terms();          // two fp16 terms instead of three bf16 ones
      const int P = h2 ? 2 : (BF16 == 2 ? 3 : 1);
      const int ntn = a.N / 32;
      const size_t bytes = (size_t)P * a.N * 128 + (size_t)FR_NW * P * 32 * 128 + sizeof(float) * (2 * 64 + FR_NW * ntn * 64);
      const int nunit = (a.R + 63) / 64;
      int gx = (nunit + FR_NW - 1) / FR_NW;
      // one 8-wave workgroup per CU, on 240 of the 256: the rest is left to the resident FPS chain of
      // the next batch (csrc/mlp_bwd.hip launch_fused has the measurements)
      static const int cus = env_int("DEMF_PERSIST_CUS", 240);
      if (gx > cus) gx = cus;
#define FRGO_(NTNv, STv, CMv)                                                                               \
      do {                                                                                                  \
        static bool configured = false;                                                                     \
        if (!configured) {                                                                                  \
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_res_kernel<NTNv, 2, POOL, CMv, STv>),   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {   \
            set_error("mlp_fwd_res: cannot reserve %zu bytes of LDS", bytes);                                \
            return DEMF_ELAUNCH;                                                                             \
          }                                                                                                 \
          configured = true;                                                                                \
        }                                                                                                   \
        hipLaunchKernelGGL((mlp_fwd_res_kernel<NTNv, 2, POOL, CMv, STv>), dim3(gx), dim3(64 * FR_NW), bytes, s, a); \
      } while (0)
#define FRGO(NTNv, STv)                                                                                     \
      do {                                                                                                  \
        if constexpr (BF16 == 2) { if (h2) FRGO_(NTNv, STv, 3); else FRGO_(NTNv, STv, 2); }                \
        else FRGO_(NTNv, STv, BF16);                                                                        \
      } while (0)
      if (a.st == 0) {
        if (ntn == 2) FRGO(2, 0); else FRGO(4, 0);
      } else {
        // bf16 storage: the two forms SA1 uses (fp32 rows in -> bf16 rows out; bf16 in -> bf16 out, pooled)
        bool done = false;
        if constexpr (BF16 == 1 && !POOL) {
          if (a.st == 2 && ntn == 2) { FRGO(2, 2); done = true; }
        }
        if constexpr (BF16 == 1 && POOL) {
          if (a.st == 3 && ntn == 4) { FRGO(4, 3); done = true; }
        }
        if constexpr (POOL) {
          // no-store form (either mode): SA1's last layer, 64-row groups
          if (a.st == 4 && ntn == 4 && a.ns == 64) { FRGO(4, 4); done = true; }
        }
        if constexpr (!POOL) {
          // input rows of the layer below instead of its output (either mode): SA1's second layer
          if (a.st == 8 && ntn == 2 && a.W0 != nullptr) { FRGO(2, 8); done = true; }
        }
        if (!done) {
          set_error("mlp_fwd_res: bf16 storage form st=%d N=%d pool=%d mode=%d not built", a.st, a.N, (int)POOL, BF16);
          return DEMF_EUNSUPPORTED;
        }
      }
#undef FRGO
#undef FRGO_
      return check_launch("mlp_fwd_res");
    }
  }
  if (a.st != 0) {
    set_error("mlp_gemm: bf16 row storage (st=%d) is only built into the weight-resident forward (K = 64, "
              "N = 64 / 128, R >= 16384, bf16 compute mode)", a.st);
    return DEMF_EUNSUPPORTED;
  }
  if constexpr ((BF16 == 1 || BF16 == 2) && !POOL && !STATS && (PRO == PRO_DY_DENSE || PRO == PRO_DY_SPARSE)) {
    // few-row input-gradient launches on the tile kernel (mlp_dx_tile_kernel); A/B switch DEMF_DX_TILE
    static const int dxt_on = env_int("DEMF_DX_TILE", 1);
    static const int dxt_max_r = env_int("DEMF_DX_TILE_MAX_R", 16384);
    constexpr bool SP = PRO == PRO_DY_SPARSE;
    if (dxt_on && a.R <= dxt_max_r && a.K % 32 == 0 && a.K >= 64 && a.K <= 1024 && a.N % 64 == 0 && a.ldx % 4 == 0 &&
        a.ldb > 0 && a.vec != nullptr && ((uintptr_t)a.X % 16 == 0) && (SP ? (a.dP && a.arg && a.ns >= 1) : (a.G != nullptr)) &&
        (SP ? ((uintptr_t)a.dP % 16 == 0 && (uintptr_t)a.arg % 16 == 0) : ((uintptr_t)a.G % 16 == 0)) &&
        (!RED || (a.fY && a.fss && a.fmi && a.stats))) {
      constexpr int P = BF16 == 2 ? 3 : 1;
      const int rc = launch_dx_tile<P, SP, RED>(a, s);
      if (rc != -1000) return rc;
    }
  }
  if constexpr ((BF16 == 1 || BF16 == 2) && !RED && !POOL && STATS && (PRO == PRO_BNRELU || PRO == PRO_NONE)) {
    // few-row forward layers on the 64 x 64 tile kernel (mlp_fwd_tile_kernel); A/B switch DEMF_FWD_TILE
    static const int tile_on = env_int("DEMF_FWD_TILE", 1);
    // (measured, tools/fewrow_fwd_micro.py: 4 096 x 512 -> 256 26.3 -> 17.3 us, 8 192 x 512 -> 256 33.3 -> 23.7,
    // 2 048 x 256 -> 128 14.3 -> 10.2; at 32 768 rows mlp_gemm_kernel's 128-row tiles are level again: 43.6 vs 45.6)
    static const int tile_max_r = env_int("DEMF_FWD_TILE_MAX_R", 16384);
    if (tile_on && a.R <= tile_max_r && a.K % 32 == 0 && a.K >= 64 && a.K <= 1024 && a.N % 64 == 0 && a.ldx % 4 == 0 &&
        a.ldb == 0 && a.ldy % 1 == 0 && a.stats != nullptr && a.fin.ss != nullptr && a.fin.ticket != nullptr &&
        ((uintptr_t)a.X % 16 == 0) && ((uintptr_t)a.Bt % 16 == 0) && (PRO == PRO_NONE || a.vec != nullptr)) {
      constexpr int P = BF16 == 2 ? 3 : 1;
      const int rc = launch_fwd_tile<P, PRO == PRO_BNRELU>(a, s);
      if (rc != -1000) return rc;
    }
  }
  const dim3 block(256);
  // Two 32-row tiles per wave (256-row block tiles) while the accumulators + the raw prefetch fit
  // in 256 VGPRs: up to 4 column tiles for the forward prologues, up to 2 for the backward ones
  // (whose raw prefetch is 2-3x wider).  Backward launches are limited to 128 columns.
  // Launches with few rows (FP / vote / head layers: 4-8 k rows) would fill only a few CUs with
  // full-width tiles, so their columns are split over blockIdx.y (A is re-read; it is tiny).
  constexpr int RT2_MAX = PRO >= PRO_DY_DENSE ? 2 : 4;
  constexpr int NT_MAX = PRO >= PRO_DY_DENSE ? 4 : 8;
  const int nt = (a.N + 31) / 32;
  // (non-pooled launches wider than NT_MAX column tiles split their columns over blockIdx.y below)
  if (nt < 1 || nt > (POOL ? NT_MAX : 16)) {
    set_error("mlp_gemm: N=%d unsupported for this prologue (max %d columns per launch)", a.N,
              (POOL ? NT_MAX : 16) * 32);
    return DEMF_EUNSUPPORTED;
  }
  if constexpr (POOL) {
    // pooled epilogue: 64-row wave tiles for ns = 64 (a group must live in one wave), 32-row ones
    // otherwise; at most 2 (resp. 4) column tiles per block so that the accumulators, the
    // prefetch and the epilogue's temporaries fit in 256 VGPRs - wider outputs go to blockIdx.y
    // ns = 64: 64-row wave tiles x <= 2 column tiles (column halves co-scheduled), or - when all the
    // columns fit 4 tiles - 32-row wave tiles x 4 column tiles with the group merged across a wave pair
    const bool half64 = a.ns == 64 && nt <= 4 && !env_int("DEMF_POOL_HALVES", 0);
    const bool rt2 = a.ns == 64 && !half64;
    const int ntl = rt2 ? (nt < 2 ? nt : 2) : (nt < 4 ? nt : 4);
    const int ys = (nt + ntl - 1) / ntl;
    int gx = mlp_grid(a.R, rt2 ? 256 : 128);
    MlpArgs a2 = a;
    dim3 grid(gx, ys);
    if (ys == 2) {                       // interleaved halves: multiple of 16 blocks
      gx = ((gx + 7) / 8) * 8;
      a2.halves = 2;
      grid = dim3(2 * gx, 1);
    }
    // persistent pooled launches take their tiles dynamically too (not the interleaved-halves form,
    // whose block -> (tile, half) mapping is static)
    {
      const int brows = rt2 ? 256 : 128;
      const int tiles = (a.R + brows - 1) / brows;
      static const int pool_dyn = env_int("DEMF_POOL_DYN", 1);
      if (pool_dyn && ys == 1 && tiles > gx && a.K > MLP_BK && gx % (8 * SCHED_GROUPS) == 0)
        a2.sched = sched_slot();
    }
    // SEL: only the extremum the sign of gamma selects (pmin / amin not given, gamma known)
    const bool sel = a.pmin == nullptr && a.fin.gamma != nullptr;
#define PGO(NTv, RTv)                                                                                       \
    do {                                                                                                    \
      if (sel) hipLaunchKernelGGL((mlp_gemm_kernel<NTv, RTv, PRO, STATS, true, false, false, BF16, true>), grid, block, 0, s, a2); \
      else hipLaunchKernelGGL((mlp_gemm_kernel<NTv, RTv, PRO, STATS, true, false, false, BF16>), grid, block, 0, s, a2);           \
    } while (0)
    if (rt2) { if (ntl == 1) PGO(1, 2); else PGO(2, 2); }
    else if (ntl == 1) PGO(1, 1);
    else if (ntl == 2) PGO(2, 1);
    else if (ntl == 3) PGO(3, 1);
    else PGO(4, 1);
#undef PGO
    return check_launch("mlp_gemm_pool");
  } else {
  const int tiles1 = (a.R + 127) / 128;
  // (three-term mode: more than 4 column tiles per block would not leave LDS for two blocks per CU -
  // the columns go to blockIdx.y instead of the launch falling back to the fp32 MFMA)
  static const int wide_split = env_int("DEMF_X3_WIDE_SPLIT", 1);
  if ((tiles1 < 192 && nt > 1) || nt > NT_MAX || (BF16 == 2 && nt > 4 && wide_split)) {
    static const int split_target = env_int("DEMF_SPLIT_TARGET", 256);   // blocks the column split aims at
    int ysplit = (split_target + tiles1 - 1) / tiles1;
    if (ysplit < (nt + 3) / 4) ysplit = (nt + 3) / 4;     // at most 4 column tiles per block
    if (ysplit > nt) ysplit = nt;
    const int ntl = (nt + ysplit - 1) / ysplit;          // column tiles per block
    const dim3 grid(tiles1, (nt + ntl - 1) / ntl);
    switch (ntl) {
#define SPLIT(NTv)                                                                              \
      case NTv: hipLaunchKernelGGL((mlp_gemm_kernel<NTv, 1, PRO, STATS, POOL, false, RED, BF16>), grid, block, 0, s, a); break;
      SPLIT(1) SPLIT(2) SPLIT(3) SPLIT(4)
#undef SPLIT
      default: break;
    }
    if (ntl <= 4) return check_launch("mlp_gemm");
  }
#define GO(NTv, RTv)                                                                            \
  do {                                                                                          \
    MlpArgs a3 = a;                                                                             \
    const int gxv = mlp_grid(a.R, 128 * RTv);                                                   \
    const int tilesv = (a.R + 128 * RTv - 1) / (128 * RTv);                                     \
    if (tilesv > gxv && a.K > MLP_BK && gxv % (8 * SCHED_GROUPS) == 0) a3.sched = sched_slot();                                \
    hipLaunchKernelGGL((mlp_gemm_kernel<NTv, RTv, PRO, STATS, POOL, false, RED, BF16>), dim3(gxv), block, 0, s, a3); \
  } while (0)
#define CASE(NTv)                                                                               \
  case NTv:                                                                                     \
    if constexpr (NTv <= NT_MAX && (BF16 != 2 || NTv <= 4)) {                                   \
      if constexpr (NTv <= RT2_MAX && !RED) GO(NTv, 2); else GO(NTv, 1);                        \
    }                                                                                           \
    break;
  // the split Bt slab of more than 4 column tiles would not leave room for two blocks per CU
  if constexpr (BF16 == 2) {
    if (nt > 4) return launch_gemm_t<PRO, STATS, POOL, RED, 0>(a, s);
  }
  switch (nt) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
  }
#undef CASE
#undef GO
  return check_launch("mlp_gemm");
  }
}

static int mlp_check(int R, int K, int N, int ldx) {
  DEMF_REQUIRE(R >= 0 && K >= 4 && K % 4 == 0 && K <= MLP_MAXK && N >= 1 && N <= 256 && ldx >= K &&
                   ldx % 4 == 0,
               "mlp: bad sizes R=%d K=%d N=%d ldx=%d (K%%4==0, K<=512, 1<=N<=256)", R, K, N, ldx);
  return DEMF_OK;
}

}  // namespace demf

using namespace demf;


extern "C" int demf_set_compute_dtype(int mode) {
  DEMF_REQUIRE(mode >= 0 && mode <= 2, "set_compute_dtype: 0 = fp32, 1 = bf16, 2 = fp32 as three bf16 terms");
  g_compute_mode.store(mode);
  return DEMF_OK;
}

extern "C" int demf_set_f16_terms(int on) {
  g_f16_terms.store(on ? 1 : 0);
  return DEMF_OK;
}

extern "C" int demf_ctx_push(const demf_ctx* ctx) {
  DEMF_REQUIRE(ctx != nullptr, "ctx_push: null context");
  DEMF_REQUIRE(ctx->compute_mode >= -1 && ctx->compute_mode <= 2,
               "ctx_push: compute_mode -1 (process default), 0 fp32, 1 bf16, 2 fp32 as three bf16 terms");
  DEMF_REQUIRE(tl_mode_top < CTX_DEPTH, "ctx_push: more than %d nested contexts on this thread", CTX_DEPTH);
  tl_mode_stack[tl_mode_top++] = ctx->compute_mode;
  return DEMF_OK;
}

extern "C" int demf_ctx_pop(void) {
  DEMF_REQUIRE(tl_mode_top > 0, "ctx_pop: no context pushed on this thread");
  --tl_mode_top;
  return DEMF_OK;
}

extern "C" int demf_get_compute_dtype(void) { return compute_mode(); }

namespace {
struct CtxScope {          // the *_ctx entry points: the context for exactly one call
  bool pushed;
  explicit CtxScope(const demf_ctx* c) : pushed(c != nullptr && demf_ctx_push(c) == DEMF_OK) {}
  ~CtxScope() { if (pushed) demf_ctx_pop(); }
};
}  // namespace

extern "C" int demf_mlp_gemm_fwd(int, int, int, int, const float*, const float*, const float*, float*, double*,
                                 demf_stream_t);
extern "C" int demf_mlp_gemm_fwd_ctx(const demf_ctx* ctx, int R, int K, int N, int ldx, const float* X,
                                     const float* pro_scale_shift, const float* Wt, float* Y, double* stats,
                                     demf_stream_t stream) {
  DEMF_REQUIRE(ctx != nullptr, "mlp_gemm_fwd_ctx: null context");
  CtxScope scope(ctx);
  if (!scope.pushed) return DEMF_EINVAL;
  return demf_mlp_gemm_fwd(R, K, N, ldx, X, pro_scale_shift, Wt, Y, stats, stream);
}

extern "C" int demf_gemm_f32(const demf_gemm_desc*, demf_stream_t);
extern "C" int demf_gemm_f32_ctx(const demf_ctx* ctx, const demf_gemm_desc* desc, demf_stream_t stream) {
  DEMF_REQUIRE(ctx != nullptr, "gemm_f32_ctx: null context");
  CtxScope scope(ctx);
  if (!scope.pushed) return DEMF_EINVAL;
  return demf_gemm_f32(desc, stream);
}

extern "C" int demf_gemm_group_f32(const demf_gemm_desc*, int, demf_stream_t);
extern "C" int demf_gemm_group_f32_ctx(const demf_ctx* ctx, const demf_gemm_desc* descs, int n, demf_stream_t stream) {
  DEMF_REQUIRE(ctx != nullptr, "gemm_group_f32_ctx: null context");
  CtxScope scope(ctx);
  if (!scope.pushed) return DEMF_EINVAL;
  return demf_gemm_group_f32(descs, n, stream);
}

extern "C" int demf_mlp_gemm_fwd(int R, int K, int N, int ldx, const float* X,
                                 const float* pro_scale_shift, const float* Wt, float* Y,
                                 double* stats, demf_stream_t stream) {
  if (int e = mlp_check(R, K, N, ldx)) return e;
  if (R == 0) return DEMF_OK;
  DEMF_REQUIRE(X && Wt && Y, "mlp_gemm_fwd: null pointer");
  MlpArgs a{};
  a.R = R; a.K = K; a.N = N; a.ldx = ldx; a.ldy = N; a.X = X; a.vec = pro_scale_shift; a.Bt = Wt;
  a.Y = Y; a.stats = stats;
  hipStream_t s = (hipStream_t)stream;
  if (pro_scale_shift)
    return stats ? launch_gemm<PRO_BNRELU, true>(a, s) : launch_gemm<PRO_BNRELU, false>(a, s);
  return stats ? launch_gemm<PRO_NONE, true>(a, s) : launch_gemm<PRO_NONE, false>(a, s);
}

// Same GEMM with the fused max-pool epilogue (last layer of a PointSAModule stack): besides Y
// and the statistics it emits, per group of ns rows and column, max / min of the raw output and
// their row offsets.  Supported: ns 16 / 32 / 64; anything else returns DEMF_EUNSUPPORTED (callers
// then run demf_bnrelu_maxpool_fwd on Y).
extern "C" int demf_mlp_gemm_fwd_pool(int R, int K, int N, int ldx, const float* X,
                                      const float* pro_scale_shift, const float* Wt, float* Y,
                                      double* stats, int ns, float* pmax, float* pmin, int* amax,
                                      int* amin, demf_stream_t stream) {
  if (int e = mlp_check(R, K, N, ldx)) return e;
  const bool ok = (ns == 16 || ns == 32 || ns == 64) && R % ns == 0;
  if (!ok) {
    set_error("mlp_gemm_fwd_pool: ns=%d with R=%d N=%d is not fused", ns, R, N);
    return DEMF_EUNSUPPORTED;
  }
  if (R == 0) return DEMF_OK;
  DEMF_REQUIRE(X && Wt && Y && pro_scale_shift && stats && pmax && pmin && amax && amin,
               "mlp_gemm_fwd_pool: null pointer");
  MlpArgs a{};
  a.R = R; a.K = K; a.N = N; a.ldx = ldx; a.ldy = N; a.X = X; a.vec = pro_scale_shift; a.Bt = Wt;
  a.Y = Y; a.stats = stats; a.ns = ns; a.pmax = pmax; a.pmin = pmin; a.amax = amax; a.amin = amin;
  return launch_gemm<PRO_BNRELU, true, true>(a, (hipStream_t)stream);
}

// forward launch + train-mode BN bookkeeping: in the kernel's last workgroup when a counter set is
// available, as a separate launch otherwise (DEMF_STATIC_TILES=1, DEMF_NO_FIN=1)
template <typename Launch>
static int launch_with_finalize(MlpArgs& a, BnFin fin, Launch launch, hipStream_t s) {
  static const int off = env_int("DEMF_NO_FIN", 0);
  fin.ticket = off ? nullptr : sched_slot();
  // (replicated accumulators: measured neutral on the forward launches - their workgroups do not end together (the
  //  persistent SA kernels) or the 8 x 2N exchanges of the finalize cost what the shorter queue saves (the 64 x 64-tile
  //  kernel) - so off unless DEMF_ACC_REPL_FWD=1; the row-reduction kernels of the BN backward gain 2-5 us each)
  fin.racc = fin.ticket != nullptr && a.N <= ACC_MAX_N && env_int("DEMF_ACC_REPL_FWD", 0) ? accum_slot() : nullptr;
  a.fin = fin;
  if (fin.ticket != nullptr) return launch(a);
  a.fin.ss = nullptr;                     // (gamma stays: the pooled epilogue selects by its sign)
  if (int e = launch(a)) return e;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(a.N, 256)), dim3(256), 0, s, a.N, fin.count, a.stats,
                     fin.gamma, fin.beta, fin.eps, fin.momentum, fin.rmean, fin.rvar, fin.nbt, fin.ss,
                     fin.mi, fin.conv_bias);
  return check_launch("bn_finalize");
}

static int fin_check(long long count, const float* gamma, const float* beta, const float* ss,
                     const float* mi, const double* stats) {
  DEMF_REQUIRE(count >= 1 && gamma && beta && ss && mi && stats, "mlp_gemm_fwd_bn: bad BN arguments");
  return DEMF_OK;
}

static int mlp_gemm_fwd_bn_impl(int R, int K, int N, int ldx, const float* X,
                                const float* pro_scale_shift, const float* Wt, float* Y,
                                double* stats, const float* gamma, const float* beta, float eps,
                                float momentum, float* running_mean, float* running_var,
                                long long* num_batches_tracked, float* scale_shift,
                                float* mean_invstd, const float* conv_bias, int st, demf_stream_t stream) {
  if (int e = mlp_check(R, K, N, ldx)) return e;
  DEMF_REQUIRE(R >= 1 && X && Wt && Y, "mlp_gemm_fwd_bn: null pointer / no rows");
  if (int e = fin_check(R, gamma, beta, scale_shift, mean_invstd, stats)) return e;
  MlpArgs a{};
  a.R = R; a.K = K; a.N = N; a.ldx = ldx; a.ldy = N; a.X = X; a.vec = pro_scale_shift; a.Bt = Wt;
  a.Y = Y; a.stats = stats; a.st = st;
  hipStream_t s = (hipStream_t)stream;
  BnFin fin{(double)R, gamma, beta, conv_bias, eps, momentum, running_mean, running_var,
            num_batches_tracked, scale_shift, mean_invstd, nullptr};
  if (pro_scale_shift)
    return launch_with_finalize(a, fin, [s](const MlpArgs& b) { return launch_gemm<PRO_BNRELU, true>(b, s); }, s);
  return launch_with_finalize(a, fin, [s](const MlpArgs& b) { return launch_gemm<PRO_NONE, true>(b, s); }, s);
}

extern "C" int demf_mlp_gemm_fwd_bn(int R, int K, int N, int ldx, const float* X,
                                    const float* pro_scale_shift, const float* Wt, float* Y,
                                    double* stats, const float* gamma, const float* beta, float eps,
                                    float momentum, float* running_mean, float* running_var,
                                    long long* num_batches_tracked, float* scale_shift,
                                    float* mean_invstd, const float* conv_bias, demf_stream_t stream) {
  return mlp_gemm_fwd_bn_impl(R, K, N, ldx, X, pro_scale_shift, Wt, Y, stats, gamma, beta, eps, momentum,
                              running_mean, running_var, num_batches_tracked, scale_shift, mean_invstd,
                              conv_bias, 0, stream);
}

// Same with the rows stored as bf16 in HBM (store_flags bit 0: X is bf16, bit 1: Y is bf16; X / Y then
// point to 2-byte elements, ldx still counts elements).  bf16 compute mode, weight-resident forward only
// (K = 64, N = 64 / 128, R >= 16384): DEMF_EUNSUPPORTED otherwise.  BASELINE configs[3].
extern "C" int demf_mlp_gemm_fwd_bn_st(int R, int K, int N, int ldx, const void* X,
                                       const float* pro_scale_shift, const float* Wt, void* Y,
                                       double* stats, const float* gamma, const float* beta, float eps,
                                       float momentum, float* running_mean, float* running_var,
                                       long long* num_batches_tracked, float* scale_shift,
                                       float* mean_invstd, const float* conv_bias, int store_flags,
                                       demf_stream_t stream) {
  DEMF_REQUIRE(store_flags >= 1 && store_flags <= 3 && pro_scale_shift, "mlp_gemm_fwd_bn_st: store_flags in 1..3, BN prologue");
  return mlp_gemm_fwd_bn_impl(R, K, N, ldx, (const float*)X, pro_scale_shift, Wt, (float*)Y, stats, gamma, beta,
                              eps, momentum, running_mean, running_var, num_batches_tracked, scale_shift,
                              mean_invstd, conv_bias, store_flags, stream);
}

static int mlp_gemm_fwd_pool_bn_impl(int R, int K, int N, int ldx, const float* X,
                                     const float* pro_scale_shift, const float* Wt, float* Y,
                                     double* stats, int ns, float* pmax, float* pmin, int* amax,
                                     int* amin, const float* gamma, const float* beta, float eps,
                                     float momentum, float* running_mean, float* running_var,
                                     long long* num_batches_tracked, float* scale_shift,
                                     float* mean_invstd, const float* conv_bias, int st,
                                     demf_stream_t stream) {
  if (int e = mlp_check(R, K, N, ldx)) return e;
  const bool ok = (ns == 16 || ns == 32 || ns == 64) && R % ns == 0 && R >= 1;
  if (!ok) {
    set_error("mlp_gemm_fwd_pool_bn: ns=%d with R=%d N=%d is not fused", ns, R, N);
    return DEMF_EUNSUPPORTED;
  }
  DEMF_REQUIRE(X && Wt && (Y || st == 4) && pro_scale_shift && pmax && amax && (pmin == nullptr) == (amin == nullptr),
               "mlp_gemm_fwd_pool_bn: null pointer");
  if (int e = fin_check(R, gamma, beta, scale_shift, mean_invstd, stats)) return e;
  MlpArgs a{};
  a.R = R; a.K = K; a.N = N; a.ldx = ldx; a.ldy = N; a.X = X; a.vec = pro_scale_shift; a.Bt = Wt;
  a.Y = Y; a.stats = stats; a.ns = ns; a.pmax = pmax; a.pmin = pmin; a.amax = amax; a.amin = amin;
  a.st = st;
  hipStream_t s = (hipStream_t)stream;
  BnFin fin{(double)R, gamma, beta, conv_bias, eps, momentum, running_mean, running_var,
            num_batches_tracked, scale_shift, mean_invstd, nullptr};
  return launch_with_finalize(a, fin, [s](const MlpArgs& b) { return launch_gemm<PRO_BNRELU, true, true>(b, s); }, s);
}

extern "C" int demf_mlp_gemm_fwd_pool_bn(int R, int K, int N, int ldx, const float* X,
                                         const float* pro_scale_shift, const float* Wt, float* Y,
                                         double* stats, int ns, float* pmax, float* pmin, int* amax,
                                         int* amin, const float* gamma, const float* beta, float eps,
                                         float momentum, float* running_mean, float* running_var,
                                         long long* num_batches_tracked, float* scale_shift,
                                         float* mean_invstd, const float* conv_bias,
                                         demf_stream_t stream) {
  return mlp_gemm_fwd_pool_bn_impl(R, K, N, ldx, X, pro_scale_shift, Wt, Y, stats, ns, pmax, pmin, amax, amin,
                                   gamma, beta, eps, momentum, running_mean, running_var,
                                   num_batches_tracked, scale_shift, mean_invstd, conv_bias, 0, stream);
}

// Pooled form with bf16 row storage (see demf_mlp_gemm_fwd_bn_st); pmin / amin must be NULL.
extern "C" int demf_mlp_gemm_fwd_pool_bn_st(int R, int K, int N, int ldx, const void* X,
                                            const float* pro_scale_shift, const float* Wt, void* Y,
                                            double* stats, int ns, float* pmax, int* amax,
                                            const float* gamma, const float* beta, float eps,
                                            float momentum, float* running_mean, float* running_var,
                                            long long* num_batches_tracked, float* scale_shift,
                                            float* mean_invstd, const float* conv_bias, int store_flags,
                                            demf_stream_t stream) {
  DEMF_REQUIRE(store_flags >= 1 && store_flags <= 4, "mlp_gemm_fwd_pool_bn_st: store_flags in 1..4");
  return mlp_gemm_fwd_pool_bn_impl(R, K, N, ldx, (const float*)X, pro_scale_shift, Wt, (float*)Y, stats, ns,
                                   pmax, nullptr, amax, nullptr, gamma, beta, eps, momentum, running_mean,
                                   running_var, num_batches_tracked, scale_shift, mean_invstd, conv_bias,
                                   store_flags, stream);
}

// ---- first layer of an SA1-shaped stack (4-float rows -> N0 channels) WITHOUT its output -----------------
// y = x.W0^T is linear in a 4-float row: the train-mode BN statistics of all N0 channels follow from the 14
// second moments of x alone,  sum y_c = w_c . sum x,   sum y_c^2 = w_c^T (sum x x^T) w_c,  taken in one pass
// over the 16-byte rows (fp32 per thread, fp64 across threads); the last workgroup turns them into
// scale / shift, mean / invstd and the running statistics (bn_finalize_channel).  The 268 MB of raw
// outputs the GEMM kernel wrote for the consumers are recomputed by them from the rows (ST bit 3 of
// mlp_fwd_res_kernel, mlp_bwd_fused_kernel).  bf16 mode: moments of the bf16-rounded rows, rounded weights.
__global__ __launch_bounds__(256) void mlp_first_stats_k(int R, int N0, const float* __restrict__ X,
                                                         const float* __restrict__ W0, double* __restrict__ mom,
                                                         BnFin fin, int bf16) {
  float m[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) m[i] = 0.f;
  double acc[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) acc[i] = 0.0;
  int n = 0;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
    float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * 4);
    if (bf16) { x.x = (float)(__bf16)x.x; x.y = (float)(__bf16)x.y; x.z = (float)(__bf16)x.z; x.w = (float)(__bf16)x.w; }
    const float v[4] = {x.x, x.y, x.z, x.w};
    int t = 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m[i] += v[i];
#pragma unroll
      for (int j = i; j < 4; ++j) { m[t] = __builtin_fmaf(v[i], v[j], m[t]); ++t; }
    }
    if (++n == 64) {                       // fp32 partial sums of at most 64 rows, then fp64
#pragma unroll
      for (int i = 0; i < 14; ++i) { acc[i] += (double)m[i]; m[i] = 0.f; }
      n = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < 14; ++i) acc[i] += (double)m[i];
  __shared__ double s_m[4][14];
  __shared__ int s_last;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    double v = acc[i];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 14)
    atomicAdd(mom + threadIdx.x, (s_m[0][threadIdx.x] + s_m[1][threadIdx.x]) + (s_m[2][threadIdx.x] + s_m[3][threadIdx.x]));
  sync_drained();
  if (threadIdx.x == 0) s_last = last_workgroup(fin.ticket, (int)gridDim.x, (int)blockIdx.x);
  __syncthreads();
  if (!s_last) return;
  __shared__ double s_t[14];
  if (threadIdx.x < 14)
    s_t[threadIdx.x] = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(mom + threadIdx.x), 0ull));
  __syncthreads();
  if (threadIdx.x == 0 && fin.nbt != nullptr) *fin.nbt += 1;
  for (int c = threadIdx.x; c < N0; c += 256) {
    double w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float wk = W0[c * 4 + k];
      w[k] = bf16 ? (double)(float)(__bf16)wk : (double)wk;
    }
    double s1 = 0.0, s2 = 0.0;
    int t = 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s1 += w[i] * s_t[i];
#pragma unroll
      for (int j = i; j < 4; ++j) { s2 += (i == j ? 1.0 : 2.0) * w[i] * w[j] * s_t[t]; ++t; }
    }
    bn_finalize_channel(c, N0, fin.count, s1, s2, fin.gamma, fin.beta, fin.eps, fin.momentum, fin.rmean, fin.rvar,
                        fin.ss, fin.mi, fin.conv_bias);
  }
}

static int pool_select_impl(int Rp, int C, const float* pmax, const float* pmin, const int* amax, const int* amin,
                            const float* scale_shift, float* out, int* arg, float* yraw, int slot0,
                            demf_stream_t stream) {
  DEMF_REQUIRE(Rp >= 0 && C >= 1, "pool_select: bad sizes");
  if (Rp == 0) return DEMF_OK;
  DEMF_REQUIRE(pmax && pmin && amax && amin && scale_shift && out && arg, "pool_select: null pointer");
  const long long RC = (long long)Rp * C;
  long long g = (RC + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(pool_select_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, RC, C, pmax,
                     pmin, amax, amin, scale_shift, out, arg, yraw, slot0);
  return check_launch("pool_select");
}

// First layer of an SA1-shaped stack without its output: BN statistics + bookkeeping from the 4-float rows
// alone (mlp_first_stats_k).  ``moments``: >= 14 doubles of a zeroed accumulator, left zeroed.
extern "C" int demf_mlp_first_stats(int R, int N0, const float* X, const float* W0, double* moments,
                                    const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, long long* num_batches_tracked,
                                    float* scale_shift, float* mean_invstd, const float* conv_bias,
                                    demf_stream_t stream) {
  DEMF_REQUIRE(R >= 1 && N0 >= 1 && X && W0 && moments, "mlp_first_stats: bad arguments");
  if (int e = fin_check(R, gamma, beta, scale_shift, mean_invstd, moments)) return e;
  BnFin fin{(double)R, gamma, beta, conv_bias, eps, momentum, running_mean, running_var,
            num_batches_tracked, scale_shift, mean_invstd, sched_slot()};
  DEMF_REQUIRE(fin.ticket != nullptr, "mlp_first_stats: needs the counter ring (DEMF_STATIC_TILES=1 is set)");
  int grid = cdiv(R, 256 * 16);
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(mlp_first_stats_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, R, N0, X, W0, moments, fin,
                     compute_mode() == 1 ? 1 : 0);
  return check_launch("mlp_first_stats");
}

// Second layer of that stack (K = 64 -> N = 64): demf_mlp_gemm_fwd_bn with the prologue's input rebuilt from the
// 4-float rows X4 and the first layer's weight W0 (64 x 4) instead of loaded from a stored (R x 64) output;
// prev_scale_shift = the first layer's [scale|shift] from demf_mlp_first_stats.  Modes 1 / 2, R >= 16384.
extern "C" int demf_mlp_gemm_fwd_bn_x4(int R, int N, const float* X4, const float* W0,
                                       const float* prev_scale_shift, const float* Wt, float* Y, double* stats,
                                       const float* gamma, const float* beta, float eps, float momentum,
                                       float* running_mean, float* running_var, long long* num_batches_tracked,
                                       float* scale_shift, float* mean_invstd, const float* conv_bias,
                                       demf_stream_t stream) {
  const int cm = compute_mode();
  DEMF_REQUIRE((cm == 1 || cm == 2) && N == 64 && R >= 64 * 256 && X4 && W0 && prev_scale_shift && Wt && Y,
               "mlp_gemm_fwd_bn_x4: unsupported shape / mode R=%d N=%d mode=%d (or null pointer)", R, N, cm);
  if (int e = fin_check(R, gamma, beta, scale_shift, mean_invstd, stats)) return e;
  MlpArgs a{};
  a.R = R; a.K = 64; a.N = N; a.ldx = 4; a.ldy = N; a.X = X4; a.W0 = W0; a.vec = prev_scale_shift; a.Bt = Wt;
  a.Y = Y; a.stats = stats; a.st = 8;
  hipStream_t s = (hipStream_t)stream;
  BnFin fin{(double)R, gamma, beta, conv_bias, eps, momentum, running_mean, running_var,
            num_batches_tracked, scale_shift, mean_invstd, nullptr};
  return launch_with_finalize(a, fin, [s](const MlpArgs& b) { return launch_gemm<PRO_BNRELU, true>(b, s); }, s);
}

extern "C" int demf_pool_select(int Rp, int C, const float* pmax, const float* pmin,
                                const int* amax, const int* amin, const float* scale_shift,
                                float* out, int* arg, float* yraw, demf_stream_t stream) {
  return pool_select_impl(Rp, C, pmax, pmin, amax, amin, scale_shift, out, arg, yraw, 0, stream);
}

// The same behind the no-store pooled forward (demf_mlp_gemm_fwd_pool_bn_st, store_flags = 4): channels
// with a zero BN scale arrive with slot 0 and its raw value in pmax / amax, so yraw is complete (no NaN).
extern "C" int demf_pool_select_slot0(int Rp, int C, const float* pmax, const int* amax,
                                      const float* scale_shift, float* out, int* arg, float* yraw,
                                      demf_stream_t stream) {
  return pool_select_impl(Rp, C, pmax, pmax, amax, amax, scale_shift, out, arg, yraw, 1, stream);
}

extern "C" int demf_bn_finalize(int N, long long count, double* stats, const float* gamma,
                                const float* beta, float eps, float momentum,
                                float* running_mean, float* running_var,
                                long long* num_batches_tracked, float* scale_shift,
                                float* mean_invstd, const float* conv_bias, demf_stream_t stream) {
  DEMF_REQUIRE(N >= 1 && count >= 1 && stats && gamma && beta && scale_shift && mean_invstd,
               "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, N,
                     (double)count, stats, gamma, beta, eps, momentum, running_mean, running_var,
                     num_batches_tracked, scale_shift, mean_invstd, conv_bias);
  return check_launch("bn_finalize");
}

extern "C" int demf_bnrelu_maxpool_fwd(int R, int ns, int C, const float* Y,
                                       const float* scale_shift, float* out, int* arg,
                                       demf_stream_t stream) {
  DEMF_REQUIRE(R >= 0 && ns >= 1 && C >= 1, "bnrelu_maxpool: bad sizes");
  if (R == 0) return DEMF_OK;
  DEMF_REQUIRE(Y && scale_shift && out && arg, "bnrelu_maxpool: null pointer");
  const long long RC = (long long)R * C;
  long long g = (RC + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(bnrelu_maxpool_fwd_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, RC,
                     ns, C, Y, scale_shift, out, arg);
  return check_launch("bnrelu_maxpool_fwd");
}

extern "C" int demf_bn_bwd_vectors(int, long long, double*, const float*, const float*, const float*, float*,
                                   float*, float*, demf_stream_t);

static int bn_bwd_reduce_impl(int R, int N, int ns, const float* G, const float* dP,
                              const int* arg, const float* Y, const float* yraw,
                              const float* scale_shift, const float* mean_invstd, double* g12,
                              const BnVecFin& vf, demf_stream_t stream, int y_bf16 = 0) {
  DEMF_REQUIRE(R >= 0 && N >= 1, "bn_bwd_reduce: bad sizes R=%d N=%d", R, N);
  if (R == 0) return DEMF_OK;
  // (Y may be NULL in the pooled form when yraw is complete - the no-store forward, whose yraw never holds
  // the NaN that asks for a gather from Y)
  DEMF_REQUIRE((Y || (!G && yraw)) && scale_shift && mean_invstd && g12 && (G || (dP && arg && ns >= 1)),
               "bn_bwd_reduce: null pointer");
  // (a "pooled" layer with one sample per group - the row MLPs of the FP / vote / head modules - is the
  // dense case: dP is the upstream gradient of every row and the selected row is the row itself)
  if (!G && ns == 1 && !y_bf16 && N % 4 == 0 && N <= 1024 && env_int("DEMF_BNRED_NS1_DENSE", 1)) G = dP;
  if (G && N % 4 == 0 && N <= 1024) {
    const int rp = 256 / (N / 4);
    // few blocks: every block ends with 2N same-address fp64 atomics, which serialise in L2
    // (~70 ns each), so 2048 blocks cost ~150 us in the tail alone
    // (with the replicated accumulators a workgroup's 2N adds queue behind an eighth of the others': one unrolled
    //  two unrolled passes of 4 row groups per workgroup, up to 1024 of them; measured on the resident step: 4 row groups
    //  4.233, 8 4.208, 16 4.208-4.218 ms; without the copies (16 per workgroup, <= 256 of them) 4.248)
    const bool repl = vf.ticket != nullptr && vf.racc != nullptr;
    int grid = cdiv(R, rp * (repl ? env_int("DEMF_BNRED_RPB", 8) : 16));
    const int cap = env_int("DEMF_BNRED_GRID", repl ? 1024 : 256);
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(bn_bwd_reduce_dense4_k<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, R, N, G,
                       Y, scale_shift, mean_invstd, g12, env_int("DEMF_BNRED_REV", 1), vf,
                       (const float*)nullptr, (const int*)nullptr, 1, 0);
    return check_launch("bn_bwd_reduce");
  }
  // pooled form with the selected rows' raw outputs at hand: the same float4 kernel over the R/ns pooled
  // rows (<= 256 blocks -> <= 256 same-address fp64 atomics per channel instead of 1 024: at SA1 the
  // sparse kernel's 31 us were mostly that tail)
  if (!G && yraw && N % 4 == 0 && N <= 1024 && R % ns == 0 && env_int("DEMF_BNRED_POOLED_DENSE", 1)) {
    const int Rp = R / ns, rp = 256 / (N / 4);
    const bool repl = vf.ticket != nullptr && vf.racc != nullptr;
    int grid = cdiv(Rp, rp * (repl ? env_int("DEMF_BNRED_RPB", 8) : 16));
    const int cap = env_int("DEMF_BNRED_GRID", repl ? 1024 : 256);
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(bn_bwd_reduce_dense4_k<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, Rp, N, dP,
                       yraw, scale_shift, mean_invstd, g12, 0, vf, Y, arg, ns, y_bf16);
    return check_launch("bn_bwd_reduce");
  }
  const int rows_par = N < 256 ? 256 / N : 1;
  const int rows = G ? R : R / ns;
  // (measured on the training step, sparse form: caps of 256 / 128 / 64 / 32 blocks cost +0.04 / +0.21 /
  // +0.45 / +1.0 ms per step - the launch is bound by its dependent loads per block, not by the
  // 2N same-address fp64 atomics each block ends with)
  int grid = cdiv(rows, rows_par * (G ? 8 : env_int("DEMF_BNRED_SPARSE_RPT", 8)));
  const int scap = G ? 1024 : env_int("DEMF_BNRED_SPARSE_GRID", 1024);
  if (grid > scap) grid = scap;
  if (grid < 1) grid = 1;
  if (G)
    hipLaunchKernelGGL((bn_bwd_reduce_k<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, R, N,
                       ns, G, dP, arg, Y, scale_shift, mean_invstd, g12, nullptr, BnVecFin{}, 0);
  else
    hipLaunchKernelGGL((bn_bwd_reduce_k<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, R, N,
                       ns, G, dP, arg, Y, scale_shift, mean_invstd, g12, yraw, vf, y_bf16);
  return check_launch("bn_bwd_reduce");
}

extern "C" int demf_bn_bwd_reduce(int R, int N, int ns, const float* G, const float* dP,
                                  const int* arg, const float* Y, const float* yraw,
                                  const float* scale_shift, const float* mean_invstd, double* g12,
                                  demf_stream_t stream) {
  return bn_bwd_reduce_impl(R, N, ns, G, dP, arg, Y, yraw, scale_shift, mean_invstd, g12, BnVecFin{}, stream);
}

// Sparse (pooled last layer) reduce + the layer's backward vectors in ONE launch: what
// demf_bn_bwd_reduce(G = NULL) followed by demf_bn_bwd_vectors does in two.
extern "C" int demf_bn_bwd_reduce_vectors(int R, int N, int ns, const float* dP, const int* arg,
                                          const float* Y, const float* yraw, const float* scale_shift,
                                          const float* mean_invstd, double* g12, const float* gamma,
                                          float* vec6, float* dgamma, float* dbeta, int y_bf16,
                                          demf_stream_t stream) {
  DEMF_REQUIRE(gamma && vec6 && dgamma && dbeta && dP && arg, "bn_bwd_reduce_vectors: null pointer");
  BnVecFin vf{(double)R, gamma, scale_shift, mean_invstd, vec6, dgamma, dbeta, sched_slot(), nullptr};
  if (vf.ticket != nullptr && N <= ACC_MAX_N) vf.racc = accum_slot();
  if (vf.ticket == nullptr) {              // DEMF_STATIC_TILES=1: no counter sets - two launches
    if (int e = bn_bwd_reduce_impl(R, N, ns, nullptr, dP, arg, Y, yraw, scale_shift, mean_invstd, g12, BnVecFin{}, stream, y_bf16)) return e;
    return demf_bn_bwd_vectors(N, R, g12, gamma, scale_shift, mean_invstd, vec6, dgamma, dbeta, stream);
  }
  return bn_bwd_reduce_impl(R, N, ns, nullptr, dP, arg, Y, yraw, scale_shift, mean_invstd, g12, vf, stream, y_bf16);
}

extern "C" int demf_bn_bwd_vectors(int N, long long count, double* g12, const float* gamma,
                                   const float* scale_shift, const float* mean_invstd, float* vec6,
                                   float* dgamma, float* dbeta, demf_stream_t stream) {
  DEMF_REQUIRE(N >= 1 && count >= 1 && g12 && gamma && scale_shift && mean_invstd && vec6 &&
                   dgamma && dbeta,
               "bn_bwd_vectors: bad arguments");
  hipLaunchKernelGGL(bn_bwd_vectors_k, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, N,
                     (double)count, g12, gamma, scale_shift, mean_invstd, vec6, dgamma, dbeta);
  return check_launch("bn_bwd_vectors");
}

// dX(R x K) = dY(R x N) @ W(N x K), dY formed on the fly.  Wtt = W^T as (K x N) row-major.
// dX has row stride ldo >= K; K may be any multiple of 4 (handled in chunks of <= 128 columns).
struct DxReduce {            // optional: layer l-1 operands for the RED epilogue
  const float* Yprev;        // (R x K) pre-BN output of layer l-1
  const float* ss;           // [scale|shift] (2K)
  const float* mi;           // [mean|invstd] (2K)
  double* g12;               // (2K) accumulated
  const float* gamma = nullptr;   // + layer l-1's backward vectors by the last workgroup when given
  float* vec6 = nullptr;
  float* dgamma = nullptr;
  float* dbeta = nullptr;
};

static int mlp_bwd_dx_impl(bool w_direct, int R, int N, int K, int ldo, const float* G, const float* dP,
                           const int* arg, int ns, const float* Y, const float* vec6,
                           const float* Wtt, float* dX, const DxReduce* red, demf_stream_t stream) {
  DEMF_REQUIRE(R >= 0 && N >= 4 && N % 4 == 0 && K >= 1 && ldo >= K,
               "mlp_gemm_bwd_dx: bad sizes R=%d N=%d K=%d ldo=%d", R, N, K, ldo);
  if (R == 0) return DEMF_OK;
  DEMF_REQUIRE(Y && vec6 && Wtt && dX && (G || (dP && arg && ns >= 1)), "mlp_gemm_bwd_dx: null pointer");
  hipStream_t s = (hipStream_t)stream;
  // Outputs wider than 128 columns: ONE launch whose column tiles are split over blockIdx.y (every
  // block rebuilds its dY row tile, as the per-chunk launches did, but the chunks now run side by side
  // instead of as 2-4 dependent launches of one block per CU each); DEMF_DX_CHUNKS=1: chunk launches.
  static const int chunked = env_int("DEMF_DX_CHUNKS", 0);
  const int cstep = (!chunked && K <= 512) ? K : 128;
  for (int c0 = 0; c0 < K; c0 += cstep) {
    // here the reduction runs over this layer's N channels and the output has K columns
    MlpArgs a{};
    a.R = R; a.K = N; a.N = (K - c0) < cstep ? (K - c0) : cstep; a.ldx = N; a.ldy = ldo; a.X = Y;
    a.G = G; a.dP = dP; a.arg = arg; a.ns = ns; a.vec = vec6;
    if (w_direct) { a.Bt = Wtt + c0; a.ldb = K; }      // Wtt is W (N x K) itself: columns c0.. of it
    else a.Bt = Wtt + (size_t)c0 * N;
    a.Y = dX + c0; a.stats = nullptr;
    int e;
    if (red) {
      a.fY = red->Yprev; a.fss = red->ss; a.fmi = red->mi; a.stats = red->g12; a.fld = K; a.fc0 = c0;
      if (red->gamma != nullptr)
        a.vfin = BnVecFin{(double)R, red->gamma, red->ss, red->mi, red->vec6, red->dgamma, red->dbeta, sched_slot(),
                          nullptr};
      if (a.vfin.ticket != nullptr && K <= ACC_MAX_N && env_int("DEMF_ACC_REPL_RED", 0)) a.vfin.racc = accum_slot();
      e = G ? launch_gemm<PRO_DY_DENSE, false, false, true>(a, s)
            : launch_gemm<PRO_DY_SPARSE, false, false, true>(a, s);
    } else {
      e = G ? launch_gemm<PRO_DY_DENSE, false>(a, s) : launch_gemm<PRO_DY_SPARSE, false>(a, s);
    }
    if (e) return e;
  }
  if (red && red->gamma != nullptr && env_int("DEMF_STATIC_TILES", 0))
    // no counter ring, no last-workgroup ticket: the vectors of layer l-1 as a launch of their own (all K channels)
    return demf_bn_bwd_vectors(K, (long long)R, red->g12, red->gamma, red->ss, red->mi, red->vec6, red->dgamma,
                               red->dbeta, stream);
  return DEMF_OK;
}

extern "C" int demf_mlp_gemm_bwd_dx(int R, int N, int K, int ldo, const float* G, const float* dP,
                                    const int* arg, int ns, const float* Y, const float* vec6,
                                    const float* Wtt, float* dX, demf_stream_t stream) {
  return mlp_bwd_dx_impl(false, R, N, K, ldo, G, dP, arg, ns, Y, vec6, Wtt, dX, nullptr, stream);
}

// Same, reading the layer's weight W (N x K row-major) directly: no transposed copy per step.
extern "C" int demf_mlp_gemm_bwd_dx_w(int R, int N, int K, int ldo, const float* G, const float* dP,
                                      const int* arg, int ns, const float* Y, const float* vec6,
                                      const float* W, float* dX, demf_stream_t stream) {
  return mlp_bwd_dx_impl(true, R, N, K, ldo, G, dP, arg, ns, Y, vec6, W, dX, nullptr, stream);
}

// Same + the BN-backward sums of layer l-1 (what demf_bn_bwd_reduce(R, K, G = dX, Yprev, ...) would
// add to g12_prev) taken from the output tiles on the way out: that pass over dX and Yprev is gone.
extern "C" int demf_mlp_gemm_bwd_dx_red(int R, int N, int K, int ldo, const float* G, const float* dP,
                                        const int* arg, int ns, const float* Y, const float* vec6,
                                        const float* W, float* dX, const float* Yprev,
                                        const float* scale_shift_prev, const float* mean_invstd_prev,
                                        double* g12_prev, demf_stream_t stream) {
  DEMF_REQUIRE(K % 4 == 0 && Yprev && scale_shift_prev && mean_invstd_prev && g12_prev,
               "mlp_gemm_bwd_dx_red: bad arguments (K %% 4 == 0)");
  const DxReduce red{Yprev, scale_shift_prev, mean_invstd_prev, g12_prev};
  return mlp_bwd_dx_impl(true, R, N, K, ldo, G, dP, arg, ns, Y, vec6, W, dX, &red, stream);
}

// Same + layer l-1's backward vectors (demf_bn_bwd_vectors on g12_prev) formed by the launch's last
// workgroup: vec6_prev (5K), dgamma_prev, dbeta_prev are complete and g12_prev is zeroed on return.
extern "C" int demf_mlp_gemm_bwd_dx_red_v(int R, int N, int K, int ldo, const float* G, const float* dP,
                                          const int* arg, int ns, const float* Y, const float* vec6,
                                          const float* W, float* dX, const float* Yprev,
                                          const float* scale_shift_prev, const float* mean_invstd_prev,
                                          double* g12_prev, const float* gamma_prev, float* vec6_prev,
                                          float* dgamma_prev, float* dbeta_prev, demf_stream_t stream) {
  DEMF_REQUIRE(K % 4 == 0 && Yprev && scale_shift_prev && mean_invstd_prev && g12_prev && gamma_prev &&
                   vec6_prev && dgamma_prev && dbeta_prev,
               "mlp_gemm_bwd_dx_red_v: bad arguments (K %% 4 == 0)");
  DxReduce red{Yprev, scale_shift_prev, mean_invstd_prev, g12_prev};
  red.gamma = gamma_prev; red.vec6 = vec6_prev; red.dgamma = dgamma_prev; red.dbeta = dbeta_prev;
  return mlp_bwd_dx_impl(true, R, N, K, ldo, G, dP, arg, ns, Y, vec6, W, dX, &red, stream);
}

// ---- first layer of a stack with a 4-float input and no input gradient (SA1) --------------------
// dY0 = gi*dZ0 + a*y0 + b is linear in (dZ0, y0, 1), so layer 0's whole backward follows from raw
// sums that do not depend on the BN-backward scalars: g1 = sum dZ0, g2 = sum dZ0*xhat,
// P = dZ0^T X, Q = y0^T X, cx = colsum(X).  The dx GEMM of layer 1 takes them from its output tile
// (FIRST epilogue) instead of storing the (R x N0) gradient, which is then never written nor read:
//   dW0 = gi*P + a*Q + b*cx^T,  dgamma0 = g2,  dbeta0 = g1.
__global__ void mlp_first_finish_k(int N, double count, double* __restrict__ sums,
                                   const float* __restrict__ gamma, const float* __restrict__ mi,
                                   float* __restrict__ dW, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta) {
  __shared__ double s_cx[4];
  if (threadIdx.x < 4) s_cx[threadIdx.x] = sums[10 * N + threadIdx.x];
  __syncthreads();
  const int c = threadIdx.x;
  if (c < N) {
    const double g1 = sums[c], g2 = sums[N + c];
    const double mean = mi[c], is = mi[N + c];
    const double gi = (double)gamma[c] * is;
    const double a = -gi * is * (g2 / count);
    const double b = -gi * (g1 / count) - a * mean;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dW[c * 4 + k] = (float)(gi * sums[2 * N + c * 4 + k] + a * sums[6 * N + c * 4 + k] + b * s_cx[k]);
      sums[2 * N + c * 4 + k] = 0.0;               // consumed: the accumulator is left zeroed
      sums[6 * N + c * 4 + k] = 0.0;
    }
    sums[c] = 0.0;
    sums[N + c] = 0.0;
    dgamma[c] = (float)g2;
    dbeta[c] = (float)g1;
  }
  if (threadIdx.x < 4) sums[10 * N + threadIdx.x] = 0.0;
}

extern "C" int demf_mlp_gemm_bwd_dx_first(int R, int N, int K0, const float* G, const float* Y1,
                                          const float* vec6, const float* W1, const float* X0,
                                          const float* Y0, const float* ss0, const float* mi0,
                                          double* sums, demf_stream_t stream) {
  DEMF_REQUIRE(R >= 1 && N >= 4 && N % 4 == 0 && N <= MLP_MAXK && K0 >= 4 && K0 % 4 == 0 && K0 <= 64,
               "mlp_gemm_bwd_dx_first: bad sizes R=%d N=%d K0=%d (K0 <= 64)", R, N, K0);
  DEMF_REQUIRE(G && Y1 && vec6 && W1 && X0 && Y0 && ss0 && mi0 && sums,
               "mlp_gemm_bwd_dx_first: null pointer");
  hipStream_t s = (hipStream_t)stream;
  MlpArgs a{};
  a.R = R; a.K = N; a.N = K0; a.ldx = N; a.ldy = K0; a.X = Y1; a.G = G; a.vec = vec6;
  a.Bt = W1; a.ldb = K0; a.Y = nullptr; a.stats = nullptr;
  a.fX = X0; a.fY = Y0; a.fss = ss0; a.fmi = mi0; a.fsum = sums; a.fld = K0; a.fc0 = 0;
  // 32-row wave tiles: with 64-row ones the staged half tile + the 24 running sums do not fit next
  // to the accumulators in 256 VGPRs
  const int gx = mlp_grid(R, 128);
  const int tiles = (R + 127) / 128;
  if (tiles > gx && a.K > MLP_BK && gx % (8 * SCHED_GROUPS) == 0) a.sched = sched_slot();
#define FIRSTGO(NTv, CMv) \
  hipLaunchKernelGGL((mlp_gemm_kernel<NTv, 1, PRO_DY_DENSE, false, false, true, false, CMv>), dim3(gx), dim3(256), 0, s, a)
  const int cm = compute_mode();
  if (K0 <= 32) {
    if (cm == 1) FIRSTGO(1, 1); else if (cm == 2) FIRSTGO(1, 2); else FIRSTGO(1, 0);
  } else {
    if (cm == 1) FIRSTGO(2, 1); else if (cm == 2) FIRSTGO(2, 2); else FIRSTGO(2, 0);
  }
#undef FIRSTGO
  return check_launch("mlp_gemm_bwd_dx_first");
}

extern "C" int demf_mlp_first_finish(int N0, long long count, double* sums, const float* gamma0,
                                     const float* mean_invstd0, float* dW0, float* dgamma0,
                                     float* dbeta0, demf_stream_t stream) {
  DEMF_REQUIRE(N0 >= 1 && N0 <= 64 && count >= 1 && sums && gamma0 && mean_invstd0 && dW0 && dgamma0 &&
                   dbeta0, "mlp_first_finish: bad arguments");
  hipLaunchKernelGGL(mlp_first_finish_k, dim3(1), dim3(64), 0, (hipStream_t)stream, N0, (double)count,
                     sums, gamma0, mean_invstd0, dW0, dgamma0, dbeta0);
  return check_launch("mlp_first_finish");
}

extern "C" int demf_mlp_gemm_bwd_dw_ld(int, int, int, int, const float*, const float*, const int*, int,
                                       const float*, const float*, const float*, const float*, float*,
                                       int, demf_stream_t);

extern "C" int demf_mlp_gemm_bwd_dw(int R, int N, int K, int ldx, const float* G, const float* dP,
                                    const int* arg, int ns, const float* Y, const float* vec6,
                                    const float* Xprev, const float* prev_scale_shift, float* dW,
                                    demf_stream_t stream) {
  return demf_mlp_gemm_bwd_dw_ld(R, N, K, ldx, G, dP, arg, ns, Y, vec6, Xprev, prev_scale_shift, dW, K,
                                 stream);
}

namespace {
struct DwPlan {
  DwArgs a;
  int gx, nsub, tn, tk, x3;      // x3: 0 fp32 MFMA, 1 three-term split, 2 bf16
  bool sparse;
  size_t lds;
};

// argument checks + launch geometry of one weight-gradient product (shared by the single and the grouped entry)
int dw_plan(int R, int N, int K, int ldx, const float* G, const float* dP, const int* arg, int ns, const float* Y,
            const float* vec6, const float* Xprev, const float* prev_scale_shift, float* dW, int lddw, bool allow_dyn,
            DwPlan& pl) {
  DEMF_REQUIRE(R >= 0 && N >= 4 && N % 4 == 0 && K >= 4 && K % 4 == 0 && ldx >= K && ldx % 4 == 0 &&
                   lddw >= K,
               "mlp_gemm_bwd_dw: bad sizes R=%d N=%d K=%d lddw=%d", R, N, K, lddw);
  pl.gx = 0;
  if (R == 0) return DEMF_OK;
  DEMF_REQUIRE(Y && vec6 && Xprev && dW && (G || (dP && arg && ns >= 1)), "mlp_gemm_bwd_dw: null pointer");
  const int TNt = cdiv(N, 32), TKt = cdiv(K, 32);
  // one launch: grid.x strides the 32-row slabs, grid.y enumerates the (<= 4x4-tile) sub-blocks
  // of the N x K output, so even a 4 k-row layer spreads over the whole chip
  int tn = TNt > 2 ? 2 : 1, tk = TKt > 2 ? 2 : 1;
  // few rows: 64 x 64 sub-blocks (4x as many blockIdx.y columns) and correspondingly fewer row chunks -
  // every block ends by adding its whole sub-block to dW with atomics, and with 128 x 128 sub-blocks x
  // R/128 chunks that flush (R/128 x N x K atomics) is most of a small launch
  // (round 5: also the 32 768-row products of the vote aggregation - they then join the grouped launch of the other
  // few-row products instead of running as two launches of their own: 292 -> 277 us for the step's 17 products,
  // tools/dw_micro.py; 5.074 -> 5.059 ms per step in situ)
  static const int small_r = env_int("DEMF_DW_SMALL_R", 65536);
  if (R <= small_r) tn = tk = 1;
  static const int tile_ab = env_int("DEMF_DW_TILE", 0);       // A/B: 12 -> 64 x 128 sub-blocks, 21 -> 128 x 64, 22, 11
  if (tile_ab) { tn = TNt > 2 ? tile_ab / 10 : 1; tk = TKt > 2 ? tile_ab % 10 : 1; }
  DwArgs a{};
  a.R = R; a.N = N; a.K = K; a.ldx = ldx; a.Yl = Y; a.G = G; a.dP = dP; a.arg = arg; a.ns = ns;
  a.vec = vec6; a.Xp = Xprev; a.pvec = prev_scale_shift; a.dW = dW; a.lddw = lddw;
  { static const int dbg = env_int("DEMF_DW_DBG", 0); a.dbg = dbg; }
  const int nsub_n = cdiv(TNt, 2 * tn);
  a.nsub_k = cdiv(TKt, 2 * tk);
  const int nsub = nsub_n * a.nsub_k;
  static const int x3mask_lds = env_int("DEMF_X3_MASK", 7);
  const int x3_lds = (compute_bf16() && !env_int("DEMF_DW_F32", 0)) ? 2 : ((compute_mode() == 2 && (x3mask_lds & 4)) ? 1 : 0);
  if (DEMF_DW_TR && x3_lds != 0)      // bf16 planes [PL][32][W + 8] of both operands (mlp_dw_body)
    pl.lds = (size_t)(x3_lds == 1 ? 3 : 1) * 32 * ((2 * tn * 32 + 8) + (2 * tk * 32 + 8)) * 2 + sizeof(float) * (5 * N + 2 * K);
  else
    pl.lds = sizeof(float) * (32 * ((2 * tn * 32 + 4) + (2 * tk * 32 + 4)) + 5 * N + 2 * K);
  int gx = cdiv(R, 32 * 4);
  const int tot = env_int("DEMF_DW_GRID", 512);
  const int cap = tot / nsub > 16 ? tot / nsub : 16;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  // dynamic chunk claims (DEMF_DW_DYN=1): measured neutral on the training step, so the default
  // stays the static slab striding (chunk = 1)
  a.chunk = 1;
  if (allow_dyn && env_int("DEMF_DW_DYN", 0)) {
    const int nslab = cdiv(R, 32);
    a.chunk = nslab / (gx * 4);                  // >= 4 claims per block, at most DW_CHUNK slabs each
    a.chunk = a.chunk < 1 ? 1 : (a.chunk > DW_CHUNK ? DW_CHUNK : a.chunk);
    const int nchunk = cdiv(nslab, a.chunk);
    if (gx > nchunk) gx = nchunk;
    if (nchunk > 2 * gx && nsub <= DW_MAX_SUB) a.sched = sched_slot();
  }
  static const int x3mask = env_int("DEMF_X3_MASK", 7);          // bit 2: weight-gradient launches
  pl.a = a; pl.gx = gx; pl.nsub = nsub; pl.tn = tn; pl.tk = tk; pl.sparse = G == nullptr;
  pl.x3 = (compute_bf16() && !env_int("DEMF_DW_F32", 0)) ? 2 : ((compute_mode() == 2 && (x3mask & 4)) ? 1 : 0);
  return DEMF_OK;
}

template <int X3>
void dw_launch_one(const DwPlan& pl, hipStream_t s) {
  const dim3 grid(pl.gx, pl.nsub);
#define DWV(TNv, TKv)                                                                                       \
  if (pl.sparse) hipLaunchKernelGGL((mlp_dw_kernel<TNv, TKv, true, X3>), grid, dim3(256), pl.lds, s, pl.a); \
  else hipLaunchKernelGGL((mlp_dw_kernel<TNv, TKv, false, X3>), grid, dim3(256), pl.lds, s, pl.a)
  if (pl.tn == 1 && pl.tk == 1) { DWV(1, 1); }
  else if (pl.tn == 1) { DWV(1, 2); }
  else if (pl.tk == 1) { DWV(2, 1); }
  else { DWV(2, 2); }
#undef DWV
}

template <int X3>
void dw_launch_group(const DwGroup& g, int tn, int tk, bool sparse, size_t lds, hipStream_t s) {
  const dim3 grid(g.start[g.n]);
#define DWG(TNv, TKv)                                                                                        \
  if (sparse) hipLaunchKernelGGL((mlp_dw_group_kernel<TNv, TKv, true, X3>), grid, dim3(256), lds, s, g);     \
  else hipLaunchKernelGGL((mlp_dw_group_kernel<TNv, TKv, false, X3>), grid, dim3(256), lds, s, g)
  if (tn == 1 && tk == 1) { DWG(1, 1); }
  else if (tn == 1) { DWG(1, 2); }
  else if (tk == 1) { DWG(2, 1); }
  else { DWG(2, 2); }
#undef DWG
}
}  // namespace

extern "C" int demf_mlp_gemm_bwd_dw_ld(int R, int N, int K, int ldx, const float* G, const float* dP,
                                       const int* arg, int ns, const float* Y, const float* vec6,
                                       const float* Xprev, const float* prev_scale_shift, float* dW,
                                       int lddw, demf_stream_t stream) {
  DwPlan pl;
  if (int e = dw_plan(R, N, K, ldx, G, dP, arg, ns, Y, vec6, Xprev, prev_scale_shift, dW, lddw, true, pl)) return e;
  if (pl.gx == 0) return DEMF_OK;
  hipStream_t s = (hipStream_t)stream;
  if (pl.x3 == 2) dw_launch_one<2>(pl, s);
  else if (pl.x3 == 1) dw_launch_one<1>(pl, s);
  else dw_launch_one<0>(pl, s);
  return check_launch("mlp_gemm_bwd_dw");
}

extern "C" int demf_mlp_gemm_bwd_dw_group(int n, const demf_dw_job* jobs, demf_stream_t stream) {
  DEMF_REQUIRE(n >= 0 && (n == 0 || jobs), "mlp_gemm_bwd_dw_group: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  std::vector<DwPlan> plans;
  plans.reserve(n);
  for (int i = 0; i < n; ++i) {
    const demf_dw_job& j = jobs[i];
    DwPlan pl;
    if (int e = dw_plan(j.R, j.N, j.K, j.ldx, j.G, j.dP, j.arg, j.ns, j.Y, j.vec6, j.Xprev, j.prev_scale_shift, j.dW,
                        j.lddw, false, pl))
      return e;
    if (pl.gx) plans.push_back(pl);
    static const int trace = env_int("DEMF_DW_TRACE", 0);       // the step's products, one line each (tools/dw_micro.py)
    if (trace && pl.gx)
      fprintf(stderr, "[dw] R=%d N=%d K=%d ldx=%d sparse=%d ns=%d bnrelu_in=%d tile=%dx%d grid=%dx%d\n", j.R, j.N, j.K, j.ldx,
              (int)pl.sparse, j.ns, j.prev_scale_shift != nullptr, pl.tn, pl.tk, pl.gx, pl.nsub);
  }
  std::vector<char> done(plans.size(), 0);
  for (size_t i = 0; i < plans.size(); ++i) {
    if (done[i]) continue;
    // every not yet issued job of the same kernel variant, DW_GROUP_MAX at a time
    DwGroup g{};
    size_t lds = 0;
    const DwPlan& f = plans[i];
    for (size_t k = i; k < plans.size() && g.n < DW_GROUP_MAX; ++k) {
      const DwPlan& q = plans[k];
      if (done[k] || q.tn != f.tn || q.tk != f.tk || q.sparse != f.sparse || q.x3 != f.x3) continue;
      g.job[g.n] = q.a; g.gx[g.n] = q.gx;
      g.start[g.n + 1] = g.start[g.n] + q.gx * q.nsub;
      lds = q.lds > lds ? q.lds : lds;
      ++g.n;
      done[k] = 1;
    }
    if (g.n == 1) {
      if (f.x3 == 2) dw_launch_one<2>(f, s);
      else if (f.x3 == 1) dw_launch_one<1>(f, s);
      else dw_launch_one<0>(f, s);
    } else if (f.x3 == 2) dw_launch_group<2>(g, f.tn, f.tk, f.sparse, lds, s);
    else if (f.x3 == 1) dw_launch_group<1>(g, f.tn, f.tk, f.sparse, lds, s);
    else dw_launch_group<0>(g, f.tn, f.tk, f.sparse, lds, s);
    if (int e = check_launch("mlp_gemm_bwd_dw_group")) return e;
  }
  return DEMF_OK;
}

#ifdef DEMF_MLP_PROFILE
// phases of mlp_gemm_kernel's K loop: 0 wait at the first barrier, 1 transform + LDS writes, 2 second barrier,
// 3 prefetch issue, 4 MFMA section (LDS reads + split + MFMAs), 5 epilogue + bookkeeping; [15] = launches
extern "C" int demf_mlp_prof_read(long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(demf::g_mlp_prof), 16 * sizeof(long long)) != hipSuccess) return -1;
  if (reset) {
    long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(demf::g_mlp_prof), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
