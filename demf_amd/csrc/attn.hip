// Self-attention core of the fusion decoder layer: softmax(q k^T / sqrt(d)) -> dropout -> . v for one
// (scene, head) at a time, forward and backward, without the (B*H, Q, Q) score / probability tensors ever
// reaching memory.
//
// Reference: mmcv DetrTransformerDecoderLayer -> nn.MultiheadAttention inside DeMFTransformerDecoderLayer
// (demf/modeling/layers/transformer.py:55-80; configs/demf/demf_votenet.py:71-91: 256 queries, 8 heads of
// 32 channels, attn_drop 0.1).  Round 3 ran it as QK^T, softmax + dropout, PV (3 launches) and five launches
// back, with two 16.8 MB tensors written and re-read each way.
//
// One workgroup = 64 "rows" of one (scene, head) against all 256 "columns", 4 waves of 16 rows each, on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: fp32 in, fp32 accumulate, bitwise an fmaf chain):
//   phase A  X . Y^T  (16 x 256 <- 32 per wave): 16 column tiles x 8 MFMAs; operands row-major in LDS, a
//            lane's reduction indices are k = 16 (s >> 2) + 4 g + (s & 3) (g = lane / 16), so both operands
//            arrive as float4s;
//   the accumulator layout puts row 4g + r of the wave in register r of the 16 lanes of group g, one column
//            per tile: row statistics are 16 in-lane values + one DPP row reduction, no cross-wave traffic;
//   phase B  W . Z    (16 x 32 <- 256 per wave): the wave's own 16 rows of W (row-major in LDS) against the
//            row-major Z copy phase A already uses - 2 channel tiles x 64 MFMAs.
// forward        X = q rows, Y = k: S -> P (row max and 1/sum kept: 2 floats per query) -> dropout -> W; O = W . v
// backward, rows X = q, Y = k: P again, bit for bit; X = dO, Y = v: dPd; dS = P (dP - D), D = dO . O;
//                dq = scale dS . k
// backward, keys the transposed problem (X = k rows of the block, Y = q; X = v, Y = dO): the same score bits
//                (a*b = b*a, same reduction order), column statistics from the forward; dv = Pd^T . dO,
//                dk = scale dS^T . q - complete per key block, no atomics.
// Compute mode 1 (bf16): operands are rounded to bf16 where the GEMM launches rounded them (q, k, v, dO when
// staged; dropout(P) and dS when they become an operand), products and sums in fp32 - the contract of
// oracle/emulate.py.  Modes 0 / 2: fp32 throughout.
// (A first version on packed fp32 FMAs - 4 x 16 register tiles, one wave per SIMD - ran 22 / 85 us:
// latency-bound; tools/attn_micro.py.)
#include "common.h"
#include "rng.h"

namespace demf {

constexpr int AT_Q = 256, AT_D = 32, AT_RB = 64;
constexpr int AT_LDR = AT_D + 4;         // row stride of the staged (rows x 32) operands
constexpr int AT_LDW = AT_Q + 4;
using f32x4 = float __attribute__((ext_vector_type(4)));

template <bool BF>
__device__ __forceinline__ float at_rnd(float x) {
  if constexpr (BF) return (float)(__bf16)x;
  return x;
}

// rows [0, NR) x 32 channels of one head (global row stride ld) -> T[row][AT_LDR], optionally rounded
template <int NR, bool BF>
__device__ __forceinline__ void at_stage(const float* __restrict__ src, size_t ld, float* __restrict__ T, int tid) {
#pragma unroll
  for (int i = 0; i < NR * 8 / 256; ++i) {
    const int f = tid + 256 * i;
    const int row = f >> 3, d4 = f & 7;
    float4 v = *reinterpret_cast<const float4*>(src + (size_t)row * ld + 4 * d4);
    v.x = at_rnd<BF>(v.x); v.y = at_rnd<BF>(v.y); v.z = at_rnd<BF>(v.z); v.w = at_rnd<BF>(v.w);
    *reinterpret_cast<float4*>(T + row * AT_LDR + 4 * d4) = v;
  }
}

// acc[t][r] = sum_k X[16w + 4g + r][k] * Y[16t + lane%16][k]   (t < 16 column tiles)
__device__ __forceinline__ void at_phase_a(const float* __restrict__ X, const float* __restrict__ Y, int wave, int lane,
                                           f32x4 acc[16]) {
  const int li = lane & 15, g = lane >> 4;
  const float* xr = X + (16 * wave + li) * AT_LDR + 4 * g;
  const float4 x0 = *reinterpret_cast<const float4*>(xr), x1 = *reinterpret_cast<const float4*>(xr + 16);
  const float xa[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float* yr = Y + (16 * t + li) * AT_LDR + 4 * g;
    const float4 y0 = *reinterpret_cast<const float4*>(yr), y1 = *reinterpret_cast<const float4*>(yr + 16);
    const float ya[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], ya[s], c, 0, 0, 0);
    acc[t] = c;
  }
}

// out[dt][r] = sum_j W[16w + 4g + r][j] * Z[j][16 dt + lane%16]   (dt < 2 channel tiles)
__device__ __forceinline__ void at_phase_b(const float* __restrict__ W, const float* __restrict__ Z, int wave, int lane,
                                           f32x4 out[2]) {
  const int li = lane & 15, g = lane >> 4;
  const float* wr = W + (16 * wave + li) * AT_LDW + 4 * g;
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int m = 0; m < AT_Q / 16; ++m) {
    const float4 w4 = *reinterpret_cast<const float4*>(wr + 16 * m);
    const float wa[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* zr = Z + (16 * m + 4 * g + e) * AT_LDR + li;
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[e], zr[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[e], zr[16], c1, 0, 0, 0);
    }
  }
  out[0] = c0; out[1] = c1;
}

// rows 16w + 4g + r, channels 16 dt + lane%16 of the head
__device__ __forceinline__ void at_store_b(const f32x4 out[2], float scale, float* __restrict__ dst, size_t ld,
                                           int wave, int lane) {
  const int li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float* p = dst + (size_t)(16 * wave + 4 * g + r) * ld + li;
    p[0] = out[0][r] * scale;
    p[16] = out[1][r] * scale;
  }
}

// the (seed, step, op) prefix of dropout_keep, once per thread
__device__ __forceinline__ uint32_t at_hash_prefix(const unsigned long long* __restrict__ rng, uint32_t op) {
  const unsigned long long seed = rng[0], step = rng[1];
  uint32_t h = mix32((uint32_t)seed ^ 0x9E3779B9u);
  h = mix32(h ^ (uint32_t)(seed >> 32));
  h = mix32(h ^ (uint32_t)step);
  return mix32(h ^ (uint32_t)(step >> 32) ^ (op * 0x632BE5ABu));
}
__device__ __forceinline__ bool at_keep(uint32_t h0, unsigned long long idx, float p) {
  uint32_t h = mix32(h0 ^ (uint32_t)idx);
  h = mix32(h ^ (uint32_t)(idx >> 32));
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;
}

struct AttnArgs {
  int B, H;
  const float* qkv;     // (B*Q, 3*H*32): q | k | v
  const float* out;     // (B*Q, H*32)  forward output (backward: D = dO . O)
  const float* dout;    // (B*Q, H*32)
  float* o;             // forward: out
  float* dqkv;          // backward: (B*Q, 3*H*32)
  float* stats;         // (B*H*Q, 2): row max of the scaled scores, 1 / sum of exp
  float* prob;          // optional (B*H, Q, Q) debug outputs of the forward (NULL in the product path)
  float* pdrop;
  float scale, p;
  const unsigned long long* rng;
  unsigned op;
};

constexpr size_t AT_LDS_FLOATS = 2 * (size_t)AT_RB * AT_LDR + 2 * (size_t)AT_Q * AT_LDR + (size_t)AT_RB * AT_LDW + 3 * AT_Q;

template <bool BF>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X = smem;                           // q rows of the block
  float* Y = X + 2 * AT_RB * AT_LDR;         // k
  float* V = Y + AT_Q * AT_LDR;              // v
  float* W = V + AT_Q * AT_LDR;              // dropout(P), [row][key]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x >> 2, rb = blockIdx.x & 3;
  const int b = bh / a.H, h = bh % a.H;
  const int E = a.H * AT_D;
  const size_t ld3 = 3 * (size_t)E;
  const float* base = a.qkv + (size_t)b * AT_Q * ld3 + h * AT_D;
  at_stage<AT_RB, BF>(base + (size_t)rb * AT_RB * ld3, ld3, X, tid);
  at_stage<AT_Q, BF>(base + E, ld3, Y, tid);
  at_stage<AT_Q, BF>(base + 2 * E, ld3, V, tid);
  __syncthreads();
  f32x4 acc[16];
  at_phase_a(X, Y, wave, lane, acc);
  const int li = lane & 15, g = lane >> 4;
  const float inv_keep = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.0f;
  const uint32_t h0 = a.p > 0.f ? at_hash_prefix(a.rng, a.op) : 0u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int lrow = 16 * wave + 4 * g + r, row = rb * AT_RB + lrow;
    float s[16];
    float mx = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 16; ++t) { s[t] = acc[t][r] * a.scale; mx = fmaxf(mx, s[t]); }
    mx = row16_allmax(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) { s[t] = __expf(s[t] - mx); sum += s[t]; }
    const float inv = 1.0f / group_allsum<16>(sum);
    const size_t o0 = ((size_t)bh * AT_Q + row) * AT_Q + li;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float pr = s[t] * inv;
      float q = pr;
      if (a.p > 0.f) q = at_keep(h0, o0 + 16 * t, a.p) ? pr * inv_keep : 0.f;
      if (a.prob) { a.prob[o0 + 16 * t] = pr; a.pdrop[o0 + 16 * t] = q; }
      W[lrow * AT_LDW + 16 * t + li] = at_rnd<BF>(q);
    }
    if (li == 0) {
      a.stats[((size_t)bh * AT_Q + row) * 2] = mx;
      a.stats[((size_t)bh * AT_Q + row) * 2 + 1] = inv;
    }
  }
  __syncthreads();
  f32x4 out[2];
  at_phase_b(W, V, wave, lane, out);
  at_store_b(out, 1.0f, a.o + ((size_t)b * AT_Q + rb * AT_RB) * E + h * AT_D, E, wave, lane);
}

// blockIdx: (scene, head) x {4 query blocks: dq, 4 key blocks: dk and dv}
template <bool BF>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X = smem;                           // rows role: q block | keys role: k block
  float* X2 = X + AT_RB * AT_LDR;            //            dO block |            v block
  float* Y = X2 + AT_RB * AT_LDR;            //            k       |            q
  float* Y2 = Y + AT_Q * AT_LDR;             //            v       |            dO
  float* W = Y2 + AT_Q * AT_LDR;             // [row][column] operand of phase B
  float* s_mx = W + AT_RB * AT_LDW;          // per query: row max, 1 / sum, D = dO . O
  float* s_inv = s_mx + AT_Q;
  float* s_D = s_inv + AT_Q;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x >> 3, role = (blockIdx.x >> 2) & 1, blk = blockIdx.x & 3;
  const int b = bh / a.H, h = bh % a.H;
  const int E = a.H * AT_D;
  const size_t ld3 = 3 * (size_t)E;
  const float* qb = a.qkv + (size_t)b * AT_Q * ld3 + h * AT_D;
  const float* dob = a.dout + (size_t)b * AT_Q * E + h * AT_D;
  const float* ob = a.out + (size_t)b * AT_Q * E + h * AT_D;
  if (role == 0) {
    at_stage<AT_RB, BF>(qb + (size_t)blk * AT_RB * ld3, ld3, X, tid);
    at_stage<AT_RB, BF>(dob + (size_t)blk * AT_RB * E, E, X2, tid);
    at_stage<AT_Q, BF>(qb + E, ld3, Y, tid);
    at_stage<AT_Q, BF>(qb + 2 * E, ld3, Y2, tid);
  } else {
    at_stage<AT_RB, BF>(qb + E + (size_t)blk * AT_RB * ld3, ld3, X, tid);
    at_stage<AT_RB, BF>(qb + 2 * E + (size_t)blk * AT_RB * ld3, ld3, X2, tid);
    at_stage<AT_Q, BF>(qb, ld3, Y, tid);
    at_stage<AT_Q, BF>(dob, E, Y2, tid);
  }
  {  // per query: the forward's statistics and D = dO . O (fp32 rows as stored)
    const int i = tid;
    s_mx[i] = a.stats[((size_t)bh * AT_Q + i) * 2];
    s_inv[i] = a.stats[((size_t)bh * AT_Q + i) * 2 + 1];
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < AT_D / 4; ++c) {
      const float4 x = *reinterpret_cast<const float4*>(dob + (size_t)i * E + 4 * c);
      const float4 y = *reinterpret_cast<const float4*>(ob + (size_t)i * E + 4 * c);
      d = __builtin_fmaf(x.x, y.x, d); d = __builtin_fmaf(x.y, y.y, d);
      d = __builtin_fmaf(x.z, y.z, d); d = __builtin_fmaf(x.w, y.w, d);
    }
    s_D[i] = d;
  }
  __syncthreads();
  const int li = lane & 15, g = lane >> 4;
  const float inv_keep = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.0f;
  const uint32_t h0 = a.p > 0.f ? at_hash_prefix(a.rng, a.op) : 0u;
  f32x4 sacc[16], dacc[16];
  at_phase_a(X, Y, wave, lane, sacc);        // rows: q . k^T          keys: k . q^T  (the same bits, transposed)
  at_phase_a(X2, Y2, wave, lane, dacc);      // rows: dO . v^T = dPd   keys: v . dO^T
  // sacc <- dropout(P) (rounded), dacc <- dS (rounded), element by element
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int yc = 16 * t + li;                                          // column of Y
    float cmx = 0.f, cinv = 0.f, cD = 0.f;
    if (role == 1) { cmx = s_mx[yc]; cinv = s_inv[yc]; cD = s_D[yc]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int xr = blk * AT_RB + 16 * wave + 4 * g + r;                // row of X
      const int qi = role == 0 ? xr : yc, kj = role == 0 ? yc : xr;      // (query, key)
      const float mx = role == 0 ? s_mx[xr] : cmx, inv = role == 0 ? s_inv[xr] : cinv, D = role == 0 ? s_D[xr] : cD;
      const float pr = __expf(sacc[t][r] * a.scale - mx) * inv;
      bool keep = true;
      if (a.p > 0.f) keep = at_keep(h0, ((size_t)bh * AT_Q + qi) * AT_Q + kj, a.p);
      const float dp = keep ? dacc[t][r] * inv_keep : 0.f;
      sacc[t][r] = at_rnd<BF>(keep ? pr * inv_keep : 0.f);
      dacc[t][r] = at_rnd<BF>(pr * (dp - D));
    }
  }
  auto put = [&](const f32x4 v[16]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 16; ++t) W[(16 * wave + 4 * g + r) * AT_LDW + 16 * t + li] = v[t][r];
  };
  f32x4 out[2];
  float* dq = a.dqkv + (size_t)b * AT_Q * ld3 + h * AT_D;
  // (a wave reads back only the 16 rows of W it wrote itself; the barriers order its own LDS traffic)
  if (role == 0) {
    put(dacc);
    __syncthreads();
    at_phase_b(W, Y, wave, lane, out);                       // dq = scale dS . k
    at_store_b(out, a.scale, dq + (size_t)blk * AT_RB * ld3, ld3, wave, lane);
  } else {
    put(sacc);
    __syncthreads();
    at_phase_b(W, Y2, wave, lane, out);                      // dv = Pd^T . dO
    at_store_b(out, 1.0f, dq + 2 * E + (size_t)blk * AT_RB * ld3, ld3, wave, lane);
    __syncthreads();
    put(dacc);
    __syncthreads();
    at_phase_b(W, Y, wave, lane, out);                       // dk = scale dS^T . q
    at_store_b(out, a.scale, dq + E + (size_t)blk * AT_RB * ld3, ld3, wave, lane);
  }
}

static int attn_check(int B, int H, int Q, int Dh) {
  DEMF_REQUIRE(B >= 0 && H >= 1, "attn_core: bad sizes B=%d H=%d", B, H);
  if (Q != AT_Q || Dh != AT_D) {
    set_error("attn_core: built for %d queries x %d channels per head (got %d x %d)", AT_Q, AT_D, Q, Dh);
    return DEMF_EUNSUPPORTED;
  }
  return DEMF_OK;
}

template <typename K>
static int attn_lds(K kernel, const char* what) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(AT_LDS_FLOATS * sizeof(float))) != hipSuccess) {
    set_error("%s: cannot reserve %zu bytes of LDS", what, AT_LDS_FLOATS * sizeof(float));
    return DEMF_ELAUNCH;
  }
  return DEMF_OK;
}

}  // namespace demf

using namespace demf;

extern "C" int demf_attn_core_fwd(int B, int H, int Q, int Dh, const float* qkv, float scale, float p,
                                  const void* rng, int op_id, float* out, float* stats, float* prob,
                                  float* prob_dropped, demf_stream_t stream) {
  if (int e = attn_check(B, H, Q, Dh)) return e;
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(qkv && out && stats && (p == 0.f || rng) && p >= 0.f && p < 1.f && (!prob == !prob_dropped),
               "attn_core_fwd: bad arguments");
  AttnArgs a{};
  a.B = B; a.H = H; a.qkv = qkv; a.o = out; a.stats = stats; a.prob = prob; a.pdrop = prob_dropped;
  a.scale = scale; a.p = p; a.rng = (const unsigned long long*)rng; a.op = (unsigned)op_id;
  static bool configured = false;
  if (!configured) {
    if (int e = attn_lds(&attn_fwd_kernel<false>, "attn_core_fwd")) return e;
    if (int e = attn_lds(&attn_fwd_kernel<true>, "attn_core_fwd")) return e;
    configured = true;
  }
  const size_t bytes = AT_LDS_FLOATS * sizeof(float);
  if (compute_bf16()) hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(B * H * 4), dim3(256), bytes, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(B * H * 4), dim3(256), bytes, (hipStream_t)stream, a);
  return check_launch("attn_core_fwd");
}

extern "C" int demf_attn_core_bwd(int B, int H, int Q, int Dh, const float* qkv, const float* out,
                                  const float* dout, const float* stats, float scale, float p, const void* rng,
                                  int op_id, float* dqkv, demf_stream_t stream) {
  if (int e = attn_check(B, H, Q, Dh)) return e;
  if (B == 0) return DEMF_OK;
  DEMF_REQUIRE(qkv && out && dout && stats && dqkv && (p == 0.f || rng) && p >= 0.f && p < 1.f,
               "attn_core_bwd: bad arguments");
  AttnArgs a{};
  a.B = B; a.H = H; a.qkv = qkv; a.out = out; a.dout = dout; a.stats = const_cast<float*>(stats); a.dqkv = dqkv;
  a.scale = scale; a.p = p; a.rng = (const unsigned long long*)rng; a.op = (unsigned)op_id;
  static bool configured = false;
  if (!configured) {
    if (int e = attn_lds(&attn_bwd_kernel<false>, "attn_core_bwd")) return e;
    if (int e = attn_lds(&attn_bwd_kernel<true>, "attn_core_bwd")) return e;
    configured = true;
  }
  const size_t bytes = AT_LDS_FLOATS * sizeof(float);
  if (compute_bf16()) hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(B * H * 8), dim3(256), bytes, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(B * H * 8), dim3(256), bytes, (hipStream_t)stream, a);
  return check_launch("attn_core_bwd");
}
