// Convolutions of the frozen image stream on gfx950: implicit GEMM over channels-last (NHWC) activations.
//
// Reference: mmdet ResNet-50 (pytorch style, frozen BatchNorm, out_indices 1-3) + ChannelMapper (1x1 / 3x3 s2
// convolutions + GroupNorm(32)) as the reference configures them in configs/deformdetr/imvotenet_image.py:3-20 and
// runs them, under no_grad and in eval mode, at demf/modeling/detectors/demfnet.py:124-132.  Upstream these are
// library convolutions (cuDNN there, MIOpen on ROCm: 16.5 ms at 8 x 3 x 800 x 1120 fp32, the largest block of the
// end-to-end step since round 4).
//
// Here a convolution is the long-row GEMM of csrc/rows_gemm.hip with the A rows GATHERED: output pixel
// p = (b, ho, wo) is GEMM row p, the reduction index runs over (kh, kw, c) and a 32-wide reduction step lies inside
// ONE tap (Cin % 32 == 0), so the A tile of a step is, per row, 128 contiguous bytes of the input pixel
// (ho*s - pad + kh, wo*s - pad + kw) - or zeros outside the image.  Nothing is unfolded in memory.  128-row x
// (128 | 64)-column tiles, 4 waves as 2 x 2, K steps of 32 through one LDS stage, three workgroups per CU
// rotating through load / split / MFMA; fp32-grade arithmetic is the three-term split (x = h + m + l in bf16,
// six products on v_mfma_f32_32x32x16_bf16, fp32 accumulate - csrc/mlp.hip), the frozen weights arrive pre-split
// as bf16 planes (P, Cout, KH*KW*Cin) with the BatchNorm scale folded in; P = 1 is the bf16 compute mode.
// Epilogue: folded-BN bias, the bottleneck's residual add, ReLU.
// The 7x7 stride-2 stem (Cin = 3) reads the image as NHWC4 (a zero fourth channel): one reduction step = one
// kernel ROW = 8 pixels x 4 channels = 128 contiguous bytes starting at (ho*2 - 3 + kh, wo*2 - 3); the weight
// has zeros at kw = 7 and c = 3, so the same tile loop serves it with K = 7 * 32.
// Also here: the 3x3 stride-2 max-pool, NCHW -> NHWC4 of the input image, and GroupNorm over NHWC rows written
// straight into the encoder's token buffer (B, S, 256) - the channels-last hand-over needs no transposition.
#include "common.h"
#include <stdlib.h>

namespace demf {

using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x2 = float __attribute__((ext_vector_type(2)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));

struct ConvArgs {
  int B, H, W, Cin;          // input (B, H, W, Cin) fp32; STEM: (B, H, W, 4)
  int Ho, Wo, Cout;
  int KH, KW, stride, pad;
  int K;                     // reduction length: KH * KW * Cin (STEM: KH * 32)
  const float* X;
  const __bf16* Wp;          // (P, Cout, K) bf16 planes
  const float* bias;         // (Cout) or null
  const float* resid;        // (B*Ho*Wo, Cout) or null: added before the ReLU
  int relu;
  float* Y;                  // (B*Ho*Wo, Cout)
  int ksplit;                // > 1: blockIdx.y owns reduction steps [y * ksteps, (y + 1) * ksteps) and ADDS its
  int ksteps;                //      partial tile into a zeroed Y (no bias / residual / ReLU)
};

template <int P>
__device__ __forceinline__ void cv_split_pair(float a, float b, unsigned (&o)[P]) {
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
// 64-byte plane rows, 16-byte chunks XOR-swizzled by bits 2-3 of the row (as rg_swz in csrc/rows_gemm.hip)
__device__ __forceinline__ int cv_swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int P>
__device__ __forceinline__ void cv_mfma(f32x16& acc, const bf16x8 (&a)[P], const bf16x8 (&b)[P]) {
  if constexpr (P == 3) {   // smallest products first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

constexpr int CV_BM = 128;

// BN = 128: waves 2 x 2 of 64 x 64; BN = 64: waves 2 x 2 of 64 x 32.
// WG3: three workgroups per CU also at three planes (168 registers: a few spills)
template <int P, int BN, bool STEM, bool WG3 = false>
__global__ __launch_bounds__(256, (P == 3 && !WG3) ? 2 : 3) void conv_nhwc_kernel(ConvArgs p) {
  constexpr int NTW = BN / 64;                  // 32-column tiles per wave
  constexpr int NJ = BN / 64;                   // B staging: columns (t >> 2) + 64 j
  constexpr int A_BYTES = CV_BM * 64, B_BYTES = BN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 31, lh = lane >> 5, wm = wave & 1, wn = wave >> 1;
  // all column tiles of a row block on ONE XCD (its gathered A rows stay in that L2)
  const int gx = p.Cout / BN;
  const int L = blockIdx.x;
  const int by = (L & 7) + 8 * (L / (8 * gx)), bx = (L >> 3) % gx;
  const int M = p.B * p.Ho * p.Wo;
  const int m0 = by * CV_BM, n0 = bx * BN;
  if (m0 >= M) return;
  // staging: A - float4 kq = t & 7 of rows (t >> 3) + 32 i; B - 16-byte chunk (t & 3) of columns (t >> 2) + 64 j
  const int ar = t >> 3, akq = t & 7, bc = t >> 2, bch = t & 3;
  // the thread's four output pixels: top-left input coordinate (two 16-bit halves of one register) and its element
  // offset (32-bit, may be negative: only dereferenced where the tap lies inside the image)
  int hw0[4], off[4];
  const int cpp = STEM ? 4 : p.Cin;             // floats per input pixel
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + ar + 32 * i;
    if (row < M) {
      const int b = row / (p.Ho * p.Wo), rem = row - b * (p.Ho * p.Wo);
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
      hw0[i] = (h0 << 16) | (w0 & 0xffff);
      off[i] = ((b * p.H + h0) * p.W + w0) * cpp;
    } else {
      hw0[i] = (int)0xc000c000u; off[i] = 0;               // never inside the image
    }
  }
  float4 ra[4];
  u32x4 rb[P * NJ];
  // this workgroup's reduction range (split-K: blockIdx.y) and the tap of its first step (uniform); the tap is
  // advanced incrementally, 32 channels at a time
  const int nsteps = p.K / 32;
  const int s_lo = p.ksplit > 1 ? (int)blockIdx.y * p.ksteps : 0;
  const int s_hi = p.ksplit > 1 ? min(nsteps, s_lo + p.ksteps) : nsteps;
  if (s_lo >= s_hi) return;
  int f_kh = 0, f_kw = 0, f_c0 = 0;
  if constexpr (STEM) {
    f_kh = s_lo;
  } else {
    const int tap0 = (32 * s_lo) / p.Cin;
    f_c0 = 32 * s_lo - tap0 * p.Cin;
    f_kh = tap0 / p.KW;
    f_kw = tap0 - f_kh * p.KW;
  }
  auto fetch = [&](int k0) {
    const int tap = STEM ? f_kh * p.W * 4 : (f_kh * p.W + f_kw) * p.Cin + f_c0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hi = (hw0[i] >> 16) + f_kh;
      const int wi = (int)(short)(hw0[i] & 0xffff) + (STEM ? akq : f_kw);
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
        ra[i] = *reinterpret_cast<const float4*>(p.X + off[i] + tap + 4 * akq);
    }
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        rb[q * NJ + j] = *reinterpret_cast<const u32x4*>(p.Wp + ((size_t)q * p.Cout + n0 + bc + 64 * j) * p.K + k0 + 8 * bch);
    if constexpr (STEM) {
      ++f_kh;
    } else {
      f_c0 += 32;
      if (f_c0 == p.Cin) { f_c0 = 0; if (++f_kw == p.KW) { f_kw = 0; ++f_kh; } }
    }
  };
  auto commit = [&]() {
    char* sa = smem;
    char* sb = sa + P * A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned lo[P], hi[P];
      cv_split_pair<P>(ra[i].x, ra[i].y, lo);
      cv_split_pair<P>(ra[i].z, ra[i].w, hi);
      const int o = cv_swz(ar + 32 * i, akq >> 1) + 8 * (akq & 1);
#pragma unroll
      for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(sa + q * A_BYTES + o) = make_uint2(lo[q], hi[q]);
    }
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        *reinterpret_cast<u32x4*>(sb + q * B_BYTES + cv_swz(bc + 64 * j, bch)) = rb[q * NJ + j];
  };
  f32x16 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  fetch(32 * s_lo);
  for (int k0 = 32 * s_lo; k0 < 32 * s_hi; k0 += 32) {
    commit();
    lds_barrier();
    if (k0 + 32 < 32 * s_hi) fetch(k0 + 32);
    const char* sa = smem;
    const char* sb = sa + P * A_BYTES;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 pa[2][P];
#pragma unroll
      for (int q = 0; q < P; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          pa[i][q] = *reinterpret_cast<const bf16x8*>(sa + q * A_BYTES + cv_swz(64 * wm + 32 * i + lc, 2 * c + lh));
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        bf16x8 pb[P];
#pragma unroll
        for (int q = 0; q < P; ++q)
          pb[q] = *reinterpret_cast<const bf16x8*>(sb + q * B_BYTES + cv_swz(32 * NTW * wn + 32 * j + lc, 2 * c + lh));
#pragma unroll
        for (int i = 0; i < 2; ++i) cv_mfma<P>(acc[i][j], pa[i], pb);
      }
    }
    lds_barrier();
  }
  // Epilogue through LDS (the stage buffers are free now): the accumulator tiles of one 64-row half at a time
  // become an fp32 tile, then every thread moves float4 pieces of whole rows - bias, the residual (read as
  // float4), ReLU, one 16-byte store.  (Straight from the accumulators a lane issues 64 4-byte stores and as
  // many 4-byte residual loads: the 64 -> 256 convolution of layer1 ran at 1.4 TB/s that way.)
  // accumulator register r of tile (i, j) = row 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 lh, column 32 NTW wn + 32 j + lc
  constexpr int LDP = BN + 4;                     // floats per tile row
  constexpr int TPR = BN / 4;                     // threads per row (one float4 each)
  constexpr int RPP = 256 / TPR;                  // rows per pass
  float* tile = reinterpret_cast<float*>(smem);
  const int e_cq = t % TPR, e_r0 = t / TPR;
  const int col4 = n0 + 4 * e_cq;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias != nullptr && p.ksplit <= 1) b4 = *reinterpret_cast<const float4*>(p.bias + col4);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tile[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDP + 32 * NTW * wn + 32 * j + lc] = acc[i][j][r];
    }
    __syncthreads();
    for (int rr = e_r0; rr < 64; rr += RPP) {
      const int row = m0 + 64 * half + rr;
      if (row >= M) break;
      float4 v = *reinterpret_cast<const float4*>(tile + rr * LDP + 4 * e_cq);
      float* dst = p.Y + (size_t)row * p.Cout + col4;
      if (p.ksplit > 1) {
        atomicAdd(dst, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
        continue;
      }
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      if (p.resid != nullptr) {
        const float4 q = *reinterpret_cast<const float4*>(p.resid + (size_t)row * p.Cout + col4);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(dst) = v;
    }
  }
}

template <int P, int BN, bool STEM, bool WG3 = false>
static int conv_launch(const ConvArgs& a, hipStream_t s) {
  constexpr int stage = P * (CV_BM * 64 + BN * 64), epi = 64 * (BN + 4) * 4;
  constexpr int lds = stage > epi ? stage : epi;
  const int M = a.B * a.Ho * a.Wo;
  const int gx = a.Cout / BN, gy = (cdiv(M, CV_BM) + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_nhwc_kernel<P, BN, STEM, WG3>), dim3(gx * gy, a.ksplit > 1 ? a.ksplit : 1), dim3(256), lds, s, a);
  return check_launch("conv_nhwc_kernel");
}

// ---- 1 x 1, stride 1, Cin = 64 (ResNet-50's first stage: 64 -> 256 with the residual, 64 -> 64) --------------------------
// These layers are HBM byte movers (1.03 GB per launch at 8 x 200 x 280 pixels for 14.7 GFLOP) and the tile kernel ran
// them at 2.7 TB/s: two reduction steps per tile, then an epilogue whose residual loads start when the MFMAs are done.
// Here the whole (Cout x 64) weight is staged ONCE per workgroup as bf16 planes (Cout = 256, three planes: 96 KB) and
// every wave walks its own 32-pixel tiles without a barrier: the pixel rows go global -> registers -> fragments (a lane
// holds exactly the 8 consecutive channels of its pixel an MFMA step wants - no LDS for A), the next tile's rows and
// the next column tile's residual are requested before the current MFMAs are issued, and a column tile (K = 64 is the
// whole reduction) is finished and stored 32 columns = one 128-byte line per row at a time.
__device__ __forceinline__ int cv_sw128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int P, int NT>
__global__ __launch_bounds__(512, 1) void conv1x1_c64_stream_kernel(ConvArgs p) {
  constexpr int COUT = 32 * NT, PLANE = COUT * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [P][COUT][64 bf16]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lc = lane & 31, lh = lane >> 5;
  for (int c = t; c < P * COUT * 8; c += 512) {
    const int q = c / (COUT * 8), rem = c - q * COUT * 8, row = rem >> 3, ch = rem & 7;
    *reinterpret_cast<u32x4*>(smem + q * PLANE + cv_sw128(row, ch)) =
        *reinterpret_cast<const u32x4*>(p.Wp + ((size_t)q * COUT + row) * 64 + 8 * ch);
  }
  __syncthreads();
  float* const epi = reinterpret_cast<float*>(smem + P * PLANE) + wave * (32 * 36);     // this wave's 32 x 32 (+4) fp32 tile
  const int M = p.B * p.Ho * p.Wo;
  const int ntile = (M + 31) / 32, nw = (int)gridDim.x * 8;
  int tile = (int)blockIdx.x * 8 + wave;
  float4 ra[8];
  auto fetch = [&](int tl) {
    const int row = min(tl * 32 + lc, M - 1);                      // (rows beyond M: a valid address, never stored)
    const float* src = p.X + (size_t)row * 64 + 8 * lh;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      ra[2 * s4] = *reinterpret_cast<const float4*>(src + 16 * s4);
      ra[2 * s4 + 1] = *reinterpret_cast<const float4*>(src + 16 * s4 + 4);
    }
  };
  if (tile < ntile) fetch(tile);
  for (; tile < ntile; tile += nw) {
    bf16x8 pa[4][P];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      unsigned o0[P], o1[P], o2[P], o3[P];
      cv_split_pair<P>(ra[2 * s4].x, ra[2 * s4].y, o0);
      cv_split_pair<P>(ra[2 * s4].z, ra[2 * s4].w, o1);
      cv_split_pair<P>(ra[2 * s4 + 1].x, ra[2 * s4 + 1].y, o2);
      cv_split_pair<P>(ra[2 * s4 + 1].z, ra[2 * s4 + 1].w, o3);
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const u32x4 v = {o0[q], o1[q], o2[q], o3[q]};
        pa[s4][q] = __builtin_bit_cast(bf16x8, v);
      }
    }
    if (tile + nw < ntile) fetch(tile + nw);
    const int row0 = tile * 32;
    const bool full = row0 + 32 <= M;
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {
      const int col = 32 * j + lc;
      // the finished 32 x 32 tile leaves through the wave's private LDS tile as float4 rows: lane -> row (lane >> 3) + 8 i,
      // columns 4 (lane & 7) .. + 3 - four 16-byte stores per column tile instead of sixteen 4-byte ones, and the residual
      // arrives the same way (requested before the MFMAs are issued)
      const int er = lane >> 3, ec = 32 * j + 4 * (lane & 7);
      float4 rs[4];
      if (p.resid != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          rs[i] = *reinterpret_cast<const float4*>(p.resid + (size_t)min(row0 + er + 8 * i, M - 1) * COUT + ec);
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        bf16x8 pb[P];
#pragma unroll
        for (int q = 0; q < P; ++q)
          pb[q] = *reinterpret_cast<const bf16x8*>(smem + q * PLANE + cv_sw128(col, 2 * s4 + lh));
        cv_mfma<P>(acc, pa[s4], pb);
      }
      const float bias = p.bias != nullptr ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) epi[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + lc] = acc[r] + bias;
      // (one wave: its own ds_writes are complete before its ds_reads are issued - program order on the LDS queue)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + er + 8 * i;
        float4 v = *reinterpret_cast<const float4*>(epi + (er + 8 * i) * 36 + 4 * (lane & 7));
        if (p.resid != nullptr) { v.x += rs[i].x; v.y += rs[i].y; v.z += rs[i].z; v.w += rs[i].w; }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (full || row < M) *reinterpret_cast<float4*>(p.Y + (size_t)row * COUT + ec) = v;
      }
    }
  }
}

template <int P, int NT>
static int conv1x1_c64_launch(const ConvArgs& a, hipStream_t s) {
  constexpr int lds = P * 32 * NT * 128 + 8 * 32 * 36 * 4;
  static unsigned long long attr_done = 0;       // bit per device
  if (!reserve_lds(reinterpret_cast<const void*>(&conv1x1_c64_stream_kernel<P, NT>), lds, &attr_done))
    return -1000;                                // caller: the tile kernel
  const int M = a.B * a.Ho * a.Wo;
  const int wgs = min(256, cdiv(cdiv(M, 32), 8));
  hipLaunchKernelGGL((conv1x1_c64_stream_kernel<P, NT>), dim3(wgs), dim3(512), lds, s, a);
  return check_launch("conv1x1_c64_stream_kernel");
}

// The same form for 1 x 1, stride 1 layers with 128 / 256 input channels (ResNet-50's 128 -> 512 and 256 -> 1024 expansions
// with the residual, 256 -> 64 / 128 reductions): a workgroup keeps the weight rows of ONE group of 32 NT output channels
// (blockIdx.y; 96 KB of planes at NT x KC = 8), the reduction is streamed in chunks of 64 channels with the next chunk's
// rows in flight, the NT accumulator tiles stay in registers until the last chunk.
template <int KB>
__device__ __forceinline__ int cv_swk(int row, int chunk) { return row * KB + ((chunk ^ (row & (KB / 16 - 1))) << 4); }

template <int P, int NT, int KC>
__global__ __launch_bounds__(512, 1) void conv1x1_stream_kernel(ConvArgs p) {
  constexpr int COLS = 32 * NT, K = 64 * KC, KB = 2 * K, PLANE = COLS * KB;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [P][COLS][K bf16], then 8 fp32 tiles of 32 x 36
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lc = lane & 31, lh = lane >> 5;
  const int n0 = (int)blockIdx.y * COLS;
  for (int c = t; c < P * COLS * (KB / 16); c += 512) {
    const int q = c / (COLS * (KB / 16)), rem = c - q * COLS * (KB / 16), row = rem / (KB / 16), ch = rem % (KB / 16);
    *reinterpret_cast<u32x4*>(smem + q * PLANE + cv_swk<KB>(row, ch)) =
        *reinterpret_cast<const u32x4*>(p.Wp + ((size_t)q * p.Cout + n0 + row) * K + 8 * ch);
  }
  __syncthreads();
  float* const epi = reinterpret_cast<float*>(smem + P * PLANE) + wave * (32 * 36);
  const int M = p.B * p.Ho * p.Wo;
  const int ntile = (M + 31) / 32, nw = (int)gridDim.x * 8;
  int tile = (int)blockIdx.x * 8 + wave;
  float4 ra[8];
  auto fetch = [&](int tl, int kc) {
    const int row = min(tl * 32 + lc, M - 1);
    const float* src = p.X + (size_t)row * K + 64 * kc + 8 * lh;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      ra[2 * s4] = *reinterpret_cast<const float4*>(src + 16 * s4);
      ra[2 * s4 + 1] = *reinterpret_cast<const float4*>(src + 16 * s4 + 4);
    }
  };
  if (tile < ntile) fetch(tile, 0);
  for (; tile < ntile; tile += nw) {
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < KC; ++kc) {
      bf16x8 pa[4][P];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        unsigned o0[P], o1[P], o2[P], o3[P];
        cv_split_pair<P>(ra[2 * s4].x, ra[2 * s4].y, o0);
        cv_split_pair<P>(ra[2 * s4].z, ra[2 * s4].w, o1);
        cv_split_pair<P>(ra[2 * s4 + 1].x, ra[2 * s4 + 1].y, o2);
        cv_split_pair<P>(ra[2 * s4 + 1].z, ra[2 * s4 + 1].w, o3);
#pragma unroll
        for (int q = 0; q < P; ++q) {
          const u32x4 v = {o0[q], o1[q], o2[q], o3[q]};
          pa[s4][q] = __builtin_bit_cast(bf16x8, v);
        }
      }
      if (kc + 1 < KC) fetch(tile, kc + 1);
      else if (tile + nw < ntile) fetch(tile + nw, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          bf16x8 pb[P];
#pragma unroll
          for (int q = 0; q < P; ++q)
            pb[q] = *reinterpret_cast<const bf16x8*>(smem + q * PLANE + cv_swk<KB>(32 * j + lc, 8 * kc + 2 * s4 + lh));
          cv_mfma<P>(acc[j], pa[s4], pb);
        }
    }
    const int row0 = tile * 32;
    const bool full = row0 + 32 <= M;
    const int er = lane >> 3;
    float4 rs[4];
    auto fetch_resid = [&](int j) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        rs[i] = *reinterpret_cast<const float4*>(p.resid + (size_t)min(row0 + er + 8 * i, M - 1) * p.Cout + n0 + 32 * j + 4 * (lane & 7));
    };
    if (p.resid != nullptr) fetch_resid(0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float bias = p.bias != nullptr ? p.bias[n0 + 32 * j + lc] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) epi[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + lc] = acc[j][r] + bias;
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = *reinterpret_cast<const float4*>(epi + (er + 8 * i) * 36 + 4 * (lane & 7));
        if (p.resid != nullptr) { v[i].x += rs[i].x; v[i].y += rs[i].y; v[i].z += rs[i].z; v[i].w += rs[i].w; }
        if (p.relu) { v[i].x = fmaxf(v[i].x, 0.f); v[i].y = fmaxf(v[i].y, 0.f); v[i].z = fmaxf(v[i].z, 0.f); v[i].w = fmaxf(v[i].w, 0.f); }
      }
      if (p.resid != nullptr && j + 1 < NT) fetch_resid(j + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + er + 8 * i;
        if (full || row < M) *reinterpret_cast<float4*>(p.Y + (size_t)row * p.Cout + n0 + 32 * j + 4 * (lane & 7)) = v[i];
      }
    }
  }
}

template <int P, int NT, int KC>
static int conv1x1_stream_launch(const ConvArgs& a, hipStream_t s) {
  constexpr int lds = P * 32 * NT * 128 * KC + 8 * 32 * 36 * 4;
  static unsigned long long attr_done = 0;       // bit per device
  if (!reserve_lds(reinterpret_cast<const void*>(&conv1x1_stream_kernel<P, NT, KC>), lds, &attr_done))
    return -1000;                                // caller: the tile kernel
  const int M = a.B * a.Ho * a.Wo;
  const int groups = a.Cout / (32 * NT);
  int wgs = 256 / groups;
  wgs = min(wgs < 1 ? 1 : wgs, cdiv(cdiv(M, 32), 8));
  hipLaunchKernelGGL((conv1x1_stream_kernel<P, NT, KC>), dim3(wgs, groups), dim3(512), lds, s, a);
  return check_launch("conv1x1_stream_kernel");
}

// 3x3 stride-2 pad-1 max-pool over NHWC rows, one float4 of channels per thread
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_k(int B, int H, int W, int C, int Ho, int Wo,
                                                           const float* __restrict__ x, float* __restrict__ y) {
  const long long n = (long long)B * Ho * Wo * (C / 4);
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n) return;
  const int c4 = (int)(id % (C / 4));
  long long r = id / (C / 4);
  const int wo = (int)(r % Wo); r /= Wo;
  const int ho = (int)(r % Ho);
  const int b = (int)(r / Ho);
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = 2 * ho - 1 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = 2 * wo - 1 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + hi) * W + wi) * C + 4 * c4);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  *reinterpret_cast<float4*>(y + (((size_t)b * Ho + ho) * Wo + wo) * C + 4 * c4) = m;
}

// (B, 3, H, W) -> (B, H, W, 4) with a zero fourth channel
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_k(long long B, long long HW, const float* __restrict__ x,
                                                        float* __restrict__ y) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * HW) return;
  const long long b = id / HW, q = id - b * HW;
  const float* s = x + b * 3 * HW + q;
  *reinterpret_cast<float4*>(y + 4 * id) = make_float4(s[0], s[HW], s[2 * HW], 0.f);
}

// GroupNorm over channels-last rows (C = 256, G groups of C / G consecutive channels), two launches:
// sums per (image, group) in double (one atomic pair per group per workgroup), then normalise + affine, written to
// rows [row0, row0 + HW) of a (B, S, C) token buffer.
__global__ __launch_bounds__(256) void gn_zero_k(int n, double* __restrict__ a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = 0.0;
}
__global__ __launch_bounds__(256) void gn_stats_nhwc_k(int HW, int C, int G, int rows_per_wg, const float* __restrict__ x,
                                                       double* __restrict__ sums) {
  // a wave per row (64 lanes x float4 = the 256 channels), four rows per step and workgroup, four steps in flight
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(HW, r0 + rows_per_wg);
  const float* p = x + ((size_t)b * HW) * C + 4 * lane;
  float s = 0.f, ss = 0.f;
  for (int r = r0 + wave; r < r1; r += 16) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r + 4 * u < r1) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(r + 4 * u) * C);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
      ss = __builtin_fmaf(v[u].x, v[u].x, __builtin_fmaf(v[u].y, v[u].y, __builtin_fmaf(v[u].z, v[u].z, __builtin_fmaf(v[u].w, v[u].w, ss))));
    }
  }
  // fold the C / G channels of a group: (C / G) / 4 consecutive lanes
  const int lpg = C / G / 4;
  double ds = s, dss = ss;
  for (int m = 1; m < lpg; m <<= 1) {
    ds += __shfl_xor(ds, m);
    dss += __shfl_xor(dss, m);
  }
  // the four waves through LDS, then one atomic pair per group and workgroup
  __shared__ double s_gn[4][64][2];
  s_gn[wave][lane][0] = ds;
  s_gn[wave][lane][1] = dss;
  __syncthreads();
  if (wave == 0 && (lane % lpg) == 0) {
    ds = (s_gn[0][lane][0] + s_gn[1][lane][0]) + (s_gn[2][lane][0] + s_gn[3][lane][0]);
    dss = (s_gn[0][lane][1] + s_gn[1][lane][1]) + (s_gn[2][lane][1] + s_gn[3][lane][1]);
    atomicAdd(sums + ((size_t)b * G + lane / lpg) * 2, ds);
    atomicAdd(sums + ((size_t)b * G + lane / lpg) * 2 + 1, dss);
  }
}
__global__ __launch_bounds__(256) void gn_apply_nhwc_k(int HW, int C, int G, float eps, const float* __restrict__ x,
                                                       const double* __restrict__ sums, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y,
                                                       long long y_batch_stride) {
  const int b = blockIdx.y;
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index inside the image
  if (id >= (long long)HW * (C / 4)) return;
  const int c = 4 * (int)(id % (C / 4));
  const int cg = C / G, g = c / cg;
  const double n = (double)HW * cg;
  const double mean = sums[((size_t)b * G + g) * 2] / n;
  const double var = sums[((size_t)b * G + g) * 2 + 1] / n - mean * mean;
  const float mu = (float)mean, rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)b * HW * C + 4 * id);
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + c);
  const float4 b4 = *reinterpret_cast<const float4*>(beta + c);
  float4 o;
  o.x = (v.x - mu) * rstd * g4.x + b4.x; o.y = (v.y - mu) * rstd * g4.y + b4.y;
  o.z = (v.z - mu) * rstd * g4.z + b4.z; o.w = (v.w - mu) * rstd * g4.w + b4.w;
  *reinterpret_cast<float4*>(y + (size_t)b * y_batch_stride + 4 * id) = o;
}

}  // namespace demf

using namespace demf;

extern "C" int demf_conv_nhwc_f32(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                  const float* x, const void* w_planes, int planes, const float* bias,
                                  const float* resid, int relu, int ksplit, float* y, demf_stream_t stream) {
  DEMF_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 &&
               x != nullptr && w_planes != nullptr && y != nullptr, "conv_nhwc: bad arguments");
  DEMF_REQUIRE(planes == 1 || planes == 3, "conv_nhwc: planes must be 1 (bf16) or 3 (fp32 as three bf16 terms)");
  DEMF_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0, "conv_nhwc: Cin %% 32 and Cout %% 64 required (Cin = %d, Cout = %d)", Cin, Cout);
  ConvArgs a{};
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  a.Ho = (H + 2 * pad - KH) / stride + 1;
  a.Wo = (W + 2 * pad - KW) / stride + 1;
  DEMF_REQUIRE(a.Ho > 0 && a.Wo > 0, "conv_nhwc: empty output");
  DEMF_REQUIRE((long long)B * a.Ho * a.Wo < (1ll << 31) - 1024, "conv_nhwc: too many output pixels");
  a.K = KH * KW * Cin;
  a.X = x; a.Wp = reinterpret_cast<const __bf16*>(w_planes); a.bias = bias; a.resid = resid; a.relu = relu; a.Y = y;
  DEMF_REQUIRE(ksplit >= 1 && ksplit <= 64, "conv_nhwc: ksplit %d", ksplit);
  DEMF_REQUIRE(ksplit == 1 || (bias == nullptr && resid == nullptr && !relu),
               "conv_nhwc: split-K adds partial tiles into a zeroed output: no bias / residual / ReLU");
  a.ksplit = ksplit;
  a.ksteps = cdiv(a.K / 32, ksplit);
  DEMF_REQUIRE((long long)B * H * W * Cin < (1ll << 31) && H < 16384 && W < 16384, "conv_nhwc: input too large for 32-bit offsets");
  hipStream_t s = (hipStream_t)stream;
  static const int stream64 = getenv("DEMF_CONV_STREAM64") ? atoi(getenv("DEMF_CONV_STREAM64")) : 1;   // A/B switch
  if (stream64 && Cin == 64 && KH == 1 && KW == 1 && stride == 1 && pad == 0 && ksplit == 1 && (Cout == 256 || Cout == 64) &&
      (long long)B * H * W >= 16384) {
    // (-1000: this device does not grant the kernel's LDS - the tile kernel below serves the layer)
    const int rc = Cout == 256 ? (planes == 3 ? conv1x1_c64_launch<3, 8>(a, s) : conv1x1_c64_launch<1, 8>(a, s))
                               : (planes == 3 ? conv1x1_c64_launch<3, 2>(a, s) : conv1x1_c64_launch<1, 2>(a, s));
    if (rc != -1000) return rc;
  }
  static const int streamk = getenv("DEMF_CONV_STREAMK") ? atoi(getenv("DEMF_CONV_STREAMK")) : 3;   // A/B switch (bits: 1 K = 128, 2 K = 256)
  static const int stream_min = getenv("DEMF_CONV_STREAM_MIN") ? atoi(getenv("DEMF_CONV_STREAM_MIN")) : 16384;   // rows
  if (streamk && KH == 1 && KW == 1 && stride == 1 && pad == 0 && ksplit == 1 && (long long)B * H * W >= stream_min) {
    int rc = -1000;
    if ((streamk & 1) && Cin == 128 && Cout % 128 == 0)
      rc = planes == 3 ? conv1x1_stream_launch<3, 4, 2>(a, s) : conv1x1_stream_launch<1, 4, 2>(a, s);
    else if ((streamk & 2) && Cin == 256 && Cout % 64 == 0)
      rc = planes == 3 ? conv1x1_stream_launch<3, 2, 4>(a, s) : conv1x1_stream_launch<1, 2, 4>(a, s);
    if (rc != -1000) return rc;
  }
  static const int wg3 = getenv("DEMF_CONV_WG3") ? atoi(getenv("DEMF_CONV_WG3")) : 0;       // A/B switch
  if (Cout % 128 == 0) {
    if (planes == 3) return wg3 ? conv_launch<3, 128, false, true>(a, s) : conv_launch<3, 128, false>(a, s);
    return conv_launch<1, 128, false>(a, s);
  }
  return planes == 3 ? conv_launch<3, 64, false>(a, s) : conv_launch<1, 64, false>(a, s);
}

extern "C" int demf_conv_stem7_nhwc4_f32(int B, int H, int W, int Cout, const float* x4, const void* w_planes,
                                         int planes, const float* bias, int relu, float* y, demf_stream_t stream) {
  DEMF_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 64 == 0 && x4 != nullptr && w_planes != nullptr && y != nullptr,
               "conv_stem7: bad arguments");
  DEMF_REQUIRE(planes == 1 || planes == 3, "conv_stem7: planes must be 1 or 3");
  ConvArgs a{};
  a.B = B; a.H = H; a.W = W; a.Cin = 4; a.Cout = Cout; a.KH = 7; a.KW = 7; a.stride = 2; a.pad = 3;
  a.Ho = (H + 6 - 7) / 2 + 1;
  a.Wo = (W + 6 - 7) / 2 + 1;
  DEMF_REQUIRE((long long)B * a.Ho * a.Wo < (1ll << 31) - 1024 && (long long)B * H * W * 4 < (1ll << 31) && H < 16384 &&
               W < 16384, "conv_stem7: too many pixels");
  a.K = 7 * 32;
  a.ksplit = 1; a.ksteps = 7;
  a.X = x4; a.Wp = reinterpret_cast<const __bf16*>(w_planes); a.bias = bias; a.resid = nullptr; a.relu = relu; a.Y = y;
  hipStream_t s = (hipStream_t)stream;
  if (Cout % 128 == 0) return planes == 3 ? conv_launch<3, 128, true>(a, s) : conv_launch<1, 128, true>(a, s);
  return planes == 3 ? conv_launch<3, 64, true>(a, s) : conv_launch<1, 64, true>(a, s);
}

extern "C" int demf_maxpool3x3s2_nhwc_f32(int B, int H, int W, int C, const float* x, float* y, demf_stream_t stream) {
  DEMF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && x && y, "maxpool3x3s2: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long n = (long long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_nhwc_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     B, H, W, C, Ho, Wo, x, y);
  return check_launch("maxpool3x3s2_nhwc_k");
}

extern "C" int demf_nchw3_to_nhwc4_f32(int B, int H, int W, const float* x, float* y, demf_stream_t stream) {
  DEMF_REQUIRE(B > 0 && H > 0 && W > 0 && x && y, "nchw3_to_nhwc4: bad arguments");
  const long long n = (long long)B * H * W;
  hipLaunchKernelGGL(nchw3_to_nhwc4_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (long long)B, (long long)H * W, x, y);
  return check_launch("nchw3_to_nhwc4_k");
}

extern "C" int demf_groupnorm_nhwc_f32(int B, int HW, int C, int G, float eps, const float* x, const float* gamma,
                                       const float* beta, double* sums, float* y, long long y_batch_stride,
                                       demf_stream_t stream) {
  DEMF_REQUIRE(B > 0 && HW > 0 && C == 256 && G > 0 && C % G == 0 && (C / G) % 4 == 0 && ((C / G) & (C / G - 1)) == 0 &&
               C / G <= 64 && x && gamma && beta && sums && y, "groupnorm_nhwc: bad arguments (C must be 256)");
  hipStream_t s = (hipStream_t)stream;
  // (a kernel, not hipMemsetAsync: memset nodes inside a captured hipGraph were found not to be ordered with the
  // kernels around them - csrc/group_gather.hip, demf_invert_index_ws)
  hipLaunchKernelGGL(gn_zero_k, dim3(cdiv(2 * B * G, 256)), dim3(256), 0, s, 2 * B * G, sums);
  // ~1 024 workgroups at most: a workgroup ends with 2 G atomics onto the image's 2 G sums
  int rows_per_wg = cdiv((int)(((long long)HW * B + 1023) / 1024), 16) * 16;
  if (rows_per_wg < 64) rows_per_wg = 64;
  hipLaunchKernelGGL(gn_stats_nhwc_k, dim3(cdiv(HW, rows_per_wg), B), dim3(256), 0, s, HW, C, G, rows_per_wg, x, sums);
  if (int e = check_launch("gn_stats_nhwc_k")) return e;
  const long long n = (long long)HW * (C / 4);
  hipLaunchKernelGGL(gn_apply_nhwc_k, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, s, HW, C, G, eps, x, sums, gamma,
                     beta, y, y_batch_stride);
  return check_launch("gn_apply_nhwc_k");
}
