// Shared helpers for the gfx950 kernels of libdemf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/demf_hip.h"

namespace demf {

// ---- error plumbing (thread-local text, C ABI return codes) --------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define DEMF_REQUIRE(cond, ...)             \
  do {                                      \
    if (!(cond)) {                          \
      ::demf::set_error(__VA_ARGS__);       \
      return DEMF_EINVAL;                   \
    }                                       \
  } while (0)

// ---- canonical arithmetic -------------------------------------------------
// Squared distance used by FPS / ball query / three_nn.  The upstream CUDA
// sources spell it  dx*dx + dy*dy + dz*dz  and are built with nvcc's default
// -fmad=true; this library (built with -ffp-contract=off) and the CPU oracle
// both pin the contraction explicitly to
//     fma(dz, dz, fma(dx, dx, dy*dy))
// so index outputs are bit-reproducible between the two.  See DESIGN.md.
__device__ __forceinline__ float dist2(float dx, float dy, float dz) {
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
}

// ---- wave64 cross-lane helpers (DPP; gfx9 row_bcast forms) ----------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                         __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                         0xF, false));
}

// max over each 16-lane row, result in every lane of the row
__device__ __forceinline__ float row16_allmax(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v, v));   // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f32<0x4E>(v, v));   // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f32<0x141>(v, v));  // row_half_mirror
  v = fmaxf(v, dpp_f32<0x140>(v, v));  // row_mirror
  return v;
}

// max over the whole wave, returned wave-uniform
__device__ __forceinline__ float wave_allmax(float v) {
  v = row16_allmax(v);
  v = fmaxf(v, dpp_f32<0x142, 0xA>(v, v));  // row_bcast:15 -> rows 1,3
  v = fmaxf(v, dpp_f32<0x143, 0xC>(v, v));  // row_bcast:31 -> rows 2,3
  return __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// sum over aligned groups of G lanes (G in {1,2,4,8,16,32,64}); every lane of the
// group receives the total.
template <int G>
__device__ __forceinline__ float group_allsum(float v) {
  if constexpr (G >= 2) v += dpp_f32<0xB1>(v, v);
  if constexpr (G >= 4) v += dpp_f32<0x4E>(v, v);
  if constexpr (G >= 8) v += dpp_f32<0x141>(v, v);
  if constexpr (G >= 16) v += dpp_f32<0x140>(v, v);
  if constexpr (G >= 32) v += __shfl_xor(v, 16);
  if constexpr (G >= 64) v += __shfl_xor(v, 32);
  return v;
}

__device__ __forceinline__ int readlane_i(int v, int lane) {
  return __builtin_amdgcn_readlane(v, lane);
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ int lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt,
// i.e. stalls until every outstanding global STORE of the wave has been acknowledged by memory - in a
// GEMM whose epilogue has just issued 64 stores per lane that is microseconds per tile.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Workgroup barrier in front of a "last workgroup" ticket (csrc/bn_fin.h): EVERY wave first waits for its own
// vector-memory operations - the sum atomics it has just issued - to be acknowledged.  __syncthreads() alone does
// not do that on this target: outside threadgroup-split mode the workgroup-scope release omits s_waitcnt vmcnt(0)
// (checked in the ISA: global_atomic_add_f64 ... s_barrier with no wait in between), so the thread that takes the
// ticket only knew about its OWN wave's atomics; a workgroup could finalise a layer's statistics while another
// wave's contribution to them was still in flight - and the straggler then landed in the self-cleaning accumulator.
__device__ __forceinline__ void sync_drained() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// One-time hipFuncSetAttribute(MaxDynamicSharedMemorySize) for kernels that need more than 64 KB of LDS, recorded
// PER DEVICE (bit d of `done`: the attribute is a property of the function on one device) and safe to call from
// several host threads (the call is idempotent; the atomic only avoids repeating it).  false: the device does not
// offer that much LDS or the call failed - the caller falls back to a form that needs less.
static inline bool reserve_lds(const void* fn, int bytes, unsigned long long* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  const unsigned long long bit = 1ull << dev;
  if (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit) return true;
  int max_lds = 0;
  if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || max_lds < bytes)
    return false;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
  return true;
}

// compute dtype switch (demf_set_compute_dtype, csrc/mlp.hip): true = bf16 MFMA, fp32 accumulate
bool compute_bf16();
int compute_mode();   // 0 fp32 MFMA, 1 bf16 MFMA, 2 fp32 as three bf16 terms (mlp.hip)

// ---- fp32 operands as TWO fp16 terms (kernel template mode CM = 3, P = 2 planes) ------------------------------------
// x = h + l with h = fp16(x), l = fp16(x - h): 22 significant bits when |x| lies in fp16's normal range, an absolute
// error of 2^-25 below it.  The three products h.h', h.l', l.h' on v_mfma_f32_32x32x16_f16 (same rate as the bf16
// instruction) leave out terms of relative weight 2^-22: half the matrix work of the six-product three-term bf16
// split (x = h + m + l exactly, terms below 2^-24 left out), an error of the size of fp32's own accumulation noise
// over K >= 16.  What fp16 does NOT have is range: operands must sit within [2^-1, 65504] at their LARGEST for the
// absolute error to stay below 2^-24 of that largest value.  Activations (BatchNorm outputs) and weights do; gradient
// operands are scaled by a power of two per slab before the split and the result is scaled back (csrc/mlp_bwd.hip).
// Values beyond +-65504 are clamped (a finite wrong product instead of inf - inf = NaN).
bool f16_terms();     // host: DEMF_F16_TERMS (default 1) and compute_mode() == 2
using f16x2_v = _Float16 __attribute__((ext_vector_type(2)));
using f16x8_v = _Float16 __attribute__((ext_vector_type(8)));
using f32x2_v = float __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_f16(float a, float b, unsigned& hi, unsigned& lo) {
  f32x2_v x = {__builtin_fminf(__builtin_fmaxf(a, -65504.f), 65504.f), __builtin_fminf(__builtin_fmaxf(b, -65504.f), 65504.f)};
  const f16x2_v h = __builtin_convertvector(x, f16x2_v);
  hi = __builtin_bit_cast(unsigned, h);
  const f32x2_v r = x - __builtin_convertvector(h, f32x2_v);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_v));
}

}  // namespace demf
