// "Last workgroup finishes the job": the per-channel vectors of a BatchNorm backward, formed by the
// LAST workgroup of the launch that produced the layer's two sums (sum dZ, sum dZ*xhat) instead of by a
// separate ~5 us launch behind it (25 of them per training step until round 3).
//
// Reference: torch.nn.BatchNorm2d's train-mode backward inside mmdet3d's ConvModule stacks
// (configs/demf/demf_votenet.py:48-62); dY = gi*dZ + a*y + b with the scalars below.
#pragma once
#include "common.h"

namespace demf {

constexpr int BV_GROUPS = 16;   // two-level exit count: members of a group first, then the group-lasts
constexpr int BV_OFF = 40;      // ints [BV_OFF, BV_OFF + 1 + BV_GROUPS) of a counter set (as the forward's finalize)

struct BnVecFin {
  double count;                 // rows the statistics were taken over
  const float* gamma;           // (N)
  const float* ss;              // [scale|shift] (2N)
  const float* mi;              // [mean|invstd] (2N)
  float* vec;                   // out: 5 x N  (scale, shift, gi, a, b)
  float* dgamma;                // out (N)
  float* dbeta;                 // out (N)
  int* ticket;                  // counter set (sched_slot()) or null: no in-kernel finalize
};

// vec layout (struct of arrays, length N each): [0] scale [1] shift [2] gi = gamma*invstd
// [3] a = -gi*invstd*mean(dZ*xhat) [4] b = -gi*mean(dZ) - a*mean   (demf_bn_bwd_vectors)
__device__ __forceinline__ void bn_bwd_vectors_channel(int c, int N, double count, double g1, double g2,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ ss,
                                                       const float* __restrict__ mi, float* __restrict__ vec,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
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

// Call from ONE thread of every workgroup, after the workgroup's own atomics have been issued and a
// __syncthreads(): returns 1 in exactly one workgroup, the last to arrive, and leaves the counters zeroed
// for the next launch that is handed this set.
// Ordering: the sums and the counters are touched by device-scope atomics ONLY (performed at the
// coherence point, read back with atomicExch), so no cache write-back is needed - a release fence at
// agent scope writes the XCD's L2 back once per workgroup and was measured at +14 us per launch.  What IS
// needed is that every sum atomic of this workgroup has been ACKNOWLEDGED before the ticket is taken:
// the other waves drained their vector-memory counters in front of the __syncthreads() the caller runs
// first (HIP's barrier waits for vmcnt(0)), and this thread drains its own here explicitly instead of
// relying on that code generation.
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int last_workgroup(int* __restrict__ set, int total, int lin) {
  drain_vmem();
  const int ngroups = total < BV_GROUPS ? total : BV_GROUPS;
  const int g = lin % BV_GROUPS;
  const int members = total / BV_GROUPS + (g < total % BV_GROUPS ? 1 : 0);
  int* t = set + BV_OFF;
  int last = 0;
  if (atomicAdd(t + 1 + g, 1) == members - 1) {
    atomicExch(t + 1 + g, 0);
    if (atomicAdd(t, 1) == ngroups - 1) { atomicExch(t, 0); last = 1; }
  }
  return last;
}

// The last workgroup's part: channels c0 .. c0+n of a layer with N channels; the sums (g12: N | N
// doubles) are read through the atomic unit and left zeroed.
__device__ __forceinline__ void bn_vec_finalize(const BnVecFin& f, int N, int c0, int n, double* __restrict__ g12,
                                                int tid, int nthreads) {
  for (int i = tid; i < n; i += nthreads) {
    const int c = c0 + i;
    const double g1 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(g12 + c), 0ull));
    const double g2 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(g12 + N + c), 0ull));
    bn_bwd_vectors_channel(c, N, f.count, g1, g2, f.gamma, f.ss, f.mi, f.vec, f.dgamma, f.dbeta);
  }
}

int* sched_slot();   // csrc/mlp.hip: the next self-resetting counter set of the ring

}  // namespace demf
