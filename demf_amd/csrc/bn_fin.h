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
  double* racc;                 // replicated accumulator block (accum_slot()) or null: the sums go to g12 itself
};

// Replicated accumulators.  An atomic add costs ~20 ns at its memory channel and adds to one cache line queue up there,
// so W workgroups that end by adding their partial sums to the same 2N doubles finish W * 20 ns after the first
// (tools/ubench/atomic_flush.cpp: 1024 workgroups x 384 doubles = 23 us, 256 = 7 us).  A launch whose last workgroup
// consumes the sums anyway (ticket != null) adds to ACC_REPL copies instead - workgroup w to copy w % ACC_REPL, i.e.
// roughly one copy per XCD - and the last workgroup folds the copies while it clears them.
constexpr int ACC_REPL = 8;
constexpr int ACC_MAX_N = 1024;                          // channels per layer a block of the ring holds (2N doubles per copy)
constexpr int ACC_SLOT_DOUBLES = ACC_REPL * 2 * ACC_MAX_N;
double* accum_slot();   // csrc/mlp.hip: the next zeroed block of the ring (left zeroed by its consumer), or null (A/B off)

// this workgroup's copy: [sum | second sum] of N doubles each
__device__ __forceinline__ double* repl_copy(double* __restrict__ racc, int N, int lin) {
  return racc + (size_t)(lin % ACC_REPL) * 2 * N;
}
// total of element i (0 .. 2N) over the copies, each left zeroed
__device__ __forceinline__ double repl_take(double* __restrict__ racc, int N, int i) {
  unsigned long long v[ACC_REPL];
#pragma unroll
  for (int r = 0; r < ACC_REPL; ++r)
    v[r] = atomicExch(reinterpret_cast<unsigned long long*>(racc + (size_t)r * 2 * N + i), 0ull);
  double t = 0.0;
#pragma unroll
  for (int r = 0; r < ACC_REPL; ++r) t += __builtin_bit_cast(double, v[r]);
  return t;
}

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
    double g1 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(g12 + c), 0ull));
    double g2 = __builtin_bit_cast(double, atomicExch(reinterpret_cast<unsigned long long*>(g12 + N + c), 0ull));
    if (f.racc != nullptr) {      // (+ the copies: a producer that knows nothing of them leaves them zero)
      g1 += repl_take(f.racc, N, c);
      g2 += repl_take(f.racc, N, N + c);
    }
    bn_bwd_vectors_channel(c, N, f.count, g1, g2, f.gamma, f.ss, f.mi, f.vec, f.dgamma, f.dbeta);
  }
}

int* sched_slot();   // csrc/mlp.hip: the next self-resetting counter set of the ring

// Train-mode BatchNorm bookkeeping of a forward launch (demf_bn_finalize's arguments).  With
// ss != null the LAST workgroup of a STATS launch turns the column sums into scale / shift, saved
// mean / invstd and the running statistics itself and leaves the sums zeroed - the separate ~5 us
// finalize launch behind every forward GEMM disappears.
constexpr int FIN_OFF = 40;    // ints [FIN_OFF, FIN_OFF + 17) of a counter set: exit counts of the finalize
struct BnFin {
  double count;
  const float *gamma, *beta, *conv_bias;
  float eps, momentum;
  float *rmean, *rvar;
  long long* nbt;
  float *ss, *mi;
  int* ticket;                 // counter set (sched_slot) or null
  double* racc;                // replicated accumulator block (accum_slot, csrc/bn_fin.h) or null
};

__device__ __forceinline__ void bn_finalize_channel(int c, int N, double count, double s1, double s2,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps,
                                                    float momentum, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var,
                                                    float* __restrict__ ss, float* __restrict__ mi,
                                                    const float* __restrict__ conv_bias) {
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;            // biased, as BN normalises with
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  ss[c] = (float)((double)gamma[c] * invstd);                                   // scale
  ss[N + c] = (float)((double)beta[c] - mean * (double)gamma[c] * invstd);      // shift
  mi[c] = (float)mean;
  mi[N + c] = (float)invstd;
  if (running_mean != nullptr) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    // a conv bias in front of a train-mode BN cancels in the output; it only shifts the batch mean
    const double bm = mean + (conv_bias != nullptr ? (double)conv_bias[c] : 0.0);
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * bm);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}


}  // namespace demf
