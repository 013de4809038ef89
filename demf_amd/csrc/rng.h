// Counter-based dropout mask shared by the dense kernels (csrc/dense.hip) and the attention core
// (csrc/attn.hip): a mask element is a pure function of (seed, step, op, element index).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace demf {

// ---- counter-based dropout mask: stateless, reproducible in the backward -------------------------
// keep(seed, step, op, idx) ; murmur3-style finaliser over the four words.  The step counter lives
// on the device (rng[1]) and is advanced by demf_rng_advance once per training step, so a captured
// hipGraph draws a fresh mask at every replay.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool dropout_keep(const unsigned long long* __restrict__ rng, uint32_t op,
                                             unsigned long long idx, float p) {
  const unsigned long long seed = rng[0], step = rng[1];
  uint32_t h = mix32((uint32_t)seed ^ 0x9E3779B9u);
  h = mix32(h ^ (uint32_t)(seed >> 32));
  h = mix32(h ^ (uint32_t)step);
  h = mix32(h ^ (uint32_t)(step >> 32) ^ (op * 0x632BE5ABu));
  h = mix32(h ^ (uint32_t)idx);
  h = mix32(h ^ (uint32_t)(idx >> 32));
  // 24-bit uniform in [0,1)
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;
}

}  // namespace demf
