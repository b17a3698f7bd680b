// Philox4x32-10, the counter-based generator behind rcgan_rng_fill (elementwise.hip) and the step-input rider (step_inputs.h):
// quad q of a stream = philox4(offset + q, seed); uniform numbers take the top 24 bits of a word.
#pragma once
#include "common.h"

__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
  uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
  uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ void philox4(uint64_t ctr, uint32_t seed_lo, uint32_t seed_hi, uint32_t out[4]) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x5eed5eedu, c3 = 0;
  uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float philox_uniform(uint32_t r, float lo, float hi) { return lo + (hi - lo) * ((float)(r >> 8) * (1.0f / 16777216.0f)); }
