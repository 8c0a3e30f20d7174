// Device code of the U[0,1) generator (noise.hip), shared with step_prologue.hip, where the
// same draws ride in the training step's prologue launch.
#pragma once
#include "common.h"

namespace scae_noise {
constexpr int NT = 256;

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  c[0] = hi1 ^ c[1] ^ k0;
  c[1] = lo1;
  c[2] = hi0 ^ c[3] ^ k1;
  c[3] = lo0;
}

// state[0] = seed, state[1] = launches so far, low word of state[2] = arrival count.
// The workgroup that arrives last advances state[1].  The arrival counter only
// orders "every workgroup has READ state[1]" before that write -- no data is
// handed between workgroups, so relaxed device-scope atomics suffice (a fenced
// "last workgroup" protocol costs ~35 ns per workgroup on the 8-XCD part; a
// second launch would cost ~4.5 us inside a graph).
// Block `blk` of `nblocks` (256 threads each) of one draw of n floats.
__device__ __forceinline__ void uniform_block(float *__restrict__ out, int64_t n,
                                              uint64_t *__restrict__ state, int blk,
                                              int nblocks) {
  const uint64_t seed = state[0], launch = state[1];
  const int64_t groups = (n + 3) / 4;
  for (int64_t g = (int64_t)blk * NT + threadIdx.x; g < groups; g += (int64_t)nblocks * NT) {
    uint32_t c[4] = {(uint32_t)g, (uint32_t)((uint64_t)g >> 32), (uint32_t)launch,
                     (uint32_t)(launch >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)  // 24 random bits -> [0, 1)
      if (4 * g + e < n) out[4 * g + e] = (float)(c[e] >> 8) * (1.0f / 16777216.0f);
  }
  __syncthreads();  // every wave of this workgroup holds its copy of state[1]
  if (threadIdx.x == 0) {
    unsigned *count = reinterpret_cast<unsigned *>(state + 2);
    // (launch & 0) keeps the atomic behind the load of state[1] in issue order
    const unsigned t = __hip_atomic_fetch_add(count, 1u + (unsigned)(launch & 0), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)nblocks - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      state[1] = launch + 1;
    }
  }
}

inline int blocks_for(int64_t n) {
  const int64_t b = ((n + 3) / 4 + NT - 1) / NT;
  return (int)(b < 2048 ? b : 2048);
}
}  // namespace scae_noise
