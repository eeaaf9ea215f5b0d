// Counter-based RNG for the Monte Carlo kernels: Philox4x32-10 (Salmon et al., SC'11) keyed by the
// trajectory seed, so a path's draws depend on (seed, counter) only — never on the wave, block,
// grid or GPU it lands on.  The reference draws from per-trajectory seeded generators
// (montecarlo.jl:331: remake(prob; seed = seeds[i])) and ONE Xoshiro(seeds[1]) stream for the exact
// laws (montecarlo.jl:456); the streams themselves live in third-party Julia packages and are not
// reproducible here — the law of the draws is what is kept.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hh {

// domain separators placed in counter word 3
constexpr uint32_t kDomEuler = 0u;
constexpr uint32_t kDomExactGbm = 1u;
constexpr uint32_t kDomBk = 2u;

struct Philox4 {
  uint32_t c0, c1, c2, c3;
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                          uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// 52 random bits + 1/2 ulp offset: u in [2^-53, 1 - 2^-53], exactly representable.
__host__ __device__ __forceinline__ double u01_from_bits(uint32_t lo, uint32_t hi) {
  const uint64_t w = ((uint64_t)hi << 32) | lo;
  return ((double)(w >> 12) + 0.5) * 0x1p-52;
}

// Two independent N(0,1) from one Philox block (Box–Muller).
__device__ __forceinline__ void normal_pair(const Philox4& b, double& z1, double& z2) {
  const double u1 = u01_from_bits(b.c0, b.c1);
  const double t = 2.0 * u01_from_bits(b.c2, b.c3);  // angle / pi, in (0, 2)
  const double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(t, &s, &c);
  z1 = r * c;
  z2 = r * s;
}

__device__ __forceinline__ void normal_pair(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2,
                                            uint32_t dom, double& z1, double& z2) {
  normal_pair(philox4x32_10(c0, c1, c2, dom, (uint32_t)key, (uint32_t)(key >> 32)), z1, z2);
}

}  // namespace hh
