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

// a ^ b ^ c in ONE VALU instruction.  Left to itself the compiler emits two v_xor_b32 per three-way xor
// of a Philox round (34 of the 170 VALU instructions of a GENERATE path-step were xors).
__host__ __device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // v_bitop3_b32, truth table of a three-way xor
#else
  return a ^ b ^ c;
#endif
}

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                          uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
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

// The same value built without integer->double conversions: 1.mantissa in [1, 2), then one exact
// subtraction of (1 - 2^-53):  (1 + k 2^-52) - (1 - 2^-53) = (k + 1/2) 2^-52.
__device__ __forceinline__ double u01_fast(uint32_t lo, uint32_t hi) {
  const uint64_t w = ((uint64_t)hi << 32) | lo;
  return __longlong_as_double((long long)(0x3FF0000000000000ull | (w >> 12))) -
         0x1.fffffffffffffp-1;
}
// 2·u01_fast(lo, hi), exactly, in the same three instructions: the mantissa under the exponent of [2, 4),
// (2 + k 2^-51) - (2 - 2^-52) = (k + 1/2) 2^-51
__device__ __forceinline__ double two_u01_fast(uint32_t lo, uint32_t hi) {
  const uint64_t w = ((uint64_t)hi << 32) | lo;
  return __longlong_as_double((long long)(0x4000000000000000ull | (w >> 12))) -
         0x1.fffffffffffffp+0;
}

// ---- transcendentals of the Box–Muller transform, specialised to its argument ranges ------------
// The generic library routines cost 98 (log) + 71 (sincospi) + 22 (sqrt) VALU instructions per pair
// of normals — more than everything else in a GENERATE path-step together; their generality (full
// range, special values, denormals) is not needed here.  Accuracy of each: <= 2 ulp.

// One Horner step p·z + c as a three-operand v_fma_f64.  Written as asm because the compiler turns
// fma(p, z, CONSTANT) in a loop into "v_mov_b64 tmp, c ; v_fmac_f64 tmp, p, z" — it selects the
// two-address v_fmac form against the materialised constant, and after the constant is hoisted out
// of the loop a 64-bit copy per step remains: 17 extra VALU instructions per pair of normals (9 %
// of a GENERATE path-step).  Same operation, same rounding.
__device__ __forceinline__ double horner(double p, double z, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(z), "v"(c));
  return d;
}

// -2 ln(u), u in (0, 1):  u = 2^e m, m in [sqrt(1/2), sqrt(2));  ln m = 2 atanh(s), s = (m-1)/(m+1)
__device__ __forceinline__ double neg2_log_unit(double u) {
  double m = __builtin_amdgcn_frexp_mant(u);  // [1/2, 1)
  int e = __builtin_amdgcn_frexp_exp(u);
  const int lowm = m < 0x1.6a09e667f3bcdp-1;  // below sqrt(1/2): m <- 2m, e <- e - 1
  m = __builtin_amdgcn_ldexp(m, lowm);        // (one select + v_ldexp_f64; `lowm ? 2m : m` is a multiply + two selects)
  e -= lowm;
  const double f = m - 1.0, d = m + 1.0;  // f exact
  double r = __builtin_amdgcn_rcp(d);     // hardware reciprocal + two Newton steps
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  double s = f * r;
  s = fma(fma(-s, d, f), r, s);           // residual correction: s = f/d to < 1 ulp
  const double z = s * s;
  // the polynomial of ln m = 2s + s z p(z), p = 2/3 + 2/5 z + …, with every coefficient times -2: scaling by a
  // power of two commutes with every rounding, so -2·ln u comes out of the recombination directly — the same
  // bits as (-2.0)·(the unscaled sum), one multiply less
  double p = -0x1.af286bca1af28p-3;        // -2 · 2/19
  p = horner(p, z, -0x1.e1e1e1e1e1e1ep-3);    // -2 · 2/17
  p = horner(p, z, -0x1.1111111111111p-2);    // -2 · 2/15
  p = horner(p, z, -0x1.3b13b13b13b14p-2);    // -2 · 2/13
  p = horner(p, z, -0x1.745d1745d1746p-2);    // -2 · 2/11
  p = horner(p, z, -0x1.c71c71c71c71cp-2);    // -2 · 2/9
  p = horner(p, z, -0x1.2492492492492p-1);    // -2 · 2/7
  p = horner(p, z, -0x1.999999999999ap-1);    // -2 · 2/5
  p = horner(p, z, -0x1.5555555555555p+0);    // -2 · 2/3
  const double de = (double)e;
  // -2 ln u = e (-2 ln2_hi) + (-4 s + (s z p' + e (-2 ln2_lo))); ln2_hi has 21 trailing zero bits (exact product)
  const double t = fma(s * z, p, de * -3.81642985854117540004e-10);
  return fma(de, -1.38629436073824763298e+00, fma(-4.0, s, t));
}

// sqrt(a) for a normal, positive a: reciprocal-square-root seed + two coupled Newton steps
__device__ __forceinline__ double sqrt_pos(double a) {
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double rr = fma(-h, g, 0.5);
  g = fma(g, rr, g);
  h = fma(h, rr, h);
  rr = fma(-h, g, 0.5);
  g = fma(g, rr, g);
  h = fma(h, rr, h);
  return fma(fma(-g, g, a), h, g);
}

// (q & 2) ? -x : x as a shift and ONE v_bitop3_b32 on the high word — hi ^ ((q << 30) & 0x80000000), truth
// table 0xF0 ^ (0xCC & 0xAA) — where `and, compare, two v_cndmask` stood: the GENERATE kernel issues one VALU
// instruction per cycle, so the four instructions saved per pair of normals are 2.6 % of it.
// Flipping the sign bit IS negation: not a bit of any result changes.
__device__ __forceinline__ double negate_if_bit1(double x, int q) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(b >> 32), (uint32_t)q << 30, 0x80000000u, 0x78);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (uint32_t)b));
}

// sin(pi t), cos(pi t) for t in (0, 2): t = q/2 + r exactly, |r| <= 1/4; Taylor in r^2
__device__ __forceinline__ void sincospi_02(double t, double& sn, double& cs) {
  const double q = __builtin_rint(2.0 * t);
  const double r = fma(q, -0.5, t);
  const double z = r * r;
  double ps = -0x1.6fadb9f155744p-16;
  ps = horner(ps, z, 0x1.e8f434d018d63p-12);
  ps = horner(ps, z, -0x1.e3074fde8871fp-8);
  ps = horner(ps, z, 0x1.50783487ee782p-4);
  ps = horner(ps, z, -0x1.32d2cce62bd86p-1);
  ps = horner(ps, z, 0x1.466bc6775aae2p+1);
  ps = horner(ps, z, -0x1.4abbce625be53p+2);
  ps = horner(ps, z, 0x1.921fb54442d18p+1);
  const double sr = r * ps;
  double pc = 0x1.20c62c2f2d7f5p-18;
  pc = horner(pc, z, -0x1.b6e24f44b128fp-14);
  pc = horner(pc, z, 0x1.f9d38a3763cc3p-10);
  pc = horner(pc, z, -0x1.a6d1f2a204a8cp-6);
  pc = horner(pc, z, 0x1.e1f506891babbp-3);
  pc = horner(pc, z, -0x1.55d3c7e3cbffap+0);
  pc = horner(pc, z, 0x1.03c1f081b5ac4p+2);
  pc = horner(pc, z, -0x1.3bd3cc9be45dep+2);
  const double cr = fma(pc, z, 1.0);
  const int qi = (int)q;
  const bool swap = qi & 1;
  const double s0 = swap ? cr : sr, c0 = swap ? sr : cr;
  sn = negate_if_bit1(s0, qi);         // quadrants 2, 3
  cs = negate_if_bit1(c0, qi + 1);     // quadrants 1, 2
}

// Two independent N(0,1) from one Philox block (Box–Muller).
__device__ __forceinline__ void normal_pair(const Philox4& b, double& z1, double& z2) {
  const double u1 = u01_fast(b.c0, b.c1);
  const double t = two_u01_fast(b.c2, b.c3);  // angle / pi = 2 u2, in (0, 2)
  const double r = sqrt_pos(neg2_log_unit(u1));
  double s, c;
  sincospi_02(t, s, c);
  z1 = r * c;
  z2 = r * s;
}

__device__ __forceinline__ void normal_pair(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2,
                                            uint32_t dom, double& z1, double& z2) {
  normal_pair(philox4x32_10(c0, c1, c2, dom, (uint32_t)key, (uint32_t)(key >> 32)), z1, z2);
}

}  // namespace hh
