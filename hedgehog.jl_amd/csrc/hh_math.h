// Elementary functions for the Broadie–Kaya characteristic-function arithmetic, specialised to the
// argument ranges that occur there.  The device library's sincos / log / atan2 carry full-range
// argument reduction, denormal scaling and special-value handling: 155 / 98 / 105 VALU instructions
// (static count), against ≈ 40 each here.  Accuracy of every routine: ≤ 2 ulp on its stated range
// (checked on the host against libm by tests/test_math_host.py, which compiles this header with
// g++; on the device the hardware reciprocal replaces the division of the host build).
//
// Published sources of the approximations: Cody & Waite two-term reduction by π/2; kernel
// polynomials for sin / cos on [-π/4, π/4] and the 11-term odd polynomial with break points
// 7/16, 11/16, 19/16, 39/16 for atan: K. C. Ng's algorithms as distributed in FDLIBM (Sun
// Microsystems, 1993; "permission to use, copy, modify, and distribute this software is freely
// granted"); log via ln m = 2 atanh((m-1)/(m+1)) as in hh_rng.h.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HH_MATH_FN __device__ __forceinline__
#else
#define HH_MATH_FN static inline
#endif

namespace hh {
namespace fm {

// 1/x to <= 1 ulp: hardware reciprocal + two Newton steps (5 instructions; an IEEE division is 12)
HH_MATH_FN double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
#else
  return 1.0 / x;
#endif
}

// sqrt(w) for w >= 0 that is zero or at least 2^-767 (the moduli and half-sums of the characteristic-function
// arithmetic: |γ|, |ν_γ|, |ϕ|): the library routine's own core sequence — v_rsq_f64, one coupled Newton step,
// two residual corrections: the same correctly rounded result — without its 2^±256 range scaling and class
// test (18 instructions -> 11; four square roots per CF evaluation).  Zero: rsq(0) = +inf is capped at 2^1000
// and the whole sequence is then exact zeros (hh_kernels.hip, sqrt_clipped).  Below 2^-767 the result is merely
// less accurate, never NaN.
// CONTRACT: w finite and >= 0 — what the callers pass (cabs: a sum of squares; csqrt: r + |re| of non-negative
// terms).  Outside it the IEEE edge behaviour of sqrt() is NOT kept: w = +inf (an overflowed sum of squares)
// gives NaN (rsq = 0, 0·inf), where sqrt gives inf — the Broadie–Kaya stopping test leaves on NaN as it leaves on
// inf; w = NaN gives NaN; w = -0.0 gives NaN (sqrt: -0.0; a sum of squares is never -0.0); w < 0 gives -inf or NaN
// (the seed NaN is replaced by the cap), never a finite positive number.  Pinned on the device by tests/test_gpu_math_device.py.
HH_MATH_FN double sqrt_lean(double w, double* half_rsqrt = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_fmin(__builtin_amdgcn_rsq(w), 0x1p1000);
  double g = w * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  if (half_rsqrt) *half_rsqrt = h;  // ≈ 1/(2 sqrt(w)) after the coupled step (relative error ~2^-45): see div_by_2sqrt
  g = fma(fma(-g, g, w), h, g);
  return fma(fma(-g, g, w), h, g);
#else
  if (half_rsqrt) *half_rsqrt = 0.5 / sqrt(w);
  return sqrt(w);
#endif
}

// a / (2t) for t = sqrt_lean(w, &h): the quotient from the square root's own h ≈ 1/(2t) and one residual
// correction (4 instructions, <= 1 ulp: the correction term is ~2^-45 of the quotient and carries h's 2^-45) —
// instead of a reciprocal of 2t (v_rcp_f64 + two Newton steps + two products: 8)
HH_MATH_FN double div_by_2sqrt(double a, double t, double h) {
  const double v = a * h;
  return fma(fma(-(t + t), v, a), h, v);
}

// sqrt(w) to ~2^-23 relative, ONE instruction (v_sqrt_f64): for quantities that only choose a regime or a loop
// length
HH_MATH_FN double sqrt_rough(double w) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrt(w);
#else
  return sqrt(w);
#endif
}

// p·z + c with c a literal: a three-operand v_fma_f64 reading the constant from an SGPR pair.  Left
// to itself the compiler materialises the literal in VGPRs and uses the two-address v_fmac form
// (2-3 VALU instructions per Horner step instead of 1; the scalar moves issue beside the VALU).
HH_MATH_FN double fma_c(double p, double z, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(z), "s"(c));
  return d;
#else
  return fma(p, z, c);
#endif
}

// (q & 2) ? -x : x — on the device a shift and one v_bitop3_b32 on the high word (hi ^ ((q << 30) & 0x80000000))
// instead of and / compare / two v_cndmask; flipping the sign bit is negation, bit for bit
HH_MATH_FN double negate_if_bit1(double x, int q) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(b >> 32), (uint32_t)q << 30, 0x80000000u, 0x78);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (uint32_t)b));
#else
  return (q & 2) ? -x : x;
#endif
}

// sin r, cos r for |r| <= π/4 (+ rounding of the reduction), rotated into quadrant q (mod 4)
HH_MATH_FN void sincos_reduced(double r, int q, double& sn, double& cs) {
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = fma_c(ps, z, -2.50507602534068634195e-08);
  ps = fma_c(ps, z, 2.75573137070700676789e-06);
  ps = fma_c(ps, z, -1.98412698298579493134e-04);
  ps = fma_c(ps, z, 8.33333333332248946124e-03);
  ps = fma_c(ps, z, -1.66666666666666324348e-01);
  const double sr = fma(r * z, ps, r);
  double pc = -1.13596475577881948265e-11;
  pc = fma_c(pc, z, 2.08757232129817482790e-09);
  pc = fma_c(pc, z, -2.75573143513906633035e-07);
  pc = fma_c(pc, z, 2.48015872894767294178e-05);
  pc = fma_c(pc, z, -1.38888888888741095749e-03);
  pc = fma_c(pc, z, 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * (z * pc));
  const bool swap = q & 1;
  const double s0 = swap ? cr : sr, c0 = swap ? sr : cr;
  sn = negate_if_bit1(s0, q);
  cs = negate_if_bit1(c0, q + 1);
}

// sin and cos of x, |x| <= 2^20 (error of the two-term reduction: |n|·2e-33)
HH_MATH_FN void sincos(double x, double& sn, double& cs) {
  const double n = rint(x * 6.36619772367581382433e-01);  // 2/π
  double r = fma(n, -1.57079632679489655800e+00, x);      // exact cancellation
  r = fma(n, -6.12323399573676603587e-17, r);
  sincos_reduced(r, (int)n, sn, cs);
}

// The same for |x| <= 2^45: three-term reduction (π/2 = c1 + c2 + c3 to 2^-161; each fma rounds the
// exact difference once, so the reduced argument carries |n|·5.6e-50 + 2^-53 of absolute error) and
// the quadrant taken from n mod 4 in floating point.  Replaces the device library's full-range
// sincos (Payne–Hanek: ~150 instructions and some 30 more live registers in the Broadie–Kaya
// kernel) for the arguments beyond 2^20 that only extreme parameter sets produce; beyond 2^45 an
// fp64 angle no longer determines its sine (spacing of the arguments > 2π·1e-3).
HH_MATH_FN void sincos_wide(double x, double& sn, double& cs) {
  const double n = rint(x * 6.36619772367581382433e-01);
  double r = fma(n, -1.57079632679489655800e+00, x);
  r = fma(n, -6.12323399573676603587e-17, r);
  r = fma(n, 1.49738490485916983294e-33, r);
  const double nq = n - 4.0 * rint(n * 0.25);  // n mod 4 in [-2, 2]
  sincos_reduced(r, (int)nq, sn, cs);
}

// e^x: x = k ln2 + r, |r| <= ln2 / 2; e^r = 1 + r + r² q(r), q the degree-11 Taylor polynomial
// (remainder r^14/14! < 6e-18); 2^k by v_ldexp_f64, which also delivers the overflow to +inf and
// the gradual underflow to 0.  <= 1.5 ulp.  The device library's exp carries a table-free
// double-double kernel: about three times the instructions and registers.
HH_MATH_FN double exp(double x) {
  const double xc = fmin(fmax(x, -1100.0), 1100.0);  // keeps k inside int; saturates anyway
  const double k = rint(xc * 1.44269504088896338700e+00);
  double r = fma(k, -6.93147180369123816490e-01, xc);  // ln2 high part (trailing zeros: k·hi exact)
  r = fma(k, -1.90821492927058770002e-10, r);
  double q = 1.6059043836821613e-10;            // 1/13!
  q = fma_c(q, r, 2.08767569878681e-09);       // 1/12!
  q = fma_c(q, r, 2.505210838544172e-08);
  q = fma_c(q, r, 2.755731922398589e-07);
  q = fma_c(q, r, 2.7557319223985893e-06);
  q = fma_c(q, r, 2.48015873015873e-05);
  q = fma_c(q, r, 1.984126984126984e-04);
  q = fma_c(q, r, 1.388888888888889e-03);
  q = fma_c(q, r, 8.333333333333333e-03);
  q = fma_c(q, r, 4.1666666666666664e-02);
  q = fma_c(q, r, 1.6666666666666666e-01);
  q = fma_c(q, r, 0.5);
  const double er = 1.0 + fma(r * r, q, r);
  const double y = ldexp(er, (int)k);
  return x != x ? x : y;
}

// e^x for |x| < 1e9 or NaN (the CF's e^{−γT/2}: Re γ >= 0, a product of finite model constants and T): exp()
// without its clamp and its NaN select — k = rint(x log2 e) fits an int without being clamped, v_ldexp_f64
// delivers the underflow to 0 and the overflow to inf, and a NaN goes through the arithmetic by itself.  The same
// bits as exp() on that range; five instructions fewer.
HH_MATH_FN double exp_finite(double x) {
  const double k = rint(x * 1.44269504088896338700e+00);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double q = 1.6059043836821613e-10;
  q = fma_c(q, r, 2.08767569878681e-09);
  q = fma_c(q, r, 2.505210838544172e-08);
  q = fma_c(q, r, 2.755731922398589e-07);
  q = fma_c(q, r, 2.7557319223985893e-06);
  q = fma_c(q, r, 2.48015873015873e-05);
  q = fma_c(q, r, 1.984126984126984e-04);
  q = fma_c(q, r, 1.388888888888889e-03);
  q = fma_c(q, r, 8.333333333333333e-03);
  q = fma_c(q, r, 4.1666666666666664e-02);
  q = fma_c(q, r, 1.6666666666666666e-01);
  q = fma_c(q, r, 0.5);
  const double er = 1.0 + fma(r * r, q, r);
  return ldexp(er, (int)k);
}

// ln x for positive, normal x
HH_MATH_FN double log(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double mant = __builtin_amdgcn_frexp_mant(x);  // [1/2, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
#else
  int e;
  double mant = frexp(x, &e);
#endif
  const int low = mant < 0x1.6a09e667f3bcdp-1;  // below sqrt(1/2): mant <- 2 mant, e <- e - 1
#if defined(__HIP_DEVICE_COMPILE__)
  mant = __builtin_amdgcn_ldexp(mant, low);      // one select + v_ldexp_f64 (a multiply + two selects otherwise)
#else
  mant = low ? 2.0 * mant : mant;
#endif
  e -= low;
  const double f = mant - 1.0, d = mant + 1.0;
  const double r = rcp(d);
  double s = f * r;
  s = fma(fma(-s, d, f), r, s);  // s = f/d to < 1 ulp
  const double z = s * s;
  double p = 0x1.af286bca1af28p-4;        // 2/19
  p = fma_c(p, z, 0x1.e1e1e1e1e1e1ep-4);    // 2/17
  p = fma_c(p, z, 0x1.1111111111111p-3);    // 2/15
  p = fma_c(p, z, 0x1.3b13b13b13b14p-3);    // 2/13
  p = fma_c(p, z, 0x1.745d1745d1746p-3);    // 2/11
  p = fma_c(p, z, 0x1.c71c71c71c71cp-3);    // 2/9
  p = fma_c(p, z, 0x1.2492492492492p-2);    // 2/7
  p = fma_c(p, z, 0x1.999999999999ap-2);    // 2/5
  p = fma_c(p, z, 0x1.5555555555555p-1);    // 2/3
  const double de = (double)e;
  const double t = fma(s * z, p, de * 1.90821492927058770002e-10);  // + e·ln2_lo
  return fma(de, 6.93147180369123816490e-01, fma(2.0, s, t));       // e·ln2_hi exact
}

// atan2(y, x) for finite arguments, not both zero
// The angle of (|x|, |y|) in the first quadrant is H + atan(n/d) with ONE division and one break point: with
// m = min, M = max of |x|, |y| and t = m/M in [0, 1],
//   t <  7/16:  |y| <= |x|: atan(|y|/|x|)                       (H = 0,   n = |y|,  d = |x|)
//               |y| >  |x|: π/2 - atan(|x|/|y|)                 (H = π/2, n = -|x|, d = |y|)
//   t >= 7/16:  π/4 + atan((|y| - |x|)/(|y| + |x|)) either way  (H = π/4; |n/d| <= 9/23 < 7/16)
// — the kernel polynomial's own range, so FDLIBM's middle break point (c = 1/2) and the second quotient
// (t - c)/(1 + c t) of a reduction from t are not needed: 84 -> 55 VALU instructions (static count), one
// v_rcp_f64 instead of two, min / max instead of four selects, H from a small integer (its low word is zero:
// one select level per word) times π/4 split in two.
HH_MATH_FN double atan2(double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
  const double mx = fmax(ax, ay), mn = fmin(ax, ay);
  const bool inv = ay > ax;
  const bool b = mn >= 0.4375 * mx;
  const double n = b ? ay - ax : (inv ? -ax : ay);
  const double d = b ? ay + ax : mx;
  const double k = b ? 1.0 : (inv ? 2.0 : 0.0);
  const double hi = k * 7.85398163397448278999e-01, lo = k * 3.06161699786838301793e-17;  // exact: k = 0, 1, 2
  const double u = n * rcp(d);
  const double z = u * u, w = z * z;
  double s1 = 1.62858201153657823623e-02;
  s1 = fma_c(s1, w, 4.97687799461593236017e-02);
  s1 = fma_c(s1, w, 6.66107313738753120669e-02);
  s1 = fma_c(s1, w, 9.09088713343650656196e-02);
  s1 = fma_c(s1, w, 1.42857142725034663711e-01);
  s1 = fma_c(s1, w, 3.33333333333329318027e-01);
  s1 *= z;
  double s2 = -3.65315727442169155270e-02;
  s2 = fma_c(s2, w, -5.83357013379057348645e-02);
  s2 = fma_c(s2, w, -7.69187620504482999495e-02);
  s2 = fma_c(s2, w, -1.11111104054623557880e-01);
  s2 = fma_c(s2, w, -1.99999999998764832476e-01);
  s2 *= w;
  double a = hi - ((u * (s1 + s2) - lo) - u);    // the angle of (|x|, |y|), in [0, π/2]
  if (x < 0.0) a = 3.14159265358979311600e+00 - (a - 1.22464679914735317723e-16);
  return y < 0.0 ? -a : a;
}

// The standard normal quantile Φ⁻¹(p), 0 < p < 1: Wichura's AS 241 (Appl. Statist. 37 (1988) 477-484), routine
// PPND16 — rational approximations of degree 7/7 in three regions, relative error below 1e-16 before rounding
// (measured against an 80-bit Newton refinement: tests/c/math_check.cpp).  The device library's normcdfinv goes
// through erfcinv: 1074 instructions, most of whose branches a wave with lanes in the body and in the tails runs
// one after the other; this is 25 in the body (|p − 1/2| <= 0.425), 75 more for a wave that has a lane in the tails,
// and a third region (p < 1.4e-11) hardly any wave enters.  The Broadie–Kaya draw kernel took the quantile of each
// trajectory's uniform (sample_from_cf.jl:33) from the library: a third of that kernel.  p = 0 / 1 give -inf / +inf.
HH_MATH_FN double normal_quantile(double p) {
  const double q = p - 0.5;
  double val;
  if (fabs(q) <= 0.425) {
    const double r = 0.180625 - q * q;
    double n = 2.5090809287301226727e+3, d = 5.2264952788528545610e+3;
    n = fma_c(n, r, 3.3430575583588128105e+4); d = fma_c(d, r, 2.8729085735721942674e+4);
    n = fma_c(n, r, 6.7265770927008700853e+4); d = fma_c(d, r, 3.9307895800092710610e+4);
    n = fma_c(n, r, 4.5921953931549871457e+4); d = fma_c(d, r, 2.1213794301586595867e+4);
    n = fma_c(n, r, 1.3731693765509461125e+4); d = fma_c(d, r, 5.3941960214247511077e+3);
    n = fma_c(n, r, 1.9715909503065514427e+3); d = fma_c(d, r, 6.8718700749205790830e+2);
    n = fma_c(n, r, 1.3314166789178437745e+2); d = fma_c(d, r, 4.2313330701600911252e+1);
    n = fma_c(n, r, 3.3871328727963666080e+0); d = fma_c(d, r, 1.0);
    return q * n * rcp(d);
  }
  const double tail = q < 0.0 ? p : 1.0 - p;
  const double r = sqrt_lean(-log(tail));  // tail < 0.075: -log > 2.59
  if (r <= 5.0) {
    const double x = r - 1.6;
    double n = 7.74545014278341407640e-4, d = 1.05075007164441684324e-9;
    n = fma_c(n, x, 2.27238449892691845833e-2); d = fma_c(d, x, 5.47593808499534494600e-4);
    n = fma_c(n, x, 2.41780725177450611770e-1); d = fma_c(d, x, 1.51986665636164571966e-2);
    n = fma_c(n, x, 1.27045825245236838258e+0); d = fma_c(d, x, 1.48103976427480074590e-1);
    n = fma_c(n, x, 3.64784832476320460504e+0); d = fma_c(d, x, 6.89767334985100004550e-1);
    n = fma_c(n, x, 5.76949722146069140550e+0); d = fma_c(d, x, 1.67638483018380384940e+0);
    n = fma_c(n, x, 4.63033784615654529590e+0); d = fma_c(d, x, 2.05319162663775882187e+0);
    n = fma_c(n, x, 1.42343711074968357734e+0); d = fma_c(d, x, 1.0);
    val = n * rcp(d);
  } else {
    const double x = r - 5.0;
    double n = 2.01033439929228813265e-7, d = 2.04426310338993978564e-15;
    n = fma_c(n, x, 2.71155556874348757815e-5); d = fma_c(d, x, 1.42151175831644588870e-7);
    n = fma_c(n, x, 1.24266094738807843860e-3); d = fma_c(d, x, 1.84631831751005468180e-5);
    n = fma_c(n, x, 2.65321895265761230930e-2); d = fma_c(d, x, 7.86869131145613259100e-4);
    n = fma_c(n, x, 2.96560571828504891230e-1); d = fma_c(d, x, 1.48753612908506148525e-2);
    n = fma_c(n, x, 1.78482653991729133580e+0); d = fma_c(d, x, 1.36929880922735805310e-1);
    n = fma_c(n, x, 5.46378491116411436990e+0); d = fma_c(d, x, 5.99832206555887937690e-1);
    n = fma_c(n, x, 6.65790464350110377720e+0); d = fma_c(d, x, 1.0);
    val = n * rcp(d);
  }
  if (!(tail > 0.0)) val = tail == 0.0 ? HUGE_VAL : tail;  // p = 0 or 1: ±inf; NaN stays NaN
  return q < 0.0 ? -val : val;
}

}  // namespace fm
}  // namespace hh
