// Modified Bessel function I_ν(z), real order ν > -1, complex argument — the special function of the
// Broadie–Kaya characteristic function (heston.jl:184-212 calls besseli through SpecialFunctions.jl,
// i.e. AMOS, which is not reproducible here: restated from the published expansions, DLMF §10).
//
//   I_ν(z) = exp(lg) · mul    the part that can be huge or tiny stays a logarithm, the O(1) sum stays a
//                             factor: the CF exponentiates log I anyway, so log(sum) would be wasted
//
//   |z| <  13              ascending series (10.25.2) at order ν
//   |z| >= R_h(ν)          Hankel expansion (10.40.5) at order ν;  R_h = max(13, ν²/6 + 13): its terms behave
//                          like (ν²/2|z|)^k / k!, 31 of them reach 1e-19 from there
//   13 <= |z| < R_h        (orders ν >= 1 only) the ascending series again where its terms do not cancel by
//                          more than e^14: I_ν(|z|)/|I_ν(z)| <= min(exp(|z| - Re z), exp((Im z)²/(2(ν+1)))), so
//                          where |z| - Re z <= 14 or (Im z)² <= 28 (ν+1), and |z| < R_t(ν), the reach of the
//                          coefficient table (63 terms); arguments of the CF are mostly of this kind —
//   else                   Hankel expansion at the base order ν0 = ν - floor(ν), then the ratios
//                          I_{ν0+k+1}/I_{ν0+k} from the backward recurrence (minimal solution): |z| + ν + 30
//                          complex divisions, the expensive way
//
// Both sums are evaluated by Horner's rule with the number of terms fixed BEFORE the loop from |z|
// (tables made on the host from ν), instead of term by term with a convergence test: 6 VALU
// instructions per series term and 4 per Hankel term, against ≈ 25 and ≈ 16 — the CF kernel spends
// most of its time here.  Each sum is split into its even and odd half: two independent Horner
// chains for the same instruction count (a lone chain leaves the fp64 pipe waiting on itself).  The
// loop runs to the longest count among the lanes of the wave, so its counter is uniform and the
// coefficients come from the scalar cache, two steps per load, fetched one iteration ahead.
//
// Compiles for the host too (tests/c/bessel_check.cpp checks it against mpmath references).
#pragma once
#include "hh_math.h"

namespace hh {

struct cx {
  double re, im;
};
HH_MATH_FN cx operator+(cx a, cx b) { return {a.re + b.re, a.im + b.im}; }
// (the library is built with -ffp-contract=off — the Euler kernels reproduce the reference's arithmetic bit for
// bit — so the contraction is written out here: a complex product is 2 multiplies + 2 fma, not 4 + 2; some forty of the
// instructions of one characteristic-function evaluation)
HH_MATH_FN cx operator*(cx a, cx b) { return {fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)}; }
HH_MATH_FN cx operator*(double s, cx a) { return {s * a.re, s * a.im}; }
HH_MATH_FN cx cdiv(cx a, cx b) {
  const double inv = fm::rcp(fma(b.re, b.re, b.im * b.im));
  return {(a.re * b.re + a.im * b.im) * inv, (a.im * b.re - a.re * b.im) * inv};
}
// 1/b (cdiv({1, 0}, b) without the products by the constant 1 and 0 the compiler may not drop)
HH_MATH_FN cx crcp(cx b) {
  const double inv = fm::rcp(fma(b.re, b.re, b.im * b.im));
  return {b.re * inv, -(b.im * inv)};
}
// |a| without hypot's range scaling (31 instructions): the moduli taken here (γ, ν_γ, ϕ, series
// sums) are far from the overflow / underflow thresholds of a² + b²
HH_MATH_FN double cabs(cx a) { return fm::sqrt_lean(fma(a.re, a.re, a.im * a.im)); }
// sin, cos: the range-specialised pair of hh_math.h; beyond |x| = 2^20 (which no parameter set of
// the tests reaches) its three-term reduction, good to 2^45
HH_MATH_FN void sincos_cf(double x, double& s, double& c) {
  if (fabs(x) <= 0x1p20) {
    fm::sincos(x, s, c);
  } else {
    fm::sincos_wide(x, s, c);
  }
}
HH_MATH_FN cx cexp(cx z) {
  const double e = fm::exp(z.re);
  double s, c;
  sincos_cf(z.im, s, c);
  return {e * c, e * s};
}
// … for |Re z| < 1e9 (fm::exp_finite)
HH_MATH_FN cx cexp_finite(cx z) {
  const double e = fm::exp_finite(z.re);
  double s, c;
  sincos_cf(z.im, s, c);
  return {e * c, e * s};
}
HH_MATH_FN cx clog(cx z) { return {fm::log(cabs(z)), fm::atan2(z.im, z.re)}; }

constexpr int kHankelTerms = 32;   // a_0 … a_31
constexpr int kHankelPairs = 16;   // Horner steps over (a_2m, a_2m+1)
constexpr int kSeriesTerms = 64;   // size of the series table: at most 63 terms (29 at |z| = 13, ν < 1)
constexpr double kSeriesR = 13.0;  // least |z| at which the series hands over (orders below 6)
constexpr double kBesselPi = 3.14159265358979323846;
constexpr double kBesselTwoPi = 6.28318530717958647692;

// Everything that depends on the order alone (host-made, bessel_table()).
struct BesselTable {
  double nu, lgam;                   // order, lgamma(ν + 1)
  double series_rmax, hankel_from;   // R_t(ν), R_h(ν) above
  double series_im2;                 // 28 (ν + 1)
  double series_n0, series_n1;       // series length at |z| = r: n0 + n1·r (the first omitted term is then
                                     // below 2^-57 of the largest one, for every r < R_t at THIS ν)
  double hankel[kHankelTerms];       // a_k(ν)  (DLMF 10.17.1)
  double hankel_rmin[kHankelPairs];  // [M]: |z| from which the sum cut after k = 2M+1 has converged
  // ascending series split by parity, c_k = 1 / (k (k + ν)):  [2m] = c_{2m-1} c_{2m} (even half),
  // [2m+1] = c_{2m} c_{2m+1} (odd half), m >= 1
  double series_de[kSeriesTerms + 4];
  double series_c1;                  // c_1
};

HH_MATH_FN int series_terms(const BesselTable& t, double r) {
  const int n = (int)fma(t.series_n1, r, t.series_n0);
  return n < kSeriesTerms ? n : kSeriesTerms;
}

// host: fills the table for one order; returns false when no series length within kSeriesTerms
// meets the bound on [0, R_s) (not for ν > -1 up to a few hundred)
static inline bool bessel_table(double nu, BesselTable& t) {
  t.nu = nu;
  t.lgam = lgamma(nu + 1.0);
  const double rh = nu * nu / 6.0 + 13.0;
  t.hankel_from = rh > kSeriesR ? rh : kSeriesR;
  t.series_im2 = 28.0 * (nu + 1.0);
  const double mu = 4.0 * nu * nu;
  long double a[kHankelTerms + 2];
  a[0] = 1.0L;
  for (int k = 1; k < kHankelTerms + 2; ++k) {
    const long double o = 2.0L * k - 1.0L;
    a[k] = a[k - 1] * ((long double)mu - o * o) / (8.0L * k);
  }
  for (int k = 0; k < kHankelTerms; ++k) t.hankel[k] = (double)a[k];
  for (int M = 0; M < kHankelPairs; ++M) {
    // first omitted term |a_{2M+2}| / r^{2M+2} < 2^-55
    const int k = 2 * M + 2;
    const long double mag = fabsl(a[k]);
    t.hankel_rmin[M] = mag == 0.0L ? 0.0 : (double)powl(mag * 0x1p55L, 1.0L / k);
  }
  auto c = [nu](int k) { return 1.0L / ((long double)k * ((long double)k + (long double)nu)); };
  t.series_de[0] = t.series_de[1] = 0.0;
  for (int m = 1; 2 * m + 1 < kSeriesTerms + 4; ++m) {
    t.series_de[2 * m] = (double)(c(2 * m - 1) * c(2 * m));
    t.series_de[2 * m + 1] = (double)(c(2 * m) * c(2 * m + 1));
  }
  t.series_c1 = (double)c(1);
  // terms needed at r: the smallest N past the largest term with T_{N+1} < 2^-57 max_k T_k, for r up to
  // R_h or as far as the table reaches (R_t); then the line n0 + n1 r above all of them on (0, R_t]
  // with the least mean (the loop length a lane computes from its r)
  constexpr int kGrid = 8;
  int need[4096];
  int npts = 0;
  for (int i = 1; i <= 4096 && (double)(i - 1) / kGrid < t.hankel_from; ++i) {
    const long double r = (long double)i / kGrid, q = 0.25L * r * r;
    long double term = 1.0L, largest = 1.0L;
    int k_largest = 0, N = -1;
    for (int k = 1; k < 400; ++k) {
      term *= q / ((long double)k * (k + (long double)nu));
      if (term > largest) { largest = term; k_largest = k; }
      else if (k > k_largest && term < 0x1p-57L * largest) { N = k - 1; break; }
    }
    if (N < 0 || N + 3 > kSeriesTerms) break;
    need[npts++] = N;
  }
  if (npts < (int)(kSeriesR * kGrid)) return false;  // the table must reach |z| = 13
  t.series_rmax = (double)npts / kGrid;
  if (t.series_rmax > t.hankel_from) t.series_rmax = t.hankel_from;
  double best_n0 = 0.0, best_n1 = 0.0, best_sum = 1e300;
  for (int j = 0; j <= 30; ++j) {
    const double n1 = 0.1 * j;
    double n0 = 0.0;
    for (int i = 1; i <= npts; ++i) {
      const double v = need[i - 1] + 1.0 - n1 * ((double)(i - 1) / kGrid);  // holds on [r_{i-1}, r_i]
      if (v > n0) n0 = v;
    }
    const double sum = n0 + 0.5 * n1 * t.series_rmax;  // mean length over the range
    if (sum < best_sum && n0 + n1 * t.series_rmax + 1.0 <= (double)kSeriesTerms) {
      best_sum = sum; best_n0 = n0; best_n1 = n1;
    }
  }
  if (best_sum == 1e300) return false;
  t.series_n0 = best_n0 + 0.01;
  t.series_n1 = best_n1;
  return true;
}

// the largest n (0 … 63) among the active lanes of the wave
HH_MATH_FN int wave_max6(int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  int r = 0;
#pragma unroll
  for (int b = 5; b >= 0; --b) {
    const int c = r | (1 << b);
    if (__ballot(n >= c) != 0ull) r = c;
  }
  return r;
#else
  return n;
#endif
}

// A value that IS the same in every active lane, said so to the compiler: table indices derived from
// wave_max6() are uniform by construction, but inside the divergent loops of a grid-stride kernel the
// compiler cannot prove it — and then copies the by-value kernel-argument tables to scratch to index
// them per lane (1.9 KB per lane in bk_fallback_kernel).  readfirstlane keeps them scalar loads.
HH_MATH_FN int uniform_index(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(i);
#else
  return i;
#endif
}

struct LogMul {
  cx lg, mul;
};

// One Horner step of two independent complex chains with real coefficients: X = X·u + cx, Y = Y·u + cy
HH_MATH_FN void horner2(cx& X, cx& Y, cx u, double cx_, double cy_) {
  X = {fma(X.re, u.re, fma(-X.im, u.im, cx_)), fma(X.re, u.im, X.im * u.re)};
  Y = {fma(Y.re, u.re, fma(-Y.im, u.im, cy_)), fma(Y.re, u.im, Y.im * u.re)};
}

// I_ν(z) by the ascending series (DLMF 10.25.2), Re z >= 0, |z| = r < t.series_rmax (R_t), arg z = phi:
//   I = (z/2)^ν / Γ(ν+1) · Σ_k T_k,  T_k = q^k / (k! (ν+1)_k) = T_{k-1} c_k q,  q = z²/4.
// Even and odd terms apart, Q = q²:  Σ T_2m = 1 + d_1 Q (1 + d_2 Q (1 + …)),  d_m = c_{2m-1} c_{2m};
//                                    Σ T_2m+1 = c_1 q (1 + e_1 Q (1 + e_2 Q (1 + …))),  e_m = c_{2m} c_{2m+1}
// (r only sets the series' length — sqrt_rough is enough; the logarithm is taken of r2 = |z|² = r² itself:
// ν log(r/2) = (ν/2) log(r²/4), one rounding of r² instead of the two of a square root)
HH_MATH_FN LogMul besseli_series(const BesselTable& t, cx z, double r, double r2, double phi) {
  const cx q = 0.25 * (z * z);
  const cx Q = q * q;
  const int Mh = (series_terms(t, r) + 1) >> 1;  // steps of each half
  cx A = {1.0, 0.0}, B = {1.0, 0.0};
  int m = uniform_index(wave_max6(Mh));
  const double* tab = t.series_de;
  if (m & 1) {  // (uniform) an odd count: its first step alone, then pairs — not a pair with one step nobody needs
    const double dm = tab[2 * m], em = tab[2 * m + 1];
    if (m <= Mh) {
      const cx qa = Q * A, qb = Q * B;
      A = {fma(dm, qa.re, 1.0), dm * qa.im};
      B = {fma(em, qb.re, 1.0), em * qb.im};
    }
    m -= 1;
  }
  if (m >= 2) {  // (uniform) even: steps (m, m-1) down to (2, 1)
    double d0 = tab[2 * m], e0 = tab[2 * m + 1], d1 = tab[2 * m - 2], e1 = tab[2 * m - 1];
    while (true) {
      const double dm = d0, em = e0, dn = d1, en = e1;
      const int mn = uniform_index(m - 2);
      if (mn >= 2) {  // the next two steps' coefficients, one iteration ahead of their use
        d0 = tab[2 * mn]; e0 = tab[2 * mn + 1]; d1 = tab[2 * mn - 2]; e1 = tab[2 * mn - 1];
      }
      if (m <= Mh) {  // X = 1 + c Q X
        const cx qa = Q * A, qb = Q * B;
        A = {fma(dm, qa.re, 1.0), dm * qa.im};
        B = {fma(em, qb.re, 1.0), em * qb.im};
      }
      if (m - 1 <= Mh) {
        const cx qa = Q * A, qb = Q * B;
        A = {fma(dn, qa.re, 1.0), dn * qa.im};
        B = {fma(en, qb.re, 1.0), en * qb.im};
      }
      if (mn < 2) break;
      m = mn;
    }
  }
  const cx S = A + t.series_c1 * (q * B);
  return {{(0.5 * t.nu) * fm::log(0.25 * r2) - t.lgam, t.nu * phi}, S};
}

// I_ν(z) by the Hankel expansion (DLMF 10.40.5), Re z >= 0, |z| = r >= t.hankel_from:
//   I = e^z/sqrt(2πz) [S1 + e^{-2z ± iπ(ν+1/2)} S2],  S2 = Σ a_k w^k,  S1 = Σ (-1)^k a_k w^k,  w = 1/z,
// upper sign for Im z >= 0.  With E = Σ a_2m u^m, O = Σ a_2m+1 u^m, u = w²: S2 = E + wO, S1 = E - wO.
// The sums are cut after k = 2M+1 where the next term is below 2^-55 — or, for r < 15.5 where the
// expansion does not get that far, at its smallest term k ≈ 2r.
HH_MATH_FN LogMul besseli_asym(const BesselTable& t, cx z, double r, double phi) {
  const cx w = crcp(z);
  const cx u = w * w;
  int M = (int)(r - 0.5);
  M = M < kHankelPairs - 1 ? M : kHankelPairs - 1;
  for (int m = kHankelPairs - 2; m >= 3; --m) M = r >= t.hankel_rmin[m] ? m : M;  // (unrolled: constant bounds)
  cx E = {0.0, 0.0}, O = {0.0, 0.0};
  int m = uniform_index(wave_max6(M) | 1);  // odd: steps (m, m-1) down to (1, 0)
  const double* tab = t.hankel;
  double a0 = tab[2 * m], a1 = tab[2 * m + 1], b0 = tab[2 * m - 2], b1 = tab[2 * m - 1];
  while (true) {
    const double ae = a0, ao = a1, be = b0, bo = b1;
    const int mn = uniform_index(m - 2);
    if (mn >= 1) {  // the next two steps' coefficients, one iteration ahead of their use
      a0 = tab[2 * mn]; a1 = tab[2 * mn + 1]; b0 = tab[2 * mn - 2]; b1 = tab[2 * mn - 1];
    }
    if (m <= M) horner2(E, O, u, ae, ao);
    if (m - 1 <= M) horner2(E, O, u, be, bo);
    if (mn < 1) break;
    m = mn;
  }
  const cx wO = w * O;
  cx m1 = {E.re - wO.re, E.im - wO.im};
  if (z.re < 18.5) {  // else e^{-2 Re z} < 1e-16: the second sum is below rounding
    const double ph = (z.im >= 0.0 ? kBesselPi : -kBesselPi) * (t.nu + 0.5);
    const cx e2 = cexp({-2.0 * z.re, -2.0 * z.im + ph});
    m1 = m1 + e2 * cx{E.re + wO.re, E.im + wO.im};
  }
  return {{z.re - 0.5 * fm::log(kBesselTwoPi * r), z.im - 0.5 * phi}, m1};
}

// I_ν(z) = exp(lg)·mul for real ν > -1 and complex z != 0 with arg z = phi given by the caller
// (principal branch; Im lg is defined modulo 2π — callers exponentiate).  t = table of ν,
// t0 = table of ν0 = ν - n_int (the same table when n_int = 0).
HH_MATH_FN LogMul besseli_logmul(const BesselTable& t, const BesselTable& t0, int n_int, cx z, double phi) {
  double refl = 0.0;
  if (z.re < 0.0) {  // I_ν(w e^{±iπ}) = e^{±iπν} I_ν(w)  (DLMF 10.34.1)
    const double pi_s = z.im >= 0.0 ? kBesselPi : -kBesselPi;
    refl = pi_s * t.nu;
    phi -= pi_s;
    z = {-z.re, -z.im};
  }
  // |z|: to 2^-23 (one v_sqrt_f64) for the choice of the regime and the series' length; the Hankel expansion, which
  // takes log r and the cut of its sums from it, gets the correctly rounded root
  const double r2 = fma(z.re, z.re, z.im * z.im);
  const double r = fm::sqrt_rough(r2);
  LogMul res;
  // (one inlined copy of the series for its two regions — below R_s, and between R_s and R_t where it does not
  // cancel and the Hankel sum has not converged yet)
  const bool hankel = n_int == 0 || r >= t.hankel_from;
  if (r < kSeriesR || (!hankel && r < t.series_rmax && (r - z.re <= 14.0 || z.im * z.im <= t.series_im2))) {
    res = besseli_series(t, z, r, r2, phi);
  } else if (hankel) {
    res = besseli_asym(t, z, fm::sqrt_lean(r2), phi);
  } else {
    // base order ν0, then I_ν = I_ν0 · Π_{k<n} I_{ν0+k+1}/I_{ν0+k}; the ratios come from the backward
    // recurrence r_k = 1 / (2(ν0+k+1)/z + r_{k+1}), the minimal solution for Re z >= 0.  |r_k| < 1: the
    // product is taken 32 ratios at a time and folded into the logarithm (a product of 32 cannot leave
    // the fp64 range; one complex log per ratio was 115 instructions each).
    res = besseli_asym(t0, z, fm::sqrt_lean(r2), phi);
    const cx w = 2.0 * crcp(z);
    const int n = n_int;
    int N = n + (int)r + 30;
    if (N > 4000) N = 4000;
    cx rk = {0.0, 0.0}, prod = {1.0, 0.0};
    int in_prod = 0;
    for (int k = N - 1; k >= 0; --k) {
      const double o = t0.nu + (double)k + 1.0;
      rk = crcp({fma(o, w.re, rk.re), fma(o, w.im, rk.im)});
      if (k < n) {
        prod = prod * rk;
        if (++in_prod == 32) {
          res.lg = res.lg + clog(prod);
          prod = {1.0, 0.0};
          in_prod = 0;
        }
      }
    }
    if (n <= 16) {
      res.mul = res.mul * prod;
    } else if (in_prod > 0) {
      res.lg = res.lg + clog(prod);
    }
  }
  res.lg.im += refl;
  return res;
}

// ---- the same function on the positive real axis ---------------------------------------------------------------
// besseli_logmul() for z = (x, ±0), x > 0, phi = 0, written in real arithmetic: OPERATION BY OPERATION the real
// parts of the complex code above — a complex product whose imaginary inputs are zeros rounds its real part
// once, like the real product; fma(a, b, ±0) = a·b — so (lg, mul) here ARE (lg.re, mul.re) there, bit for bit,
// at a third of the instructions (two real Horner chains of 2 instead of 6 instructions per term, no atan2, no
// second sine).  The Broadie–Kaya set-up needs exactly that: log I_ν(ν_κ) and the characteristic function at 0
// (hh_bk.hip, cf_setup) are real evaluations that went through the complex code, a tenth of a trajectory's work.
// tests/c/bessel_check.cpp compares the two on the host; tools/bk_ab.py on the device (same sums).
struct LogMulRe {
  double lg, mul;
};

HH_MATH_FN LogMulRe besseli_series_re(const BesselTable& t, double x, double r, double r2) {
  const double q = 0.25 * (x * x);
  const double Q = q * q;
  const int Mh = (series_terms(t, r) + 1) >> 1;
  double A = 1.0, B = 1.0;
  int m = uniform_index(wave_max6(Mh));
  const double* tab = t.series_de;
  for (; m >= 1; --m) {  // (uniform) X = 1 + c Q X
    const double dm = tab[2 * m], em = tab[2 * m + 1];
    if (m <= Mh) {
      const double qa = Q * A, qb = Q * B;
      A = fma(dm, qa, 1.0);
      B = fma(em, qb, 1.0);
    }
  }
  const double S = A + t.series_c1 * (q * B);
  return {(0.5 * t.nu) * fm::log(0.25 * r2) - t.lgam, S};
}

HH_MATH_FN LogMulRe besseli_asym_re(const BesselTable& t, double x, double r) {
  const double w = x * fm::rcp(x * x);  // crcp((x, 0)).re
  const double u = w * w;
  int M = (int)(r - 0.5);
  M = M < kHankelPairs - 1 ? M : kHankelPairs - 1;
  for (int m = kHankelPairs - 2; m >= 3; --m) M = r >= t.hankel_rmin[m] ? m : M;
  double E = 0.0, O = 0.0;
  const double* tab = t.hankel;
  for (int m = uniform_index(wave_max6(M) | 1); m >= 0; --m) {  // (uniform)
    const double ae = tab[2 * m], ao = tab[2 * m + 1];
    if (m <= M) {
      E = fma(E, u, ae);
      O = fma(O, u, ao);
    }
  }
  const double wO = w * O;
  double m1 = E - wO;
  if (x < 18.5) {
    const double ph = kBesselPi * (t.nu + 0.5);
    const double e = fm::exp(-2.0 * x);
    double sn, cs;
    sincos_cf(ph, sn, cs);
    m1 = m1 + (e * cs) * (E + wO);
  }
  return {x - 0.5 * fm::log(kBesselTwoPi * r), m1};
}

// x = |x|: the complex code reflects an argument left of the imaginary axis, and on the real axis the phases it adds
// for that cancel exactly
HH_MATH_FN LogMulRe besseli_logmul_re(const BesselTable& t, const BesselTable& t0, int n_int, double x) {
  x = fabs(x);
  const double r2 = x * x;
  const double r = fm::sqrt_rough(r2);
  const bool hankel = n_int == 0 || r >= t.hankel_from;
  if (r < kSeriesR || (!hankel && r < t.series_rmax)) return besseli_series_re(t, x, r, r2);
  if (hankel) return besseli_asym_re(t, x, fm::sqrt_lean(r2));
  LogMulRe res = besseli_asym_re(t0, x, fm::sqrt_lean(r2));
  const double w = 2.0 * (x * fm::rcp(x * x));
  const int n = n_int;
  int N = n + (int)r + 30;
  if (N > 4000) N = 4000;
  double rk = 0.0, prod = 1.0;
  int in_prod = 0;
  for (int k = N - 1; k >= 0; --k) {
    const double o = t0.nu + (double)k + 1.0;
    const double b = fma(o, w, rk);
    rk = b * fm::rcp(b * b);
    if (k < n) {
      prod = prod * rk;
      if (++in_prod == 32) {
        res.lg = res.lg + fm::log(fm::sqrt_lean(prod * prod));
        prod = 1.0;
        in_prod = 0;
      }
    }
  }
  if (n <= 16) {
    res.mul = res.mul * prod;
  } else if (in_prod > 0) {
    res.lg = res.lg + fm::log(fm::sqrt_lean(prod * prod));
  }
  return res;
}

}  // namespace hh
