// Host-side accuracy check of hedgehog.jl_amd/csrc/hh_math.h against 80-bit libm.
// Prints one line per function: name, samples, max error in ulp of the fp64 result.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>

#include "hh_math.h"

static double ulp_err(double got, long double want) {
  if (want == 0.0L) return got == 0.0 ? 0.0 : 1e9;
  int e;
  std::frexp((double)want, &e);
  const long double ulp = std::ldexp(1.0L, e - 53);
  return (double)(fabsl((long double)got - want) / ulp);
}

int main() {
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const int N = 2000000;
  double e_sin = 0, e_cos = 0, e_log = 0, e_at = 0, e_exp = 0, e_wsin = 0, e_wcos = 0;
  for (int i = 0; i < N; ++i) {
    // angles: dense near 0, up to +-2^20 on a log scale
    const double mag = std::exp2(-30.0 + 50.0 * U(rng));
    const double x = (U(rng) < 0.5 ? -mag : mag);
    double s, c;
    hh::fm::sincos(x, s, c);
    // relative to max(|value|, 2^-20): the CF uses both components at O(1) magnitude together
    const long double ws = sinl((long double)x), wc = cosl((long double)x);
    e_sin = std::fmax(e_sin, fabsl(ws) > 1e-6L ? ulp_err(s, ws) : (double)(fabsl(s - ws) / 1.2e-22L));
    e_cos = std::fmax(e_cos, fabsl(wc) > 1e-6L ? ulp_err(c, wc) : (double)(fabsl(c - wc) / 1.2e-22L));
    {  // the wide reduction: up to +-2^45, where an angle still determines its sine to ~1e-3
      const double wm = std::exp2(18.0 + 27.0 * U(rng));
      const double wx = (U(rng) < 0.5 ? -wm : wm);
      double ws2, wc2;
      hh::fm::sincos_wide(wx, ws2, wc2);
      const long double rs = sinl((long double)wx), rc = cosl((long double)wx);
      e_wsin = std::fmax(e_wsin, fabsl(rs) > 1e-6L ? ulp_err(ws2, rs) : (double)(fabsl(ws2 - rs) / 1.2e-22L));
      e_wcos = std::fmax(e_wcos, fabsl(rc) > 1e-6L ? ulp_err(wc2, rc) : (double)(fabsl(wc2 - rc) / 1.2e-22L));
      hh::fm::sincos_wide(x, ws2, wc2);  // and it agrees with the narrow form on the narrow range
      e_wsin = std::fmax(e_wsin, fabsl(ws) > 1e-6L ? ulp_err(ws2, ws) : (double)(fabsl(ws2 - ws) / 1.2e-22L));
      const double ex = (U(rng) < 0.1 ? 1400.0 : 60.0) * (U(rng) - 0.5) * (U(rng) < 0.3 ? std::exp2(-20.0 * U(rng)) : 1.0);
      const long double we = expl((long double)ex);
      const double ge = hh::fm::exp(ex);
      if (we > 1e-300L && we < 1e300L) e_exp = std::fmax(e_exp, ulp_err(ge, we));
      if (hh::fm::exp_finite(ex) != ge) e_exp = 1e9;  // the form without clamp and NaN select: the same bits
    }
    const double lx = std::exp2(-600.0 + 1200.0 * U(rng)) * (1.0 + U(rng));
    e_log = std::fmax(e_log, ulp_err(hh::fm::log(lx), logl((long double)lx)));
    const double lx1 = 1.0 + (U(rng) - 0.5) * std::exp2(-40.0 * U(rng));  // around 1
    e_log = std::fmax(e_log, std::fabs(lx1 - 1.0) > 1e-300 ? ulp_err(hh::fm::log(lx1), logl((long double)lx1)) : 0.0);
    const double r = std::exp2(-40.0 + 80.0 * U(rng)), ph = 6.283185307179586 * U(rng);
    const double ay = r * std::sin(ph) * std::exp2(-30.0 * U(rng) * (U(rng) < 0.3)), ax = r * std::cos(ph);
    e_at = std::fmax(e_at, ulp_err(hh::fm::atan2(ay, ax), atan2l((long double)ay, (long double)ax)));
  }
  // the normal quantile against an 80-bit Newton refinement of Φ(x) = p, Φ by erfcl: body, both tails, far tails
  double e_nq = 0;
  for (int i = 0; i < 400000; ++i) {
    double p = U(rng);
    if (i % 4 == 1) p = std::exp2(-1000.0 * U(rng));            // far lower tail (third region below 1.4e-11)
    if (i % 4 == 2) p = 1.0 - std::exp2(-52.0 * U(rng));         // upper tail up to 1 - 2^-52
    if (!(p > 0.0 && p < 1.0)) continue;
    const double got = hh::fm::normal_quantile(p);
    long double x = got;
    for (int it = 0; it < 4; ++it) {
      const long double phi = expl(-0.5L * x * x) / sqrtl(2.0L * 3.14159265358979323846264338327950288L);
      // Φ(x) - p; in the upper half from the complements, 1 - p being exact and erfcl free of cancellation there
      const long double res = p < 0.5 ? 0.5L * erfcl(-x / sqrtl(2.0L)) - (long double)p
                                      : (1.0L - (long double)p) - 0.5L * erfcl(x / sqrtl(2.0L));
      x -= res / phi;
    }
    // near p = 1/2 the quantile passes through 0: absolute there
    e_nq = std::fmax(e_nq, fabsl(x) > 1e-3L ? ulp_err(got, x) : (double)(fabsl((long double)got - x) / 2.2e-19L));
  }
  if (!(hh::fm::normal_quantile(0.0) < -1e300) || !(hh::fm::normal_quantile(1.0) > 1e300) ||
      !std::isnan(hh::fm::normal_quantile(std::nan(""))) || hh::fm::normal_quantile(0.5) != 0.0)
    e_nq = 1e9;
  // axes and diagonals
  const double pts[][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}, {1, 1}, {-1, 1}, {1, -1}, {-1, -1},
                           {0.4375, 1}, {0.6875, 1}, {1, 0.4375}, {1e-300, 1}, {1, 1e-300}};
  for (auto& p : pts)
    e_at = std::fmax(e_at, ulp_err(hh::fm::atan2(p[0], p[1]), atan2l((long double)p[0], (long double)p[1])));
  // saturation and special values of exp
  if (!(hh::fm::exp_finite(-2000.0) == 0.0) || !std::isinf(hh::fm::exp_finite(2000.0)) ||
      !std::isnan(hh::fm::exp_finite(std::nan(""))) || !(hh::fm::exp_finite(-1e6) == 0.0))
    e_exp = 1e9;
  if (!(hh::fm::exp(-2000.0) == 0.0) || !std::isinf(hh::fm::exp(2000.0)) || !(hh::fm::exp(0.0) == 1.0) ||
      !std::isnan(hh::fm::exp(std::nan(""))) || !(hh::fm::exp(-745.0) > 0.0))
    e_exp = 1e9;
  std::printf("sin %d %.3f\ncos %d %.3f\nlog %d %.3f\natan2 %d %.3f\nexp %d %.3f\nwsin %d %.3f\nwcos %d %.3f\nnquant %d %.3f\n", N, e_sin,
              N, e_cos, N, e_log, N, e_at, N, e_exp, N, e_wsin, N, e_wcos, 400000, e_nq);
  return 0;
}
