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
  double e_sin = 0, e_cos = 0, e_log = 0, e_at = 0;
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
    const double lx = std::exp2(-600.0 + 1200.0 * U(rng)) * (1.0 + U(rng));
    e_log = std::fmax(e_log, ulp_err(hh::fm::log(lx), logl((long double)lx)));
    const double lx1 = 1.0 + (U(rng) - 0.5) * std::exp2(-40.0 * U(rng));  // around 1
    e_log = std::fmax(e_log, std::fabs(lx1 - 1.0) > 1e-300 ? ulp_err(hh::fm::log(lx1), logl((long double)lx1)) : 0.0);
    const double r = std::exp2(-40.0 + 80.0 * U(rng)), ph = 6.283185307179586 * U(rng);
    const double ay = r * std::sin(ph) * std::exp2(-30.0 * U(rng) * (U(rng) < 0.3)), ax = r * std::cos(ph);
    e_at = std::fmax(e_at, ulp_err(hh::fm::atan2(ay, ax), atan2l((long double)ay, (long double)ax)));
  }
  // axes and diagonals
  const double pts[][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}, {1, 1}, {-1, 1}, {1, -1}, {-1, -1},
                           {0.4375, 1}, {0.6875, 1}, {1, 0.4375}, {1e-300, 1}, {1, 1e-300}};
  for (auto& p : pts)
    e_at = std::fmax(e_at, ulp_err(hh::fm::atan2(p[0], p[1]), atan2l((long double)p[0], (long double)p[1])));
  std::printf("sin %d %.3f\ncos %d %.3f\nlog %d %.3f\natan2 %d %.3f\n", N, e_sin, N, e_cos, N, e_log, N, e_at);
  return 0;
}
