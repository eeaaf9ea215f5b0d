// Host build of hedgehog.jl_amd/csrc/hh_bessel.h: reads "nu re im" lines, prints log I_nu(z) as
// "re im" (the imaginary part modulo 2 pi is what the kernels use).  tests/test_bessel_host.py
// compares with mpmath.  A point on the positive real axis is also put through besseli_logmul_re(), which must
// return the complex code's real parts bit for bit.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "hh_bessel.h"

int main() {
  double nu, re, im;
  double last_nu = NAN;
  hh::BesselTable t, t0;
  int n_int = 0;
  while (std::scanf("%lf %lf %lf", &nu, &re, &im) == 3) {
    if (!(nu == last_nu)) {
      n_int = nu >= 1.0 ? (int)std::floor(nu) : 0;
      if (!hh::bessel_table(nu, t) || !hh::bessel_table(nu - n_int, t0)) {
        std::printf("table-bound-violated\n");
        return 1;
      }
      last_nu = nu;
    }
    const hh::cx z = {re, im};
    const hh::LogMul r = hh::besseli_logmul(t, t0, n_int, z, std::atan2(im, re));
    if (im == 0.0 && re > 0.0) {  // the real-axis form must BE the complex one's real parts
      const hh::LogMulRe q = hh::besseli_logmul_re(t, t0, n_int, re);
      if (std::memcmp(&q.lg, &r.lg.re, 8) != 0 || std::memcmp(&q.mul, &r.mul.re, 8) != 0) {
        std::printf("real-axis-mismatch nu=%.17g x=%.17g: %.17g %.17g vs %.17g %.17g\n", nu, re, q.lg, q.mul,
                    r.lg.re, r.mul.re);
        return 1;
      }
    }
    const hh::cx lm = hh::clog(r.mul);
    std::printf("%.17g %.17g\n", r.lg.re + lm.re, r.lg.im + lm.im);
  }
  return 0;
}
