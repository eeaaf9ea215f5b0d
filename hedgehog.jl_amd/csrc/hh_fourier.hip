// Carr–Madan Fourier price on the device: the analytic value the reference's
// Monte Carlo tests compare against (test/agreement/montecarlo_heston.jl:47-49, price_agreement.jl),
// so that a GPU box can self-check Heston / lognormal MC prices without a CPU-side quadrature.
//
// Reference: solve(prob, ::CarrMadan) src/pricing_methods/carr_madan.jl:47-71, call_transform
// :88-92, Heston log-price CF src/distributions/heston.jl:307-319, Normal-law CF
// src/distributions/sample_from_cf.jl:14-16, parity_transform src/payoffs/payoffs.jl:172-193.
// The reference integrates over (-bound, bound) with adaptive Gauss–Kronrod (QuadGK, third party);
// here 256 panels x 16-point Gauss–Legendre, one panel per lane — the integrand is entire and the
// panels are 2·bound/256 wide, so the rule is exact to rounding for the bounds the reference uses.
#include <cmath>

#include "hh_kernels.h"

namespace hh {

namespace {

struct cz {
  double re, im;
};
__device__ __forceinline__ cz operator+(cz a, cz b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cz operator-(cz a, cz b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cz operator*(cz a, cz b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cz operator*(double s, cz a) { return {s * a.re, s * a.im}; }
__device__ __forceinline__ cz zdiv(cz a, cz b) {
  const double inv = 1.0 / (b.re * b.re + b.im * b.im);
  return {(a.re * b.re + a.im * b.im) * inv, (a.im * b.re - a.re * b.im) * inv};
}
__device__ __forceinline__ cz zsqrt(cz z) {
  const double r = hypot(z.re, z.im);
  if (r == 0.0) return {0.0, 0.0};
  if (z.re >= 0.0) {
    const double t = sqrt(0.5 * (r + z.re));
    return {t, z.im / (2.0 * t)};
  }
  const double t = sqrt(0.5 * (r - z.re));
  return {fabs(z.im) / (2.0 * t), copysign(t, z.im)};
}
__device__ __forceinline__ cz zexp(cz z) {
  const double e = exp(z.re);
  double s, c;
  sincos(z.im, &s, &c);
  return {e * c, e * s};
}
__device__ __forceinline__ cz zlog(cz z) { return {log(hypot(z.re, z.im)), atan2(z.im, z.re)}; }

struct FourierArgs {
  int dynamics;
  double logS0, V0, kappa, theta, sigma, rho, r, T;  // Heston law (montecarlo.jl:310-320)
  double law_mu, law_sd;                             // lognormal law (montecarlo.jl:293-303)
  double alpha, bound, logK, discount;
  double* out;  // [0] = damped-call integral (the call price)
  // basket form (one workgroup per payoff): per-payoff scalars in device memory, [4][n]:
  // log K, T, r_drift, discount; NULL = the fields above
  const double* per_payoff;
  uint32_t n_payoffs, compat_sqrt_alpha;
  double sigma_ln;  // lognormal volatility (law_mu / law_sd are formed per payoff)
};

// heston.jl:307-319, complex argument u
__device__ cz heston_cf(const FourierArgs& a, cz u) {
  const cz iu = {-u.im, u.re};
  const cz kri = {a.kappa - a.rho * a.sigma * iu.re, -a.rho * a.sigma * iu.im};  // κ − ρσ·iu
  const cz u2 = u * u;
  const cz d1 = zsqrt(kri * kri + (a.sigma * a.sigma) * (iu + u2));
  const cz g = zdiv(kri - d1, kri + d1);
  const cz ed = zexp({-d1.re * a.T, -d1.im * a.T});
  const cz one = {1.0, 0.0};
  const cz one_m_ged = one - g * ed;
  const cz C = (a.kappa * a.theta / (a.sigma * a.sigma)) *
               (a.T * (kri - d1) - 2.0 * zlog(zdiv(one_m_ged, one - g)));
  const cz Dv = (1.0 / (a.sigma * a.sigma)) * ((kri - d1) * zdiv(one - ed, one_m_ged));
  return zexp(C + a.V0 * Dv + a.logS0 * iu + (a.r * a.T) * iu);
}

// sample_from_cf.jl:14-16: cf(Normal(μ, σ), t) = exp(i t μ − σ² t²/2)
__device__ cz normal_cf(const FourierArgs& a, cz t) {
  const cz it = {-t.im, t.re};
  return zexp(a.law_mu * it - (0.5 * a.law_sd * a.law_sd) * (t * t));
}

// carr_madan.jl:59-61, 88-92: damp · call_transform(v) · exp(−i v log K), real part
__device__ double integrand(const FourierArgs& a, double v) {
  const cz u = {v, -(a.alpha + 1.0)};
  const cz phi = a.dynamics == HH_HESTON ? heston_cf(a, u) : normal_cf(a, u);
  const cz den = {a.alpha * a.alpha + a.alpha - v * v, v * (2.0 * a.alpha + 1.0)};
  const cz val = zdiv(a.discount * phi, den) * zexp({0.0, -v * a.logK});
  return exp(-a.alpha * a.logK) / 6.28318530717958647692 * val.re;
}

// 16-point Gauss–Legendre on [-1, 1] (positive half; symmetric)
__constant__ double kGLx[8] = {0.0950125098376374401853193, 0.2816035507792589132304605,
                               0.4580167776572273863424194, 0.6178762444026437484466718,
                               0.7554044083550030338951012, 0.8656312023878317438804679,
                               0.9445750230732325760779884, 0.9894009349916499325961542};
__constant__ double kGLw[8] = {0.1894506104550684962853967, 0.1826034150449235888667637,
                               0.1691565193950025381893121, 0.1495959888165767320815017,
                               0.1246289712555338720524763, 0.0951585116824927848099251,
                               0.0622535239386478928628438, 0.0271524594117540948517806};

__global__ __launch_bounds__(256) void carr_madan_kernel(const FourierArgs a0) {
  FourierArgs a = a0;
  if (a0.per_payoff) {  // payoff blockIdx.x of a basket: its expiry-dependent scalars
    const uint32_t k = blockIdx.x, n = a0.n_payoffs;
    a.logK = a0.per_payoff[k];
    a.T = a0.per_payoff[n + k];
    a.r = a0.per_payoff[2 * n + k];
    a.discount = a0.per_payoff[3 * n + k];
    const double sqT = sqrt(a.T), tmul = a0.compat_sqrt_alpha ? sqT : a.T;
    a.law_mu = a.logS0 + (a.r - 0.5 * a0.sigma_ln * a0.sigma_ln) * tmul;  // montecarlo.jl:302
    a.law_sd = a0.sigma_ln * sqT;
    a.out = a0.out + k;
  }
  const double w = 2.0 * a.bound / 256.0;  // panel width
  const double mid = -a.bound + (threadIdx.x + 0.5) * w, half = 0.5 * w;
  double s = 0.0;
#pragma unroll 1
  for (int k = 0; k < 8; ++k)
    s += kGLw[k] * (integrand(a, mid - half * kGLx[k]) + integrand(a, mid + half * kGLx[k]));
  s *= half;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) a.out[0] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

}  // namespace

int launch_carr_madan_basket(const hh_model& m, int dynamics, int compat_sqrt_alpha, double alpha,
                             double bound, const double* per_payoff_dev, uint32_t n_payoffs,
                             double* out_dev, hipStream_t s) {
  FourierArgs a{};
  a.dynamics = dynamics;
  a.logS0 = log(m.S0); a.V0 = m.V0; a.kappa = m.kappa; a.theta = m.theta; a.sigma = m.sigma;
  a.rho = m.rho;
  a.sigma_ln = m.sigma;
  a.alpha = alpha; a.bound = bound;
  a.out = out_dev;
  a.per_payoff = per_payoff_dev;
  a.n_payoffs = n_payoffs;
  a.compat_sqrt_alpha = (uint32_t)(compat_sqrt_alpha != 0);
  hipLaunchKernelGGL(carr_madan_kernel, dim3(n_payoffs), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int launch_carr_madan(const hh_model& m, int dynamics, int compat_sqrt_alpha, double alpha,
                      double bound, double* out_dev, hipStream_t s) {
  FourierArgs a{};
  a.dynamics = dynamics;
  a.logS0 = log(m.S0); a.V0 = m.V0; a.kappa = m.kappa; a.theta = m.theta; a.sigma = m.sigma;
  a.rho = m.rho; a.r = m.r_drift; a.T = m.T;
  const double sqT = sqrt(m.T), tmul = compat_sqrt_alpha ? sqT : m.T;
  a.law_mu = a.logS0 + (m.r_drift - 0.5 * m.sigma * m.sigma) * tmul;  // montecarlo.jl:302
  a.law_sd = m.sigma * sqT;
  a.alpha = alpha; a.bound = bound; a.logK = log(m.strike); a.discount = m.discount;
  a.out = out_dev;
  hipLaunchKernelGGL(carr_madan_kernel, dim3(1), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace hh
