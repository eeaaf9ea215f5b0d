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

// ---- gradient of the price: complex numbers carrying P complex partials --------------------------
// The reference differentiates its calibration objective with ForwardDiff, i.e. pushes Duals through
// carr_madan.jl:47-92 and heston.jl:307-319.  Here the partials of the CF with respect to κ, σ, ρ are
// carried through the same expressions (3 directions); V0, θ, S0 and the rate enter log ϕ linearly:
//   log ϕ = C(κ,θ,σ,ρ) + V0·D(κ,σ,ρ) + iu (log S0 + rT),  C ∝ θ
//   ∂ϕ/∂V0 = ϕ D,  ∂ϕ/∂θ = ϕ C/θ,  ∂ϕ/∂S0 = ϕ iu/S0,  ∂ϕ/∂r = ϕ iu T.
template <int P>
struct zd {
  cz v;
  cz d[P];
};
template <int P>
__device__ __forceinline__ zd<P> operator+(const zd<P>& a, const zd<P>& b) {
  zd<P> r;
  r.v = a.v + b.v;
#pragma unroll
  for (int k = 0; k < P; ++k) r.d[k] = a.d[k] + b.d[k];
  return r;
}
template <int P>
__device__ __forceinline__ zd<P> operator-(const zd<P>& a, const zd<P>& b) {
  zd<P> r;
  r.v = a.v - b.v;
#pragma unroll
  for (int k = 0; k < P; ++k) r.d[k] = a.d[k] - b.d[k];
  return r;
}
template <int P>
__device__ __forceinline__ zd<P> operator*(const zd<P>& a, const zd<P>& b) {
  zd<P> r;
  r.v = a.v * b.v;
#pragma unroll
  for (int k = 0; k < P; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k];
  return r;
}
template <int P>
__device__ __forceinline__ zd<P> operator*(cz s, const zd<P>& a) {
  zd<P> r;
  r.v = s * a.v;
#pragma unroll
  for (int k = 0; k < P; ++k) r.d[k] = s * a.d[k];
  return r;
}
template <int P>
__device__ __forceinline__ zd<P> operator*(double s, const zd<P>& a) {
  return cz{s, 0.0} * a;
}
template <int P>
__device__ __forceinline__ zd<P> zdivd(const zd<P>& a, const zd<P>& b) {
  zd<P> r;
  const cz ib = zdiv({1.0, 0.0}, b.v);
  r.v = a.v * ib;
#pragma unroll
  for (int k = 0; k < P; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) * ib;
  return r;
}
template <int P>
__device__ __forceinline__ zd<P> zsqrtd(const zd<P>& a) {
  zd<P> r;
  r.v = zsqrt(a.v);
  const cz h = zdiv({0.5, 0.0}, r.v);
#pragma unroll
  for (int k = 0; k < P; ++k) r.d[k] = a.d[k] * h;
  return r;
}
template <int P>
__device__ __forceinline__ zd<P> zexpd(const zd<P>& a) {
  zd<P> r;
  r.v = zexp(a.v);
#pragma unroll
  for (int k = 0; k < P; ++k) r.d[k] = r.v * a.d[k];
  return r;
}
template <int P>
__device__ __forceinline__ zd<P> zlogd(const zd<P>& a) {
  zd<P> r;
  r.v = zlog(a.v);
  const cz ia = zdiv({1.0, 0.0}, a.v);
#pragma unroll
  for (int k = 0; k < P; ++k) r.d[k] = a.d[k] * ia;
  return r;
}
template <int P>
__device__ __forceinline__ zd<P> zconst(double x, int dir = -1) {  // a real parameter, seeded along `dir`
  zd<P> r;
  r.v = {x, 0.0};
#pragma unroll
  for (int k = 0; k < P; ++k) r.d[k] = {k == dir ? 1.0 : 0.0, 0.0};
  return r;
}

// C and D of heston.jl:307-319 with their partials along (κ, σ, ρ); same expressions as heston_cf
__device__ void heston_CD(const FourierArgs& a, cz u, zd<3>& C, zd<3>& Dv) {
  const zd<3> kappa = zconst<3>(a.kappa, 0), sigma = zconst<3>(a.sigma, 1), rho = zconst<3>(a.rho, 2);
  const cz iu = {-u.im, u.re};
  const zd<3> rs = rho * sigma;
  const zd<3> kri = kappa - iu * rs;                              // κ − ρσ·iu
  const zd<3> s2 = sigma * sigma;
  const zd<3> d1 = zsqrtd(kri * kri + (iu + u * u) * s2);
  const zd<3> g = zdivd(kri - d1, kri + d1);
  const zd<3> ed = zexpd(cz{-a.T, 0.0} * d1);
  const zd<3> one = zconst<3>(1.0);
  const zd<3> one_m_ged = one - g * ed;
  const zd<3> kms = zdivd(kappa, s2);                             // κ/σ²  (· θ below)
  C = a.theta * (kms * (a.T * (kri - d1) - 2.0 * zlogd(zdivd(one_m_ged, one - g))));
  Dv = zdivd((kri - d1) * zdivd(one - ed, one_m_ged), s2);
}

// Price and gradient of one payoff per workgroup.  out[k][0] = call price, out[k][1..7] = its partials
// along S0, V0, κ, θ, σ, ρ, r_drift (lognormal: S0, σ, r_drift in slots 1, 5, 7; the others 0).
constexpr int kGradVals = 8;
__global__ __launch_bounds__(256) void carr_madan_grad_kernel(const FourierArgs a0) {
  FourierArgs a = a0;
  const uint32_t kp = blockIdx.x, n = a0.n_payoffs;
  a.logK = a0.per_payoff[kp];
  a.T = a0.per_payoff[n + kp];
  a.r = a0.per_payoff[2 * n + kp];
  a.discount = a0.per_payoff[3 * n + kp];
  const double sqT = sqrt(a.T), tmul = a0.compat_sqrt_alpha ? sqT : a.T;
  a.law_mu = a.logS0 + (a.r - 0.5 * a0.sigma_ln * a0.sigma_ln) * tmul;
  a.law_sd = a0.sigma_ln * sqT;
  const double S0 = exp(a.logS0);
  const double w = 2.0 * a.bound / 256.0;
  const double mid = -a.bound + (threadIdx.x + 0.5) * w, half = 0.5 * w;
  const double damp = exp(-a.alpha * a.logK) / 6.28318530717958647692;
  double acc[kGradVals];
#pragma unroll
  for (int i = 0; i < kGradVals; ++i) acc[i] = 0.0;
  auto node = [&](double v, double wt) {
    const cz u = {v, -(a.alpha + 1.0)};
    const cz iu = {-u.im, u.re};
    const cz den = {a.alpha * a.alpha + a.alpha - v * v, v * (2.0 * a.alpha + 1.0)};
    const cz kern = (wt * damp * a.discount) * (zdiv({1.0, 0.0}, den) * zexp({0.0, -v * a.logK}));
    cz phi, dl[7];  // ∂ log ϕ along S0, V0, κ, θ, σ, ρ, r
#pragma unroll
    for (int i = 0; i < 7; ++i) dl[i] = {0.0, 0.0};
    if (a.dynamics == HH_HESTON) {
      zd<3> C, Dv;
      heston_CD(a, u, C, Dv);
      phi = zexp(C.v + a.V0 * Dv.v + (a.logS0 + a.r * a.T) * iu);
      dl[0] = (1.0 / S0) * iu;
      dl[1] = Dv.v;
      dl[2] = C.d[0] + a.V0 * Dv.d[0];
      dl[3] = (1.0 / a.theta) * C.v;
      dl[4] = C.d[1] + a.V0 * Dv.d[1];
      dl[5] = C.d[2] + a.V0 * Dv.d[2];
      dl[6] = a.T * iu;
    } else {  // sample_from_cf.jl:14-16: log ϕ = i t μ − σ_T² t²/2, μ = log S0 + (r − σ²/2)·tmul
      const cz t = u, it = iu, t2 = t * t;
      phi = zexp(a.law_mu * it - (0.5 * a.law_sd * a.law_sd) * t2);
      dl[0] = (1.0 / S0) * it;
      dl[4] = (-a0.sigma_ln * tmul) * it - (a0.sigma_ln * a.T) * t2;
      dl[6] = tmul * it;
    }
    const cz base = kern * phi;
    acc[0] += base.re;
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[1 + i] += (base * dl[i]).re;
  };
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    node(mid - half * kGLx[k], kGLw[k] * half);
    node(mid + half * kGLx[k], kGLw[k] * half);
  }
  __shared__ double sm[4][kGradVals];
#pragma unroll
  for (int i = 0; i < kGradVals; ++i) {
    double s = acc[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < kGradVals)
    a0.out[(size_t)kp * kGradVals + threadIdx.x] =
        ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
}

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

int launch_carr_madan_grad(const hh_model& m, int dynamics, int compat_sqrt_alpha, double alpha,
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
  hipLaunchKernelGGL(carr_madan_grad_kernel, dim3(n_payoffs), dim3(256), 0, s, a);
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
