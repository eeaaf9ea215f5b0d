/*
 * hh_oracle.c — CPU restatement of the Monte Carlo solve() path of Hedgehog.jl.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the product
 * (hedgehog.jl_amd/, libhedgehog_mc.so); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call it, as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: per-draw parity with the reference is UNPINNED.  The reference is Julia and cannot
 * run in this build environment; its integrator, noise processes, RNG streams, NCχ² sampler, Bessel
 * functions and root finder live in third-party packages that are not vendored under
 * /root/reference (StochasticDiffEq 6, DiffEqNoiseProcess 5.24.1, Distributions 0.25, Random,
 * SpecialFunctions 2.5.0, Roots 2.2.6, ForwardDiff — Project.toml:6-45), and its own tests hold no
 * per-path golden vectors.  What pins this file is (i) the reference's known-answer
 * values for the analytic prices its MC tests aim at (test/unit/black_scholes.jl:93-127) and the
 * statistical tolerances of test/agreement/montecarlo_*.jl, checked in tests/test_oracle_pins.py,
 * and (ii) line-by-line correspondence with the cited source lines below.
 *
 * Arithmetic follows the reference line by line in fp64:
 *   sde_problem                      src/pricing_methods/montecarlo.jl:166-202
 *   LogGBMProblem / LogHestonProblem src/distributions/heston.jl:7-52
 *   EM() step  K = u + dt f(u); u' = K + g(.) dW   [StochasticDiffEq, third party; split form]
 *   simulate_paths (+ antithetic)    montecarlo.jl:342-375, 252-263
 *   final_sample                     montecarlo.jl:384-402
 *   marginal_law (lognormal)         montecarlo.jl:293-303
 *   reduce_payoffs / payoff functor  montecarlo.jl:428-432, src/payoffs/payoffs.jl:154-156
 *   solve                            montecarlo.jl:478-493
 *   dual numbers                     src/greeks/greeks_problem.jl:249-262 (ForwardDiff.derivative)
 *
 * Build: gcc -O2 -std=c11 -fPIC -shared -fopenmp -mavx2 -mfma -ffp-contract=off (oracle/Makefile).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/hedgehog_mc.h"

#define TILE HH_TILE_PATHS
#define MAXP HH_MAX_PARTIALS

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011; Random123 reference constants)              */
/* ------------------------------------------------------------------------------------------ */

void hho_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    if (r > 0) {
      k0 += 0x9E3779B9u; /* golden ratio */
      k1 += 0xBB67AE85u; /* sqrt(3) - 1  */
    }
    uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u01(uint32_t lo, uint32_t hi) {
  uint64_t w = ((uint64_t)hi << 32) | (uint64_t)lo;
  return ((double)(w >> 12) + 0.5) * 0x1p-52;
}

/* sin(pi t), cos(pi t) for t in (0, 2): exact quadrant reduction, then libm on |arg| <= pi/4 */
static void sincospi_02(double t, double* s, double* c) {
  double q = nearbyint(2.0 * t); /* nearest multiple of 1/2 */
  double r = t - 0.5 * q;        /* exact, |r| <= 1/4 */
  double sr = sin(M_PI * r), cr = cos(M_PI * r);
  switch (((int)q) & 3) {
    case 0: *s = sr;  *c = cr;  break;
    case 1: *s = cr;  *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
  }
}

/* Box–Muller pair from the Philox block of (key, c0, c1, c2, domain) */
void hho_normal_pair(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t dom, double* z1,
                     double* z2) {
  uint32_t ctr[4] = {c0, c1, c2, dom}, k[2] = {(uint32_t)key, (uint32_t)(key >> 32)}, o[4];
  hho_philox4x32_10(ctr, k, o);
  double u1 = u01(o[0], o[1]);
  double t = 2.0 * u01(o[2], o[3]);
  double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi_02(t, &s, &c);
  *z1 = r * c;
  *z2 = r * s;
}

/* two U(0,1) from the Philox block of (key, c0, c1, c2, domain) — used by oracle/bk_oracle.py */
void hho_uniform_pair(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t dom, double* u1,
                      double* u2) {
  uint32_t ctr[4] = {c0, c1, c2, dom}, k[2] = {(uint32_t)key, (uint32_t)(key >> 32)}, o[4];
  hho_philox4x32_10(ctr, k, o);
  *u1 = u01(o[0], o[1]);
  *u2 = u01(o[2], o[3]);
}

/* ------------------------------------------------------------------------------------------ */
/* REPLAY buffers                                                                              */
/* ------------------------------------------------------------------------------------------ */

static int ncomp_of(int dynamics) { return dynamics == HH_HESTON ? 2 : 1; }

size_t hho_replay_elems(uint64_t n_paths, uint32_t n_steps, int32_t dynamics) {
  uint64_t tiles = (n_paths + TILE - 1) / TILE;
  return (size_t)tiles * n_steps * (size_t)ncomp_of(dynamics) * TILE;
}

static size_t tile_index(uint64_t path, uint32_t step, int comp, uint32_t n_steps, int nc) {
  uint64_t tile = path / TILE, lane = path % TILE;
  return (size_t)(((tile * n_steps + step) * (uint64_t)nc + (uint64_t)comp) * TILE + lane);
}

/* increments of trajectory `key` at `step`: dW = sqrt(dt)·A·z, A the lower-triangular factor of
 * Γ = [1 ρ; ρ 1] (heston.jl:18-20; any A with A Aᵀ = Γ gives the reference's law) */
static void draw_increment(int nc, uint64_t key, uint32_t step, double sqrt_dt, double rho,
                           double rho_c, double* d1, double* d2) {
  double z1, z2;
  if (nc == 2) {
    hho_normal_pair(key, step, 0u, 0u, 0u, &z1, &z2);
    *d1 = sqrt_dt * z1;
    *d2 = sqrt_dt * fma(rho, z1, rho_c * z2);
  } else {
    hho_normal_pair(key, step >> 1, 0u, 0u, 0u, &z1, &z2);
    *d1 = sqrt_dt * ((step & 1u) ? z2 : z1);
    *d2 = 0.0;
  }
}

void hho_wiener_fill(int32_t dynamics, double rho, double T, uint32_t n_steps, uint64_t n_paths,
                     const uint64_t* seeds, double* dst) {
  const int nc = ncomp_of(dynamics);
  const double sqrt_dt = sqrt(T / (double)n_steps), rho_c = sqrt(1.0 - rho * rho);
  memset(dst, 0, hho_replay_elems(n_paths, n_steps, dynamics) * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n_paths; ++i) {
    for (uint32_t s = 0; s < n_steps; ++s) {
      double d1, d2;
      draw_increment(nc, seeds[i], s, sqrt_dt, rho, rho_c, &d1, &d2);
      dst[tile_index((uint64_t)i, s, 0, n_steps, nc)] = d1;
      if (nc == 2) dst[tile_index((uint64_t)i, s, 1, n_steps, nc)] = d2;
    }
  }
}

void hho_replay_pack(int32_t dynamics, uint64_t n_paths, uint32_t n_steps, const double* src,
                     double* dst) {
  const int nc = ncomp_of(dynamics);
  memset(dst, 0, hho_replay_elems(n_paths, n_steps, dynamics) * sizeof(double));
  for (uint64_t i = 0; i < n_paths; ++i)
    for (uint32_t s = 0; s < n_steps; ++s)
      for (int c = 0; c < nc; ++c)
        dst[tile_index(i, s, c, n_steps, nc)] = src[((size_t)i * n_steps + s) * nc + c];
}

/* ------------------------------------------------------------------------------------------ */
/* dual numbers (value + np partials), the ForwardDiff rules the path needs                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  double v;
  double d[MAXP];
} dual;

static dual dual_of(double v, const double* seeds, uint32_t np) {
  dual r;
  r.v = v;
  for (uint32_t k = 0; k < MAXP; ++k) r.d[k] = (seeds && k < np) ? seeds[k] : 0.0;
  return r;
}

/* ------------------------------------------------------------------------------------------ */
/* one Euler–Maruyama step                                                                     */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  dual kappa, theta, sigma, r, gdrift;
  double dt;
  uint32_t np;
  int split;
} step_params;

/* heston.jl:7-16: f(u) = [μ - 0.5 max(u2,0), κ(Θ - max(u2,0))], g(u) = [√max(u2,0), σ√max(u2,0)];
 * EM: K = u + dt f(u) (@muladd), u' = K + g(·) .* dW with g at K (split) or at u. */
static void heston_em_step(dual* x, dual* v, const step_params* p, double dW1, double dW2) {
  const uint32_t np = p->np;
  const int pos = v->v > 0.0;
  const double adj_var = pos ? v->v : 0.0; /* max(u[2], 0) */
  const double th_m_v = p->theta.v - adj_var;
  const double Kx = fma(p->dt, p->r.v - 0.5 * adj_var, x->v);
  const double Kv = fma(p->dt, p->kappa.v * th_m_v, v->v);
  const int wpos = p->split ? (Kv > 0.0) : pos;
  const double w = p->split ? (wpos ? Kv : 0.0) : adj_var;
  const double sq = sqrt(w); /* sqrt(max(u[2],0)), heston.jl:14 */
  /* ∂√(w⁺) = ∂w / (2√w) for w > 0; taken as 0 at the clip ("dual rules") */
  const double inv2s = wpos ? 0.5 / sq : 0.0;
  for (uint32_t k = 0; k < np; ++k) {
    const double vpd = pos ? v->d[k] : 0.0;
    const double Kxd = fma(p->dt, p->r.d[k] - 0.5 * vpd, x->d[k]);
    const double fvd = fma(p->kappa.d[k], th_m_v, p->kappa.v * (p->theta.d[k] - vpd));
    const double Kvd = fma(p->dt, fvd, v->d[k]);
    const double wd = p->split ? (wpos ? Kvd : 0.0) : vpd;
    const double sd = wd * inv2s;
    const double gvd = fma(p->sigma.d[k], sq, p->sigma.v * sd);
    x->d[k] = fma(sd, dW1, Kxd);
    v->d[k] = fma(gvd, dW2, Kvd);
  }
  x->v = fma(sq, dW1, Kx);
  v->v = fma(p->sigma.v * sq, dW2, Kv);
}

/* heston.jl:33-39: f = μ - 0.5σ², g = σ */
static void gbm_em_step(dual* x, const step_params* p, double dW) {
  for (uint32_t k = 0; k < p->np; ++k)
    x->d[k] = fma(p->sigma.d[k], dW, fma(p->dt, p->gdrift.d[k], x->d[k]));
  x->v = fma(p->sigma.v, dW, fma(p->dt, p->gdrift.v, x->v));
}

/* payoffs.jl:154-156 on S = exp(x) (montecarlo.jl:398) */
static double payoff_dual(const dual* x, const dual* strike, double cp, uint32_t np, double* S_out,
                          double* pd) {
  const double S = exp(x->v);
  const double m = cp * (S - strike->v);
  const int itm = m > 0.0;
  *S_out = S;
  for (uint32_t k = 0; k < np; ++k) pd[k] = itm ? cp * fma(S, x->d[k], -strike->d[k]) : 0.0;
  return itm ? m : 0.0;
}

/* ------------------------------------------------------------------------------------------ */
/* solve                                                                                       */
/* ------------------------------------------------------------------------------------------ */

int hho_mc_finalize(const hh_model* m, const hh_config* c, const double* acc, hh_result* out) {
  const double n = acc[HH_ACC_NPATHS];
  const double mean = acc[HH_ACC_SUM] / n;
  memset(out, 0, sizeof(*out));
  out->sum_payoff = acc[HH_ACC_SUM];
  out->sumsq_payoff = acc[HH_ACC_SUMSQ];
  out->price = m->discount * mean; /* montecarlo.jl:489-490 */
  double var = n > 1.0 ? (acc[HH_ACC_SUMSQ] - n * mean * mean) / (n - 1.0) : 0.0;
  if (!(var > 0.0)) var = 0.0;
  out->std_error = m->discount * sqrt(var / n);
  for (uint32_t k = 0; k < c->n_partials && k < MAXP; ++k) {
    const double dD = m->ddiscount ? m->ddiscount[k] : 0.0;
    out->dprice[k] = dD * mean + m->discount * (acc[HH_ACC_DSUM + k] / n);
  }
  out->n_paths_done = (uint64_t)n;
  return 0;
}

/*
 * Restatement of solve(prob, MonteCarlo(dynamics, EulerMaruyama | BlackScholesExact, config)).
 * `accum` (nullable) receives the HH_ACC_LEN sums; `terminal` (nullable) the samples at expiry
 * (n_paths, then n_paths mirrored ones when antithetic).  All buffers are host memory.
 * Returns 0, or -2 for combinations this file does not cover (Broadie–Kaya lives in
 * oracle/bk_oracle.py).
 */
int hho_mc_solve(const hh_model* m, const hh_config* c, hh_result* out, double* terminal,
                 double* accum, int n_threads) {
  if (!m || !c || !out) return -1;
  if (c->strategy == HH_BROADIE_KAYA) return -2;
  const uint32_t np = c->n_partials;
  const uint64_t N = c->n_paths;
  const int anti = c->antithetic != 0;
  const int replay = c->noise_mode == HH_NOISE_REPLAY;
  const int euler = c->strategy == HH_EULER_MARUYAMA;
  const int nc = euler ? ncomp_of(c->dynamics) : 1;
  const uint32_t M = euler ? c->n_steps : 1;

  /* montecarlo.jl:172-182, 196-201 */
  dual x0 = dual_of(log(m->S0), NULL, 0), v0 = dual_of(m->V0, m->dV0, np);
  for (uint32_t k = 0; k < np; ++k) x0.d[k] = (m->dS0 ? m->dS0[k] : 0.0) / m->S0;
  step_params sp;
  sp.kappa = dual_of(m->kappa, m->dkappa, np);
  sp.theta = dual_of(m->theta, m->dtheta, np);
  sp.sigma = dual_of(m->sigma, m->dsigma, np);
  sp.r = dual_of(m->r_drift, m->dr_drift, np);
  sp.gdrift = dual_of(m->r_drift - 0.5 * m->sigma * m->sigma, NULL, 0); /* heston.jl:35 */
  for (uint32_t k = 0; k < np; ++k) sp.gdrift.d[k] = sp.r.d[k] - m->sigma * sp.sigma.d[k];
  sp.dt = m->T / (double)M; /* montecarlo.jl:349 */
  sp.np = np;
  sp.split = c->em_split;
  const dual strike = dual_of(m->strike, m->dstrike, np);
  const double sqrt_dt = sqrt(sp.dt), rho_c = sqrt(1.0 - m->rho * m->rho);

  /* exact law, montecarlo.jl:302: Normal(log S0 + (r - σ²/2)·√α, σ·√α) (Q1: √α kept on request) */
  const double sqT = sqrt(m->T), tmul = c->compat_sqrt_alpha ? sqT : m->T;
  dual law_mu, law_sd;
  law_mu.v = x0.v + sp.gdrift.v * tmul;
  law_sd.v = m->sigma * sqT;
  for (uint32_t k = 0; k < MAXP; ++k) {
    law_mu.d[k] = k < np ? x0.d[k] + sp.gdrift.d[k] * tmul : 0.0;
    law_sd.d[k] = k < np ? sp.sigma.d[k] * sqT : 0.0;
  }

  const double* rp = c->replay;
  double* packed = NULL;
  if (replay && c->replay_layout == HH_REPLAY_PATH_MAJOR) {
    const int dyn = euler ? c->dynamics : HH_LOGNORMAL;
    packed = (double*)malloc(hho_replay_elems(N, M, dyn) * sizeof(double));
    if (!packed) return -4;
    hho_replay_pack(dyn, N, M, c->replay, packed);
    rp = packed;
  }

  double* pay = (double*)malloc((size_t)N * (1 + np) * sizeof(double));
  if (!pay) { free(packed); return -4; }

#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t ii = 0; ii < (int64_t)N; ++ii) {
    const uint64_t i = (uint64_t)ii;
    dual x = x0, v = v0, xa = x0, va = v0;
    if (euler) {
      /* simulate_paths: EM(), dt = T/steps; the mirrored run replays -W (montecarlo.jl:258) */
      const uint64_t key = replay ? 0 : c->seeds[i]; /* montecarlo.jl:331 */
      for (uint32_t s = 0; s < M; ++s) {
        double d1, d2 = 0.0;
        if (replay) {
          d1 = rp[tile_index(i, s, 0, M, nc)];
          if (nc == 2) d2 = rp[tile_index(i, s, 1, M, nc)];
        } else {
          draw_increment(nc, key, s, sqrt_dt, m->rho, rho_c, &d1, &d2);
        }
        if (c->dynamics == HH_HESTON) {
          heston_em_step(&x, &v, &sp, d1, d2);
          if (anti) heston_em_step(&xa, &va, &sp, -d1, -d2);
        } else {
          gbm_em_step(&x, &sp, d1);
          if (anti) gbm_em_step(&xa, &sp, -d1);
        }
      }
    } else {
      /* get_final_samples(::ExactSimulation): ONE stream keyed by seeds[1] (montecarlo.jl:456);
       * trajectory G takes component G&1 of block G>>1 */
      double z;
      if (replay) {
        z = rp[i];
      } else {
        const uint64_t G = c->path_offset + i;
        double z1, z2;
        hho_normal_pair(c->seeds[0], (uint32_t)(G >> 1), (uint32_t)(G >> 33), 0u, 1u, &z1, &z2);
        z = (G & 1ull) ? z2 : z1;
      }
      x.v = fma(law_sd.v, z, law_mu.v); /* rand(rng, Normal(μ, σ)) = μ + σ·randn */
      xa.v = 2 * law_mu.v - x.v;        /* montecarlo.jl:387 */
      for (uint32_t k = 0; k < np; ++k) {
        x.d[k] = fma(law_sd.d[k], z, law_mu.d[k]);
        xa.d[k] = 2 * law_mu.d[k] - x.d[k];
      }
    }
    /* final_sample + reduce_payoffs */
    double S, pd[MAXP], p = payoff_dual(&x, &strike, m->cp, np, &S, pd);
    if (terminal) terminal[i] = S;
    if (anti) {
      double Sa, pda[MAXP], pa = payoff_dual(&xa, &strike, m->cp, np, &Sa, pda);
      if (terminal) terminal[N + i] = Sa;
      p = (p + pa) / 2; /* montecarlo.jl:431 */
      for (uint32_t k = 0; k < np; ++k) pd[k] = (pd[k] + pda[k]) / 2;
    }
    pay[(size_t)i * (1 + np)] = p;
    for (uint32_t k = 0; k < np; ++k) pay[(size_t)i * (1 + np) + 1 + k] = pd[k];
  }

  /* mean(payoffs), montecarlo.jl:490 — summed in extended precision, fixed order */
  long double s = 0, s2 = 0, sd[MAXP] = {0};
  for (uint64_t i = 0; i < N; ++i) {
    const double p = pay[(size_t)i * (1 + np)];
    s += p;
    s2 += (long double)p * p;
    for (uint32_t k = 0; k < np; ++k) sd[k] += pay[(size_t)i * (1 + np) + 1 + k];
  }
  free(pay);
  free(packed);

  double acc[HH_ACC_LEN] = {0};
  acc[HH_ACC_SUM] = (double)s;
  acc[HH_ACC_SUMSQ] = (double)s2;
  for (uint32_t k = 0; k < np; ++k) acc[HH_ACC_DSUM + k] = (double)sd[k];
  acc[HH_ACC_NPATHS] = (double)N;
  if (accum) memcpy(accum, acc, sizeof(acc));
  return hho_mc_finalize(m, c, acc, out);
}

int hho_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* full spot grid for the LSM consumer (least_squares_montecarlo.jl:105-106)                    */
/* ------------------------------------------------------------------------------------------ */

/*
 * simulate_paths on the NoiseProblem of sde_problem(::LognormalDynamics, ::BlackScholesExact)
 * (montecarlo.jl:140-159): a GeometricBrownianMotionProcess stepped with dt = T/steps, each step an
 * exact lognormal increment dW = W (exp((μ-σ²/2) dt + σ √dt z) - 1) [DiffEqNoiseProcess, third party],
 * trajectory i seeded with seeds[i] (montecarlo.jl:331); the antithetic ensemble flips σ
 * (montecarlo.jl:270-284).  out[(s)*ntot + p], s = 0..n_steps, ntot = n_paths·(1+anti): the
 * transpose of extract_spot_grid's (nsteps+1) x npaths matrix (least_squares_montecarlo.jl:47-85).
 */
void hho_gbm_grid(const uint64_t* seeds, uint64_t n_paths, uint32_t n_steps, double S0, double r,
                  double sigma, double T, int anti, double* out) {
  const uint64_t ntot = n_paths * (anti ? 2u : 1u);
  const double dt = T / (double)n_steps;
  const double a = (r - 0.5 * sigma * sigma) * dt, b = sigma * sqrt(dt);
#pragma omp parallel for schedule(static)
  for (int64_t ii = 0; ii < (int64_t)n_paths; ++ii) {
    const uint64_t i = (uint64_t)ii;
    double S = S0, Sa = S0;
    out[i] = S0;
    if (anti) out[n_paths + i] = S0;
    for (uint32_t s = 0; s < n_steps; ++s) {
      double z1, z2;
      hho_normal_pair(seeds[i], s >> 1, 0u, 0u, 0u, &z1, &z2);
      const double z = (s & 1u) ? z2 : z1;
      S = S + S * (exp(fma(b, z, a)) - 1.0);
      out[(size_t)(s + 1) * ntot + i] = S;
      if (anti) {
        Sa = Sa + Sa * (exp(fma(-b, z, a)) - 1.0);
        out[(size_t)(s + 1) * ntot + n_paths + i] = Sa;
      }
    }
  }
}
