// Broadie–Kaya exact Heston sampling on gfx950: one trajectory per lane, everything in registers.
//
// Per trajectory (reference: heston.jl:246-259 called through montecarlo.jl:416-419, 454-459):
//   1. V_T = c·NCχ²(d, λ)                                   heston.jl:125-133
//   2. ∫V | V0,V_T by Fourier inversion of its CF            heston.jl:140-212, sample_from_cf.jl
//        moments by central differences (:50-64), trapezoid CDF series with the reference's
//        stopping rule (:75-96), secant then the bisection / max_guess ladder (:105-135)
//   3. log S_T = μ + sqrt((1-ρ²)∫V)·Z                         heston.jl:278-300
// then S_T = exp(.), payoff and the workgroup reduction as in hh_kernels.hip.
//
// Launch structure — TWO kernels per solve (three with the caller's own draws):
//   bk_cf_kernel    one tile of 256 trajectories per workgroup: the three draws (GENERATE: made here; REPLAY: the
//                   caller's, read here; a grid's: by its variance kernel in front), characteristic function,
//                   moments, series terms, the secant inversion on the cached terms — and, for the trajectories
//                   whose secant failed (2 %), the bisection ladder of inverse_cdf, run by the WAVE: the lanes of a
//                   wave evaluate the next 3-6 levels of a failed trajectory's bisection tree at once (wave_ladder).
//                   The tiles of the chain's last round start at raised priority (the drain of a grid whose waves
//                   issue oldest-first; bk_cf_kernel)
//   bk_tail_kernel  (a) trajectories whose series outgrew the term cache run whole (none with the reference's
//                   controls: the workgroups find a zero and go on), (b) the records of both kernels are added into
//                   the accumulator.
// bk_tables_kernel (the Bessel tables of ν and the model's ϕ(0) constants into device memory) runs in front when
// they are not there already.
// (Until round 5 the chain was CF kernel -> prefix scan -> packed ladder kernel -> fall-back kernel -> record
// reduction: at 10^6 trajectories the four small kernels and their boundaries were 37 µs of 330 — the ladder alone
// 19 µs, one wave's dependent chain of fifteen CDF evaluations.  profiles/r05_n_bk_sizes.txt, r05_o_bk_ladder_stamps.txt.)
// (The draws had a launch of their own until round 5: with the library's pow / lgamma / log / normcdfinv and
// machine LICM hoisting their literals, sampler and CF arithmetic together took 228 registers per lane.  Built
// without that pass, and with hh_math.h's own normal quantile — the library's normcdfinv was a third of the draw
// kernel — the sampler takes 92 and the CF arithmetic 95: one kernel of 95, five waves per SIMD.)
//
// Third-party pieces of the reference restated here from their published algorithms (DESIGN.md
// "Broadie–Kaya"): complex log I_ν(z) for real ν > -1 (power series / Hankel asymptotics /
// backward ratio recurrence), NCχ² (normal shift for d > 1, else Poisson mixture; Marsaglia–Tsang
// gamma; inversion / PTRS Poisson), secant and bisection root finding.  Compute-bound (fp64 VALU);
// loop lengths are data dependent, so lanes of a wave diverge — see DESIGN.md for the measured cost.
//
// This file is compiled with -mllvm -disable-machine-licm (hedgehog.jl_amd/_build.py).  The machine-code LICM
// pass lifts every fp64 literal of log / exp / lgamma / sincos / Philox out of the rejection and root-search
// loops into a register of its own: 544 of them in the draw kernel of a grid (512 registers, 156 spilled),
// 248 registers + scratch in the fall-back kernel.  Without the pass the literals stay where they are used:
// the draw kernels take 92-120 registers with nothing in scratch (bk_draw_grid_kernel 0.173 -> 0.111 ms), and
// the hot CF kernel — whose Horner constants are SGPR literals already — drops from 122 to 105 registers and
// runs 2-5 % faster (interleaved A/B, tools/bk_ab.py: config 4 0.499 -> 0.474 ms, exact grid 1.736 -> 1.706).
#include <algorithm>
#include <atomic>
#include <cmath>

#include "hh_bessel.h"
#include "hh_kernels.h"
#include "hh_math.h"
#include "hh_reduce.h"
#include "hh_rng.h"

namespace hh {

namespace {

constexpr double kPi = kBesselPi;
constexpr double kTwoPi = kBesselTwoPi, kInvTwoPi = 0.15915494309189533577;
constexpr double kInvPi = 0.31830988618379067154, kTwoOverPi = 0.63661977236758134308;

#ifndef HH_BK_SLOTS
#define HH_BK_SLOTS 1536
#endif
#ifndef HH_BK_HEAVY_GRID
#define HH_BK_HEAVY_GRID 64
#endif
// Two switches of a TEST build (tests/c/build_bk_check.py, tests/test_gpu_bk_forms.py), which must give the shipped
// build's samples bit for bit: the trajectory's two real-axis evaluations through the complex code (1), and the
// bisection ladder run by the failed lane alone, statement for statement as sample_from_cf.jl:123-133 has it (1)
#ifndef HH_BK_COMPLEX_SETUP
#define HH_BK_COMPLEX_SETUP 0
#endif
#ifndef HH_BK_SERIAL_LADDER
#define HH_BK_SERIAL_LADDER 0
#endif


// Columns of cached series terms.  A lane's column belongs to a workgroup SLOT that a workgroup of the
// CF kernel (or of the ladder kernel behind it) takes when it starts and gives back when it is done —
// not to the trajectory: the cache is kSlots x 256 columns however many trajectories the chain has
// (1536 slots: the CF kernel takes 95 registers since its real-axis evaluations, so FIVE of its workgroups are
// resident per CU, 1280 in all — with 1024 slots the fifth spun for a slot and the kernel ran 3 % slower than at a
// forced four; with a slot for it, 5 % faster (profiles/r05_h_bk_ab.txt).  A bitmap word is 64 slots, hence 192
// per XCD.  256 terms of 8 bytes: 0.81 GB for 10^4 and for 10^8 trajectories alike).  A trajectory's terms are only needed again if its secant fails (2 % of
// them): the ladder kernel re-derives those.
constexpr int kSlots = HH_BK_SLOTS;
constexpr int kHeavyGrid = HH_BK_HEAVY_GRID;   // workgroups of the tail kernel (169 registers: it holds the whole-trajectory
                                  // code, idle with the reference's controls and the whole job when no series fits the
                                  // term cache; its first kRecStride workgroups then add the records)
static_assert(kHeavyGrid <= kSlots, "the tail kernel's workgroup b uses slot b");
static_assert(kHeavyGrid >= kRecStride, "one workgroup of the tail kernel per accumulator slot");

// The weight of term j of the CDF series (sample_from_cf.jl:86: (2/π)·sin(h j x)·Re ϕ(h j)/j) is (2/π)/j — an
// IEEE quotient, so the compile-time constants of the unrolled evaluation, this table (filled on the device by
// bk_tables_kernel, read with the loop's uniform j: scalar loads) and a division in the kernel (beyond the
// table) are the same numbers.
constexpr int kRootsMaxIters = 40;  // Roots.jl's own `maxiters` of a secant-type method (hh_config.bk_caps = 1)
constexpr int kCoefTerms = 1024;  // = the largest term cache (phi_cache_cap)
constexpr int kRegTerms = 16;     // series terms a lane holds in registers through its root search (load_terms)
struct BkBessel {
  BesselTable t[2];  // order ν, base order ν0
};
// What the characteristic function at argument 0 needs of the MODEL (κ, σ², T) — the part of evaluate_chf that does
// not depend on the trajectory, evaluated once (bk_tables_kernel) by the very code evaluate_chf runs: see chf_at_zero
struct CfZero {
  double x_re;    // Re(γ/(1−e)·e^{−γT/2}) at a = 0: ν_γ(0) = (4√(V0 VT)/σ²) · x_re
  double c0, c1;  // Re of the exponent: c0 + (V0+VT)/σ² · c1 + log I_ν(ν_γ) − log I_ν(ν_κ)
  double w_re;    // Re(ζ_κ γ/(1−e))
};
struct BkTables {
  BesselTable t[2];
  CfZero zero;
  double coef[kCoefTerms + 1];  // [j] = (2/π)/j, j >= 1
};
__device__ __forceinline__ double cdf_weight(const double* coef, int j) {
  return j <= kCoefTerms ? coef[uniform_index(j)] : kTwoOverPi / (double)j;
}

struct BkArgs {
  // model
  double kappa, theta, sigma, sigma2, inv_sigma2, rho, V0, T, logS0, r, strike, cp;
  // sample_V_T constants (heston.jl:128-130); λ = lam_num·V0 / lam_den
  double d, lam_num, lam_den, cscale;
  // HestonCFIterator constants (heston.jl:167-172)
  double nu, zeta_k, eta_k, nuk_factor;  // ν_κ = nuk_factor·sqrt(V0·VT)
  // Bessel tables (host-made: they depend on ν only) of ν and of the base order ν0 = ν - n_int, in
  // device memory (written by bk_tables_kernel), handed to the kernels that evaluate the CF as an
  // argument of their own.  NOT by value in this block: with 2 KB of tables in the kernel arguments, indexed in loops, the
  // compiler keeps the block in the argument segment only while it can follow every use — one more
  // inlined copy of the CF code and the whole block is copied to scratch, per lane, at kernel entry
  // (2.4 KB per lane; the CF kernel ran 2.2x slower).
  int n_int;
  const BkTables* tabs_dev;
  // controls (sample_from_cf.jl:27,50,75,105-113)
  double n_sigma, cf_tol, atol, moment_h;
  int newton_maxiter, bisect_maxiter;
  int root_form, bracket_form, caps;  // hh_config.bk_root_form / bk_bracket_form / bk_caps: other than 0, 0, 0 the
                                      // whole chain runs in the tail kernel (tail_whole_trajectory)
  // problem
  uint64_t n_paths, path_offset;
  const uint64_t* seeds;
  double* terminal;
  // one transition of a multi-date grid (HestonNoise, heston.jl:82-91): per-trajectory start state
  // read from row `step`, end state written to row `step + 1`; NULL = the one-shot terminal law
  const double* in_spot;
  const double* in_var;
  double* out_spot;
  double* out_var;
  uint32_t step;
  // several dates of a grid in ONE chain (launch_bk_grid): the chain's "trajectories" are the (date, trajectory)
  // pairs of the batch, in_var = the variance rows of the batch (row-major, so pair i starts at in_var[i]),
  // in_spot = NULL, and the chain leaves the sampled ∫V of pair i in iv_out[i] instead of a spot
  double* iv_out;
  double* iv_keep;                 // the one-shot law also LEAVES each trajectory's sampled ∫V here (iv_store): a model that
                                   // differs from this one in nothing the variance process sees (spot, rate, ρ, strike:
                                   // a bumped Greek's partner) is finished from it without a chain of its own
                                   // (launch_bk_refinish)
  const uint32_t* order;           // a chain that runs its pairs in another order (launch_bk_grid): position s of the
                                   // chain stands for pair order[s] — its draws, its start variance and its ∫V are
                                   // read and written THERE (pair_of); what the chain keeps for itself (decision words,
                                   // side-store indices, cached terms) is indexed by s.  NULL: s itself
  double* records;                 // [n_tiles + kHeavyGrid][kRecStride]: the CF kernel's tiles, then the tail kernel's workgroups
  double* draws;                   // [4][draw_stride]: Z, u, normal quantile of u, V_T per trajectory
  size_t draw_stride;
  const double* replay;            // REPLAY: the caller's [3][n_paths] V_T, u, Z (device), else NULL
  double* iv_store;                // [draw_stride]: where a grid chain keeps the sampled ∫V of its pairs (iv_out)
  unsigned long long* long_mask;   // [n_tiles][4] ballots: series longer than the cache, not inverted yet
                                   //              (bk_tail_kernel runs these whole)
  double* phi_cache;               // [cache_cap][cache_stride] cached Re ϕ(h·j), one column per lane of a
  size_t cache_stride;             //   workgroup SLOT (kSlots·256 columns), not per trajectory
  int cache_cap;
  uint32_t* slot_busy;             // slot bitmaps, one 128-byte line per XCD: 0 free / 1 taken.  Zeroed when the tables
                                   // beside them are made; every workgroup gives its slot back, so a chain leaves
                                   // them as it found them
  uint32_t static_slots;           // 1: the chain has at most kSlots tiles, slot = tile (no bitmap)
  uint32_t n_tiles;
  uint32_t* diag;                  // [2][draw_stride] per trajectory: decision word (BkDecision), series length
  uint32_t* counters;              // the 128-byte line behind the slot bitmaps, zeroed with them:
                                   //   [0] trajectories of this chain whose series outgrew the term cache (added to by
                                   //       the CF kernel's tiles, read and zeroed again by the tail kernel)
                                   //   [1] records of the last chain the reduction read (n_tiles, or n_tiles +
                                   //       kHeavyGrid when [0] was not zero): what bk_refinish_kernel must reproduce
                                   //   [2], [3] arrival counters of the tail kernel's two waits (zero between chains)
  uint32_t drain_tile;             // first tile of the chain's last round (n_tiles - the workgroups the device holds at
                                   // once; 0xffffffff for a chain of one round): from here on tiles start at raised priority
  void* args_dev;                  // a copy of this struct in device memory (written by tile 0 of the CF kernel)
                                   // for bk_tail_kernel, whose code is too large to inline: passing
                                   // a by-value kernel argument by reference to its functions would put
                                   // a 2 KB copy per lane in scratch
};

// where the draws / start state / results of the chain's position `path` lie
// (ORD: what the kernel was compiled for — the two hot kernels exist in both forms, so that the plain chain pays
// nothing for the order: the test in the kernel cost it 1 %; -1: look at p.order)
template <int ORD = -1>
__device__ __forceinline__ uint64_t pair_of(const BkArgs& p, uint64_t path) {
  if (ORD == 0) return path;
  if (ORD == 1) return (uint64_t)p.order[path];
  return p.order ? (uint64_t)p.order[path] : path;
}

// 1/x to <= 1 ulp: hardware reciprocal + two Newton steps (5 instructions; the IEEE division
// sequence is 12).  The quantities divided here are moderate in size: no scaling needed.
__device__ __forceinline__ double rcp_nr(double x) { return fm::rcp(x); }
__device__ __forceinline__ cx csqrt(cx z) {
  const double r = cabs(z);
  if (r == 0.0) return {0.0, 0.0};
  double h;
  if (z.re >= 0.0) {
    const double t = fm::sqrt_lean(0.5 * (r + z.re), &h);
    return {t, fm::div_by_2sqrt(z.im, t, h)};
  }
  const double t = fm::sqrt_lean(0.5 * (r - z.re), &h);
  return {fm::div_by_2sqrt(fabs(z.im), t, h), copysign(t, z.im)};
}

// csqrt() of a number that is not left of the imaginary axis (κ² − 2iσ²a): its first branch alone
__device__ __forceinline__ cx csqrt_right(cx z) {
  const double r = cabs(z);
  if (r == 0.0) return {0.0, 0.0};
  double h;
  const double t = fm::sqrt_lean(0.5 * (r + z.re), &h);
  return {t, fm::div_by_2sqrt(z.im, t, h)};
}

// per-trajectory CF state: HestonCFIterator (heston.jl:150-157)
struct CfIter {
  double VT, sqrtV0VT, logI_k, sumV;  // sumV = (V0+VT)/σ²
};

// The part of evaluate_chf that depends on the model and the argument a alone (heston.jl:186-196):
//   γ = sqrt(κ² − 2iσ²a), e^{−γT/2}, γ/(1−e^{−γT}), η_γ, and x = γ e^{−γT/2}/(1−e^{−γT}) (ν_γ = 4√(V0 VT)/σ² · x)
struct ChfPrefix {
  cx g, g_over_ome, eta_g, x;
};
__device__ __forceinline__ ChfPrefix chf_prefix(double kappa, double sigma2, double T, double a) {
  ChfPrefix f;
  f.g = csqrt_right({kappa * kappa, -2.0 * sigma2 * a});
  const cx eh = cexp_finite({-0.5 * f.g.re * T, -0.5 * f.g.im * T});  // exp(-γT/2): a finite exponent
  const cx e = eh * eh;                                        // exp(-γT)
  const cx ome = {1.0 - e.re, -e.im};
  const cx ope = {1.0 + e.re, e.im};
  // ζ_γ = (1−e)/γ, η_γ = γ(1+e)/(1−e), ν_γ = 4√(V0 VT) γ e^{−γT/2} / (σ²(1−e))  (heston.jl:188-196):
  // all three divide by 1−e, and ζ_γ only enters as ζ_κ/ζ_γ = ζ_κ γ/(1−e) — one complex reciprocal
  f.g_over_ome = f.g * crcp(ome);
  f.eta_g = f.g_over_ome * ope;
  f.x = f.g_over_ome * eh;
  return f;
}

// evaluate_chf (heston.jl:184-212).  theta_prev = NaN starts a new unwrapping sequence.
__device__ __forceinline__ cx evaluate_chf(const BkArgs& p, const BesselTable* bt, const CfIter& it, double a,
                                           double& theta_prev) {
  const ChfPrefix f = chf_prefix(p.kappa, p.sigma2, p.T, a);
  const cx g = f.g, g_over_ome = f.g_over_ome, eta_g = f.eta_g;
  const cx nu_g = (it.sqrtV0VT * 4.0 * p.inv_sigma2) * f.x;
  // continuous unwrapping of arg(ν_γ) (heston.jl:198-205)
  const double th = fm::atan2(nu_g.im, nu_g.re);
  double thu;
  if (isnan(theta_prev)) {
    thu = th;
  } else {
    // (the quotient by a multiply: an IEEE division is 13 instructions here; the two can only round to different
    // integers when the step is π to the last bit, where either branch continues the angle equally well)
    double dl = th - theta_prev;
    dl -= kTwoPi * rint(dl * kInvTwoPi);
    thu = theta_prev + dl;
  }
  theta_prev = thu;
  LogMul I = besseli_logmul(bt[0], bt[1], p.n_int, nu_g, th);  // principal branch at |ν_γ| cis(θ_unwrapped)
  I.lg.im += p.nu * (thu - th);            // + i ν (θ_unwrapped − θ)  (heston.jl:207)
  // ϕ = e^{-(γ-κ)T/2} (ζκ/ζγ) · exp((V0+VT)/σ² (ηκ-ηγ)) · Iγ / Iκ
  const cx ex = {-0.5 * (g.re - p.kappa) * p.T + it.sumV * (p.eta_k - eta_g.re) + I.lg.re - it.logI_k,
                 -0.5 * g.im * p.T - it.sumV * eta_g.im + I.lg.im};
  return (cexp(ex) * I.mul) * (p.zeta_k * g_over_ome);
}

// Re ϕ(0) as evaluate_chf(…, a = 0, θ_prev = the angle at a small a) computes it — bit for bit — in real arithmetic.
// At a = 0 every imaginary part above is a zero (γ = κ): the unwrapped angle comes out 0 whatever θ_prev was, the
// Bessel function is taken on the real axis, both sines are of zero, and the prefix is the same for every
// trajectory — bk_tables_kernel runs chf_prefix(κ, σ², T, 0) ONCE, with the code above, and leaves the four numbers
// that survive in CfZero.  What remains per trajectory is a product, besseli_logmul_re, one exponential and two
// products: ~150 instructions for ~510 (moments_from_cf evaluates ϕ(0) although it equals 1: cf_setup says why).
__device__ __forceinline__ CfZero chf_zero(double kappa, double sigma2, double T, double eta_k, double zeta_k) {
  const ChfPrefix f = chf_prefix(kappa, sigma2, T, 0.0);
  return {f.x.re, -0.5 * (f.g.re - kappa) * T, eta_k - f.eta_g.re, zeta_k * f.g_over_ome.re};
}
__device__ __forceinline__ double chf_at_zero(const BkArgs& p, const BesselTable* bt, const CfZero& z, const CfIter& it) {
  const double x0 = (it.sqrtV0VT * 4.0 * p.inv_sigma2) * z.x_re;
  const LogMulRe I = besseli_logmul_re(bt[0], bt[1], p.n_int, x0);
  const double ex = z.c0 + it.sumV * z.c1 + I.lg - it.logI_k;
  return (fm::exp(ex) * I.mul) * z.w_re;
}

// The series terms ϕ(h·j) of cdf_from_cf do not depend on x: the reference re-evaluates them in
// every one of the ~5-15 CDF evaluations of a trajectory's root search; here they are evaluated ONCE
// (in the reference's order, with its continuous phase unwrapping) and Re ϕ_j is kept in a
// per-lane column of a device scratch array, [term][lane] so a wave's accesses are contiguous.  The
// stopping index does not depend on x either.  Same values, same summation order, bit-identical
// results; a later evaluation costs one load + one sin per term instead of one complex-Bessel CF.
struct PhiCache {
  double* col;     // this lane's column: term j at col[(j-1)*stride]
  size_t stride;
  int cap;         // terms the column can hold
  int filled;      // terms evaluated so far (≥ j_stop once known)
  int j_stop;      // index of the last term (0 = not known yet)
  double theta_run;  // unwrapped angle after term `filled`
  double theta_cap;  // unwrapped angle after term `cap` (restart point when j_stop > cap)
};

// cdf_from_cf (sample_from_cf.jl:75-96)
__device__ double cdf_from_cf(const BkArgs& p, const BesselTable* bt, const double* coef, const CfIter& it, double x, double h,
                              PhiCache& c, double& n_terms) {
  if (x < 0.0) return 0.0;
  double result = (h * x) * kInvPi;
  const double stop = kPi * p.cf_tol / 2.0;
  double theta_tail = c.theta_cap;  // for terms beyond the cached ones
  // sin(h j x), j = 1, 2, …, by recurrence instead of one sin() call per term: the first kRegTerms by
  // sin((j+1)θ) = 2 cos θ sin(jθ) − sin((j−1)θ) (one fma; error ~ j² eps), the rest by rotation (four instructions;
  // ~ j eps), re-anchored every 32 terms — cdf_cached() says why, and runs the same operations
  double s1, c1;
  sincos_cf(h * x, s1, c1);
  double sj = s1, cj = c1, sp = 0.0;
  const double tc = c1 + c1;
  for (int j = 1; j < 1000000; ++j) {
    const double aj = h * (double)j;
    double re;
    bool last;
    if (j <= c.filled && j <= c.cap) {
      re = c.col[(size_t)(j - 1) * c.stride];
      last = j == c.j_stop;
    } else {
      double& th = (j <= c.cap || c.filled < c.cap) ? c.theta_run : theta_tail;
      const cx phi = evaluate_chf(p, bt, it, aj, th);
      re = phi.re;
      // |ϕ|/j < π·tol/2 (sample_from_cf.jl:88); also leaves on NaN and on an overflowed ϕ (a characteristic
      // function is bounded by 1: inf would keep the reference's stopping test false for 10^6 terms)
      // — and at term j_max = 2/(π·tol) + 1: |ϕ| <= 1 for a characteristic function, so the test MUST have
      // been met by then; a computed |ϕ| above 1 (variances of 1e16, cancellation gone wrong) would otherwise
      // run the series to its 10^6-term guard in every CDF evaluation
      const double aphi = cabs(phi);
      last = !(aphi >= stop * (double)j && aphi <= 0x1p100) || (double)j * stop > 1.0;
      if (j > c.filled) {  // first time this term is seen
        c.filled = j;
        if (j <= c.cap) c.col[(size_t)(j - 1) * c.stride] = re;
        if (j == c.cap) c.theta_cap = theta_tail = c.theta_run;  // (theta_tail: this very call goes on beyond the cache)
        if (last) c.j_stop = j;
      }
    }
    result = fma(sj, cdf_weight(coef, j) * re, result);
    n_terms += 1.0;
    if (last) break;
    if (j < kRegTerms) {  // the first terms by the three-term recurrence, as cdf_cached() takes them
      const double sn = fma(tc, sj, -sp);
      sp = sj;
      sj = sn;
    } else if (j == kRegTerms) {
      sincos_cf(h * x * (double)(j + 1), sj, cj);
    } else {
      const double sn = fma(sj, c1, cj * s1);
      cj = fma(cj, c1, -(sj * s1));
      sj = sn;
      if ((j & 31) == 0) sincos_cf(h * x * (double)(j + 1), sj, cj);  // re-anchor long series
    }
  }
  return result;
}

struct PathDraws {
  uint32_t k0, k1, c0, c1;
  __device__ Philox4 block(uint32_t b) const { return philox4x32_10(c0, c1, b, kDomBk, k0, k1); }
  __device__ void normals(uint32_t b, double& z1, double& z2) const { normal_pair(block(b), z1, z2); }
  __device__ void uniforms(uint32_t b, double& u1, double& u2) const {
    const Philox4 q = block(b);
    u1 = u01_from_bits(q.c0, q.c1);
    u2 = u01_from_bits(q.c2, q.c3);
  }
};

// Marsaglia–Tsang (2000), shape >= 1
__device__ double gamma_mt(double shape, const PathDraws& dr, int& it) {
  const double d = shape - 1.0 / 3.0;
  const double c = 1.0 / sqrt(9.0 * d);
  const int it0 = it;
  double v;
  while (true) {
    double x, u, unused;
    dr.normals(2u + 2u * (uint32_t)it, x, unused);
    dr.uniforms(3u + 2u * (uint32_t)it, u, unused);
    ++it;
    v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    const double x2 = x * x;
    if (u < 1.0 - 0.0331 * x2 * x2 || fm::log(u) < 0.5 * x2 + d * (1.0 - v + fm::log(v))) break;  // u, v > 0, normal
    if (it - it0 > 200) break;
  }
  return d * v;
}

__device__ double gamma_any(double shape, const PathDraws& dr, int& it, double u_boost) {
  if (shape >= 1.0) return gamma_mt(shape, dr, it);
  const double g = gamma_mt(shape + 1.0, dr, it);
  return g * fm::exp(fm::log(u_boost) / shape);  // u^(1/shape), u in [2^-53, 1): 3e-14 relative at worst (|log u| / shape <= 94 x 2.5 ulp)
}

__device__ int poisson(double mu, const PathDraws& dr, int& it) {
  if (mu < 10.0) {
    double u, unused;
    dr.uniforms(3u + 2u * (uint32_t)it, u, unused);
    ++it;
    double p = exp(-mu), F = p;
    int k = 0;
    while (u > F && k < 1000) {
      ++k;
      p *= mu / (double)k;
      F += p;
    }
    return k;
  }
  // PTRS (Hörmann 1993)
  const double smu = sqrt(mu), b = 0.931 + 2.53 * smu, a = -0.059 + 0.02483 * b;
  const double inv_alpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
  const int it0 = it;
  while (true) {
    double u1, V;
    dr.uniforms(3u + 2u * (uint32_t)it, u1, V);
    ++it;
    const double U = u1 - 0.5, us = 0.5 - fabs(U);
    const double k = floor((2.0 * a / us + b) * U + mu + 0.43);
    if (us >= 0.07 && V <= vr) return (int)k;
    if (k < 0.0 || (us < 0.013 && V > us)) {
      if (it - it0 > 200) return k > 0.0 ? (int)k : 0;
      continue;
    }
    if (log(V) + log(inv_alpha) - log(a / (us * us) + b) <= -mu + k * log(mu) - lgamma(k + 1.0))
      return (int)k;
    if (it - it0 > 200) return (int)k;
  }
}

// The three draws of a trajectory (reference order, heston.jl:246-259: V_T from the NCχ² law, then u
// inside sample_from_cf, then Z inside sample_log_S_T) and the normal quantile of u, left in
// draws[4][stride].  REPLAY: V_T, u, Z are the caller's (replay[3][n_paths]) —
// the seam through which the reference's own draws reach sample_from_cf / inverse_cdf per trajectory.
// The draws of one transition from start variance V0: stream `key`, indexed by G
__device__ __forceinline__ void draw_transition(const BkArgs& p, uint64_t key, uint64_t G, double V0, double& Z,
                                                double& u, double& VT) {
  const double lam = p.lam_num * V0 / p.lam_den;  // heston.jl:129
  const PathDraws dr{(uint32_t)key, (uint32_t)(key >> 32), (uint32_t)G, (uint32_t)(G >> 32)};
  double zshift, u_boost;
  dr.normals(0u, Z, zshift);
  dr.uniforms(1u, u, u_boost);
  // V_T (heston.jl:131)
  int it = 0;
  double chi;
  if (p.d > 1.0) {
    const double g = gamma_any(0.5 * (p.d - 1.0), dr, it, u_boost);
    const double sh = zshift + sqrt(lam);
    chi = sh * sh + 2.0 * g;
  } else {
    const int n = poisson(0.5 * lam, dr, it);
    chi = 2.0 * gamma_any(0.5 * p.d + (double)n, dr, it, u_boost);
  }
  // With d = 4κθ/σ² far below 1 the variance is absorbed at zero with real probability and the gamma
  // draw u^(1/shape) underflows to an exact 0 (d = 0.012: 1.4 % of the trajectories).  The CF of ∫V has
  // a finite limit for V_T -> 0 (the Bessel ratio tends to (ν_γ/ν_κ)^ν), but at V_T = 0 itself
  // log I_ν(0) is ±inf and the ratio NaN — in the reference too (heston.jl:193,207), whose series loop
  // then never ends.  The smallest variance kept is 2^-1000: the limit, to every digit.
  VT = fmax(p.cscale * chi, 0x1p-1000);
}

__device__ __forceinline__ void store_draws(const BkArgs& p, uint64_t i, double Z, double u, double VT) {
  double* d = p.draws + i;  // trajectory index == lane index of the CF kernels' tiles
  d[0] = Z;
  d[p.draw_stride] = u;
  d[2 * p.draw_stride] = fm::normal_quantile(u);  // quantile(Normal(), u) (sample_from_cf.jl:33)
  d[3 * p.draw_stride] = VT;
}

// … for one transition of a grid run date by date (launch_bk with a BkTransition): the trajectory's own seed
// (montecarlo.jl:331), indexed by the transition.  (The one-shot law's draws are made, or read from the caller's
// buffer, by the CF kernel itself.)
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(4, 8))) void bk_draw_kernel(const BkArgs p) {
  const uint64_t path = (uint64_t)blockIdx.x * kTile + threadIdx.x;
  if (path >= p.n_paths) return;
  double Z, u, VT;
  draw_transition(p, p.seeds[path], (uint64_t)p.step, p.in_var[path], Z, u, VT);
  store_draws(p, path, Z, u, VT);
}

// The key a batched grid chain orders its (date, trajectory) pairs by (launch_bk_grid): the size of the Bessel
// argument, V0·V_T, coarsely — exponent + three mantissa bits = eighth-octave bins, counted from 2^-28 and cut to ONE
// radix digit: products below 2^-28 share bin 0, above 2^4 bin 255 (variances from 1e-4 to 1 give 2^-27 … 2^0).
// Both factors are floored at 2^-1000, so the product is positive, and positive doubles order like their bits.
__device__ __forceinline__ uint32_t grid_order_key(double v0, double vt) {
  const int k = (int)((unsigned long long)__double_as_longlong(v0 * vt) >> 49) - ((1023 - 28) << 3);
  return (uint32_t)(k < 0 ? 0 : k > 255 ? 255 : k);
}

// The variance chain of dates k0 … k0 + n_dates of a grid, one trajectory per thread: V of each date from the
// one before (cheap: one non-central χ² draw), its draws left where the chain's pair (date, trajectory) =
// b·n_row + trajectory finds them, the variance rows written on the way.
// (keys: non-NULL when the chain will run its pairs in order — each pair's key, for the counting sort)
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(4, 8))) void bk_draw_grid_kernel(const BkArgs p, uint64_t n_row, uint32_t k0,
                                                             uint32_t n_dates, double* __restrict__ var_rows,
                                                             uint8_t* __restrict__ keys) {
  const uint64_t path = (uint64_t)blockIdx.x * kTile + threadIdx.x;
  if (path >= n_row) return;
  const uint64_t key = p.seeds[path];
  double V = var_rows[path];  // row k0
  for (uint32_t b = 0; b < n_dates; ++b) {
    double Z, u, VT;
    draw_transition(p, key, (uint64_t)(k0 + b), V, Z, u, VT);
    const uint64_t pair = (uint64_t)b * n_row + path;
    store_draws(p, pair, Z, u, VT);
    var_rows[(uint64_t)(b + 1) * n_row + path] = VT;
    if (keys) keys[pair] = (uint8_t)grid_order_key(V, VT);  // uniform
    V = VT;
  }
}

// HestonCFIterator (heston.jl:165-176) and the moment heuristics (sample_from_cf.jl:31-37) of one
// trajectory, from its start variance, its V_T and the normal quantile of its uniform
__device__ __forceinline__ void cf_setup(const BkArgs& p, const BesselTable* bt, double V0, double VT, double q_u, CfIter& cf,
                                         double& initial_guess, double& max_guess, double& h) {
  cf.VT = VT;
  const double v0vt = V0 * VT;  // (a grid's start variance can itself be the 2^-1000 floor: no underflow to 0)
  cf.sqrtV0VT = v0vt >= 0x1p-960 ? sqrt(v0vt) : sqrt(V0) * sqrt(VT);
  cf.sumV = (V0 + VT) / p.sigma2;
#if HH_BK_COMPLEX_SETUP  // a test build: both real evaluations through the complex code, as until round 5
  const LogMul Ik = besseli_logmul(bt[0], bt[1], p.n_int, {p.nuk_factor * cf.sqrtV0VT, 0.0}, 0.0);
  cf.logI_k = Ik.lg.re + fm::log(Ik.mul.re);
#else
  const LogMulRe Ik = besseli_logmul_re(bt[0], bt[1], p.n_int, p.nuk_factor * cf.sqrtV0VT);
  cf.logI_k = Ik.lg + fm::log(Ik.mul);  // real, positive argument: I_ν > 0
#endif
  // moments_from_cf (sample_from_cf.jl:50-61): mean = Re(-i ϕ'(0)), variance = Re(-ϕ''(0)) - mean²
  // by central differences of step hm over ϕ(hm), ϕ(0), ϕ(-hm).  The law is real, so ϕ(-a) is the
  // conjugate of ϕ(a) — operation by operation, also in floating point — and is not evaluated.
  // ϕ(0) IS evaluated although it equals 1: its rounding error (1e-13 when the Bessel logarithms
  // are ~10³, short steps) is common to ϕ(±hm) and cancels in the second difference, which is
  // formed from numbers 1e-10 apart.
  double th = __builtin_nan("");
  const double hm = p.moment_h;
  const cx pp = evaluate_chf(p, bt, cf, hm, th);
#if HH_BK_COMPLEX_SETUP
  const double p0_re = evaluate_chf(p, bt, cf, 0.0, th).re;
#else
  // (bt is BkTables::t: the model's ϕ(0) constants lie behind the two Bessel tables)
  const double p0_re = chf_at_zero(p, bt, reinterpret_cast<const BkTables*>(bt)->zero, cf);
#endif
  const double mean = pp.im / hm;                                          // (ϕ₊ - ϕ₋)/(2h)
  const double var = -(2.0 * (pp.re - p0_re) / (hm * hm)) - mean * mean;  // (ϕ₊ - 2ϕ₀ + ϕ₋)/h²
  const double sd = sqrt(fmax(var, 1e-12));
  const double normal_sample = mean + sd * q_u;
  initial_guess = normal_sample > 0.0 ? normal_sample : mean * 0.01;
  max_guess = mean + 11.0 * sd;
  h = kPi / (mean + p.n_sigma * sd);
}

// everything a trajectory of the fall-back kernel keeps THROUGH the CDF inversion (what only the finish
// needs — Z, V_T, the start state — is read again there: registers, not bandwidth, are short here)
struct PathSetup {
  double u;
  CfIter cf;
  PhiCache cache;
  double initial_guess, max_guess, h;
};

__device__ __forceinline__ void bk_setup(const BkArgs& p, const BesselTable* bt, uint64_t path, PathSetup& s) {
  s.cache.col = p.phi_cache + (size_t)blockIdx.x * kTile + threadIdx.x;  // slot b of workgroup b: this kernel
  s.cache.stride = p.cache_stride;                                        // runs alone, behind the other two
  s.cache.cap = p.cache_cap;
  s.cache.filled = 0;
  s.cache.j_stop = 0;
  s.cache.theta_run = __builtin_nan("");
  s.cache.theta_cap = __builtin_nan("");
  const uint64_t src = pair_of(p, path);
  const double V0 = p.in_var ? p.in_var[src] : p.V0;
  const double* d = p.draws + src;
  s.u = d[p.draw_stride];
  cf_setup(p, bt, V0, d[3 * p.draw_stride], d[2 * p.draw_stride], s.cf, s.initial_guess, s.max_guess, s.h);
}

// 3. log S_T (heston.jl:288-297), S_T = exp(.) (montecarlo.jl:384), payoff
__device__ __forceinline__ double bk_spot(const BkArgs& p, double logS0, double V0, double VT, double Z,
                                          double IV) {
  const double mu = logS0 + p.r * p.T - 0.5 * IV +
                    (p.rho / p.sigma) * (VT - V0 - p.kappa * p.theta * p.T + p.kappa * IV);
  const double sigma2 = (1.0 - p.rho * p.rho) * IV;
  return fm::exp(mu + sqrt(sigma2) * Z);
}
__device__ __forceinline__ double bk_finish(const BkArgs& p, double logS0, double V0, double VT,
                                            double Z, double IV, uint64_t path) {
  if (p.iv_out) {  // a batch of dates: the spot rows are chained afterwards (bk_grid_spots_kernel)
    p.iv_out[path] = IV;
    return 0.0;
  }
  if (p.iv_keep) p.iv_keep[path] = IV;
  const double S = bk_spot(p, logS0, V0, VT, Z, IV);
  if (p.terminal) p.terminal[path] = S;
  if (p.out_spot) {
    p.out_spot[path] = S;
    p.out_var[path] = VT;
  }
  const double m = p.cp * (S - p.strike);
  return m > 0.0 ? m : 0.0;
}
// … and the finish of that trajectory from its sampled ∫V
__device__ __forceinline__ double bk_finish_path(const BkArgs& p, uint64_t path, double IV) {
  const uint64_t src = pair_of(p, path);
  const double V0 = p.in_var ? p.in_var[src] : p.V0;
  const double logS0 = p.in_spot ? fm::log(p.in_spot[src]) : p.logS0;  // heston.jl:84: S = exp(W[1]), then log(S0) :289
  const double* d = p.draws + src;
  return bk_finish(p, logS0, V0, d[3 * p.draw_stride], d[0], IV, src);
}

// … and the spot rows of the same dates once the chain has left ∫V of every pair in iv_out: log S chained date by
// date, the arithmetic of bk_finish on the same operands (bit-identical with the launch-per-date form)
__global__ __launch_bounds__(kTile) void bk_grid_spots_kernel(const BkArgs p, uint64_t n_row, uint32_t n_dates,
                                                              const double* __restrict__ var_rows,
                                                              double* __restrict__ spot_rows) {
  const uint64_t path = (uint64_t)blockIdx.x * kTile + threadIdx.x;
  if (path >= n_row) return;
  double S = spot_rows[path], V0 = var_rows[path];
  for (uint32_t b = 0; b < n_dates; ++b) {
    const uint64_t i = (uint64_t)b * n_row + path;
    const double VT = p.draws[3 * p.draw_stride + i];
    S = bk_spot(p, fm::log(S), V0, VT, p.draws[i], p.iv_out[i]);
    spot_rows[(uint64_t)(b + 1) * n_row + path] = S;
    V0 = VT;
  }
}

// (long_counter, when given, is added the tile's number of too-long trajectories — the sum of `n_long` — when there
// are any: what tells bk_tail_kernel that it has work)
__device__ __forceinline__ void bk_store_record(double (&acc)[6], double* rec, uint32_t* long_counter = nullptr,
                                                double n_long = 0.0) {
  if (long_counter) {  // uniform
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_long += __shfl_down(n_long, off, 64);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_down(acc[i], off, 64);
  }
  __shared__ double sm[kTile / 64][7];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) sm[wave][i] = acc[i];
    sm[wave][6] = n_long;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      t[i] = sm[0][i];
      for (int w = 1; w < kTile / 64; ++w) t[i] += sm[w][i];
    }
    if (long_counter) {
      double nl = sm[0][6];
      for (int w = 1; w < kTile / 64; ++w) nl += sm[w][6];
      if (nl > 0.0) atomicAdd(long_counter, (uint32_t)nl);
    }
    for (int i = 0; i < kRecStride; ++i) rec[i] = 0.0;
    rec[HH_ACC_SUM] = t[0];
    rec[HH_ACC_SUMSQ] = t[1];
    rec[HH_ACC_BK_NEWTON_FAIL] = t[2];
    rec[HH_ACC_BK_BISECT] = t[3];
    rec[HH_ACC_BK_MAXGUESS] = t[4];
    rec[HH_ACC_BK_CF_TERMS] = t[5];
  }
}

// secant of inverse_cdf (sample_from_cf.jl:116-122) on a CDF given as a callable: Order2 restated as
// the secant iteration from (x0 + dx, x0), dx = h + |x0| h², h = eps^(1/3)
// A trajectory's decision word (BkArgs::diag, hh_bk_decisions): what the root search of inverse_cdf
// DID — the number of CDF evaluations of the secant, which branch finished it, the bisection's
// iterations.  Two implementations that agree on this word (and on the series length) have run the
// same sequence of floating-point operations up to rounding; where they differ, a stopping test sat
// within rounding of its threshold.
enum BkDecision : uint32_t {
  kDecEvalsMask = 0xffu,        // CDF evaluations of the secant iteration (2 … newton_maxiter)
  kDecBisect = 1u << 8,         // secant failed, bisection finished it (sample_from_cf.jl:127-133)
  kDecMaxGuess = 2u << 8,       // secant failed, no sign change: max_guess (:124-126)
  kDecItersShift = 16,          // bits 16-23: bisection iterations
  kDecLongSeries = 1u << 31     // series longer than the term cache: ran whole in bk_fallback_kernel
};

template <class Cdf>
__device__ __forceinline__ bool secant_inverse(Cdf&& cdf, double u, double guess, double atol,
                                               int maxiter, double& root, uint32_t& evals_out) {
  const double hs = 6.0554544523933395e-06;
  double x1 = guess;
  double x0 = x1 + hs + fabs(x1) * hs * hs;
  double f0 = cdf(x0) - u;
  double f1 = cdf(x1) - u;
  int evals = 2;
  bool ok = false;
  while (true) {
    if (fabs(f1) <= atol) {
      ok = true;
      break;
    }
    if (evals >= maxiter || f1 == f0) break;
    const double x2 = x1 - f1 * (x1 - x0) / (f1 - f0);
    if (!isfinite(x2)) break;
    x0 = x1;
    f0 = f1;
    x1 = x2;
    f1 = cdf(x2) - u;
    ++evals;
  }
  root = x1;
  evals_out = (uint32_t)evals;
  return ok && !(x1 < 0.0);
}

// (what the phases of the CF kernel hand each other through LDS: see wave_ladder)
constexpr int kLadderStash = 32;
struct LadderShared {
  double h[kTile], u[kTile], max_guess[kTile];  // in
  double guess[kTile];                          // the secant's first guess (series_phase -> invert_phase)
  double iv[kTile];                             // out: the sampled ∫V
  int j_stop[kTile];                            // in
  uint32_t res[kTile];                          // out: bisection iterations | max_guess branch << 31
  double terms[kTile / 64][8][kRegTerms];       // [wave][group]: (2/π)/j · Re ϕ(h j), zero beyond the series' end
  unsigned long long fail[kTile / 64];          // the waves' ballots of failed lanes
  // a failed lane leaves the sixteen weighted terms it holds in registers here (entry = the order of arrival, kept in
  // stash_of[thread]), so the ladder starts without a round trip to the term cache; the 33rd failure of a tile finds
  // no entry and its group reads the column instead — same numbers
  double stash[kLadderStash][kRegTerms];
  unsigned char stash_of[kTile];
  uint32_t stash_n;
};
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

// Series phase: the characteristic-function work of a trajectory — CF iterator, moments, and the
// series terms Re ϕ(h·j), j = 1 … J (J set by the reference's stopping rule, sample_from_cf.jl:88),
// evaluated ONCE in the reference's order with its continuous phase unwrapping and left in the
// trajectory's column of the cache.  Nothing here depends on the CDF argument, so the root search
// needs none of the complex Bessel machinery.  Everything a lane keeps here is the CF state — that
// is what fits 96 registers (5 waves per SIMD).  Returns h and the series length (0: longer than the cache —
// the tail kernel runs this trajectory whole); leaves the secant's first guess and max_guess in LDS.
template <int ORD>
__device__ __forceinline__ void series_phase(const BkArgs& p, const BesselTable* bt, uint64_t path, double* col,
                                             size_t col_stride, LadderShared& sh, double& h, int& j_stop) {
  const bool grid = p.in_var != nullptr;
  const uint64_t src = pair_of<ORD>(p, path);
  const double V0 = grid ? p.in_var[src] : p.V0;
  const double* d = p.draws + src;
  const double q_u = d[2 * p.draw_stride];
  const double VT = d[3 * p.draw_stride];
  CfIter cf;
  {  // (not in registers through the series: they are what this kernel is short of)
    double initial_guess, max_guess;
    cf_setup(p, bt, V0, VT, q_u, cf, initial_guess, max_guess, h);
    sh.guess[threadIdx.x] = initial_guess;
    sh.max_guess[threadIdx.x] = max_guess;
  }
  const double stop = kPi * p.cf_tol / 2.0;
  double theta = __builtin_nan("");
  j_stop = 0;
  for (int j = 1; j <= p.cache_cap; ++j) {
    const cx phi = evaluate_chf(p, bt, cf, h * (double)j, theta);
    col[(size_t)(j - 1) * col_stride] = phi.re;
    const double sj = stop * (double)j;
    const double mag2 = fma(phi.re, phi.re, phi.im * phi.im);
    if (!(mag2 >= sj * sj && mag2 <= 0x1p200)) {  // |ϕ|/j < π·tol/2, squared; also leaves on NaN / overflow
      j_stop = j;
      break;
    }
  }
}

// The first kRegTerms series terms of a trajectory, held in registers for all the CDF evaluations of
// its root search (H252: 10-13 terms, ~5 evaluations by the secant, ~13 by the ladder): the cache
// column is read once instead of once per evaluation — and each term already times its weight (2/π)/j
// (cdf_weight), zero beyond the series' end: an evaluation is then one rotation step and ONE fma per term, no
// test (adding sin·0 changes nothing).  Was 14 instructions per term: three products, the sum, the test and its
// selects, the term counter and its selects.
__device__ __forceinline__ void load_terms(const double* col, size_t stride, int j_stop,
                                           double (&t)[kRegTerms]) {
#pragma unroll
  for (int j = 0; j < kRegTerms; ++j) t[j] = j < j_stop ? (kTwoOverPi / (double)(j + 1)) * col[(size_t)j * stride] : 0.0;
}

// cdf_from_cf (sample_from_cf.jl:75-96) on the cached terms: same values, same summation order as
// cdf_from_cf() above, one rotation step per term; terms beyond the registers come from the column
template <class Terms>
__device__ __forceinline__ double cdf_cached(const Terms& t, const double* col, const double* coef,
                                             size_t stride, int j_stop, double h, double x,
                                             double& n_terms) {
  if (x < 0.0) return 0.0;
  double result = (h * x) * kInvPi;
  double s1, c1;
  sincos_cf(h * x, s1, c1);
  double sj = s1, cj = c1;
  n_terms += (double)j_stop;
  // sin((j+1)θ) = 2 cos θ · sin(jθ) − sin((j−1)θ): one fma per term where the rotation takes four instructions
  // (a sixth of the inversion's work).  Its error grows like j² eps (a double root as θ -> 0) — 3e-14 over the sixteen
  // register terms, nine orders below the inversion's tolerance — so the column terms beyond them keep the
  // rotation, re-anchored.  Config 4: -6 %.
  {
    const double tc = c1 + c1;
    double sp = 0.0;
#pragma unroll
    for (int j = 1; j <= kRegTerms; ++j) {
      result = fma(sj, t[j - 1], result);
      const double sn = fma(tc, sj, -sp);
      sp = sj;
      sj = sn;
      // (the two chains in step: left to itself the scheduler runs the sines ahead of the sum — sixteen of them
      // live while the first evaluation's terms are still loading — and the kernel takes 107 registers instead of
      // 95: four waves per SIMD instead of five)
      asm volatile("" : "+v"(result), "+v"(sj));
    }
    if (j_stop > kRegTerms) sincos_cf(h * x * (double)(kRegTerms + 1), sj, cj);
  }
  for (int j = kRegTerms + 1; j <= j_stop; ++j) {
    result = fma(sj, cdf_weight(coef, j) * col[(size_t)(j - 1) * stride], result);
    if (j == j_stop) break;
    const double sn = fma(sj, c1, cj * s1);
    cj = fma(cj, c1, -(sj * s1));
    sj = sn;
    if ((j & 31) == 0) sincos_cf(h * x * (double)(j + 1), sj, cj);  // re-anchor long series
  }
  return result;
}

// The bisection ladder of inverse_cdf (sample_from_cf.jl:123-133) for the trajectories of a tile whose secant failed
// (2 % of them: 5.7 per tile), run by ONE WAVE of the workgroup for all of them at once.
//
// A failed trajectory's ladder is ~15 CDF evaluations in a dependent chain.  Left to its own lane it holds its wave
// for all of them (three waves in four have such a lane); sent to a kernel of its own, densely packed, it was that
// kernel's whole 19 µs behind a prefix scan.  But the abscissae of a bisection are known in advance as a TREE: the
// midpoint of [lo, hi], then the midpoints of its two halves, … — each the same 0.5·(a + b) of the same two ancestors
// the sequential loop would form.  So the 64 lanes of a wave are dealt to the failed trajectories of a batch in groups
// of W = 64 / 32 / 16 / 8 (for 1 / 2 / 3-4 / 5-8 of them; a tile's failures go in batches of eight, batch b to wave
// b mod 4), a group evaluates the W − 1 nodes of the next log2 W levels of its trajectory's tree in ONE turn, and then
// WALKS them with the statements of the sequential loop — its sign test, its exact-zero test, its width test, its
// iteration cap — on the values already there.  Every node's lane knows what the loop would do AT it (which half it
// keeps: the sign of f(lo) never changes in a bisection; whether it stops there), two ballots hand that to the group,
// and the walk is bit tests.  First turn: the two end points and log2 W − 1 levels.  A ladder of 13 midpoints is 3
// turns for a lone trajectory, 5 in a group of eight — each one CDF evaluation long, for the whole batch.  Same
// abscissae, same CDF values, same decisions: the same ∫V, decision word and counters as the sequential loop, bit for
// bit (HH_BK_SERIAL_LADDER builds that loop; tests/test_gpu_bk_forms.py holds the two against each other).
// (What it costs: ~20 µs of the 0.32 ms at 10^6 trajectories, however it is organised — one wave per tile in batches
// of 8, 4 or 2, every wave its own lanes with no barrier in front (tools/variants/bk_ladder_variants.inc), the
// walking wave at raised priority: all within 1 % of each other, profiles/r06_g_bk_ab_ladder_forms.txt, r06_k_*.
// Half of it is the instructions — a turn is one CDF evaluation, ~250 of a tile's 39 000 wave-instructions — the rest
// the three waves that wait.  The four small kernels this replaced cost 37 µs + their boundaries.)
//
// What a failed lane hands over and gets back lies in LDS (by thread), and so do a group's sixteen weighted terms
// (a broadcast read per term instead of sixteen registers per lane: with them in registers, and a lane's own state
// kept across the ladder, the CF kernel took 128 registers instead of 95 — four waves per SIMD instead of five).
// thread of the r-th failed trajectory of the tile (r below their number)
__device__ __forceinline__ uint32_t nth_failed(const LadderShared& sh, uint32_t r) {
#if defined(HH_BK_LADDER_VARIANTS) && HH_BK_LADDER_PER_WAVE  // every wave its OWN failed lanes (bk_ladder_variants.inc)
  {
    const uint32_t wv = threadIdx.x >> 6;
    unsigned long long mw = sh.fail[wv];
    for (uint32_t i = 0; i < r; ++i) mw &= mw - 1ull;
    return wv * 64u + (uint32_t)__ffsll((long long)mw) - 1u;
  }
#endif
  uint32_t w = 0;
#pragma unroll
  for (uint32_t i = 0; i + 1 < (uint32_t)(kTile / 64); ++i) {
    const uint32_t c = (uint32_t)__popcll(sh.fail[i]);
    if (w == i && r >= c) {
      r -= c;
      w = i + 1;
    }
  }
  unsigned long long m = sh.fail[w];
  for (uint32_t i = 0; i < r; ++i) m &= m - 1ull;
  return w * 64u + (uint32_t)__ffsll((long long)m) - 1u;
}

// Called by every lane of a wave, for batch `b0` of the tile's failed trajectories (the n_fail ballots are in sh.fail;
// a barrier lies between their stores and this).  `col`: the caller's column of the term cache — a group reads its
// trajectory's terms from THAT lane's column (same slot).
__device__ __forceinline__ void wave_ladder(const BkArgs& p, const double* coef, const double* col, LadderShared& sh,
                                            uint32_t first, uint32_t n) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const size_t stride = p.cache_stride;
  const uint32_t lg = n == 1u ? 6u : n == 2u ? 5u : n <= 4u ? 4u : 3u;  // log2 of the lanes per group
  const uint32_t W = 1u << lg;
  const uint32_t grp = lane >> lg, node = lane & (W - 1u), base = lane & ~(W - 1u);
  const bool gactive = grp < n;
  const uint32_t owner = gactive ? nth_failed(sh, first + grp) : tid;  // (a thread of the workgroup)
  const int JS = gactive ? sh.j_stop[owner] : 0;
  const double* COL = col + ((int)owner - (int)tid);
  double* T = sh.terms[wave][grp];
  const uint32_t entry = gactive ? sh.stash_of[owner] : 0xffu;
  for (uint32_t j = node; j < (uint32_t)kRegTerms; j += W)  // load_terms(), a term per lane
    T[j] = entry < (uint32_t)kLadderStash ? sh.stash[entry][j] : (int)j < JS ? coef[j + 1u] * COL[(size_t)j * stride] : 0.0;
  lds_fence();
  // (what is read once per turn stays in LDS — h, u — and max_guess is the first `hi`: registers are what this
  // kernel is short of)
  double lo = 0.0, hi = sh.max_guess[owner];  // the state of the sequential loop: the same in every lane of a group
  bool neg0 = false;         // f(lo) < 0: the loop's `fa < 0`, which no step of a bisection changes
  uint32_t iters = 0;
  bool done = !gactive, maxg = false, first_turn = true;
  for (;;) {  // (uniform) one CDF evaluation per lane and turn
    const uint32_t depth = first_turn ? lg - 1u : lg;  // levels of the tree this turn evaluates
    // this lane's abscissa: an end point (first turn), or node `node` of the tree over [lo, hi] in heap order —
    // root 1, children 2k (left half: hi = mid) and 2k + 1 (right half: lo = mid); [a, b] is what the loop holds
    // when it comes to this node
    double x = first_turn && node != 0u ? hi : 0.0, a = lo, b = hi;
    bool ev = first_turn && (node == 0u || node == (W >> 1));
    const bool in_tree = !ev && node >= 1u && node < (1u << depth);
    if (in_tree) {
      ev = true;
      for (int i = 30 - __clz((int)node); i >= 0; --i) {
        const double m = 0.5 * (a + b);
        if ((node >> i) & 1u) a = m; else b = m;
      }
      x = 0.5 * (a + b);
    }
    double f = 0.0;
    if (ev && !done) {
      double unused = 0.0;
      f = cdf_cached(T, COL, coef, stride, JS, sh.h[owner], x, unused) - sh.u[owner];
    }
    if (first_turn) {  // (uniform) the two end points: sample_from_cf.jl:123-126
      const double f0 = __shfl(f, (int)base, 64), f1 = __shfl(f, (int)(base + (W >> 1)), 64);
      if (!done) {
        neg0 = f0 < 0.0;
        if (f0 * f1 > 0.0) maxg = done = true;
      }
    }
    // what the loop does AT this node (sample_from_cf.jl:127-133): keeps the right half when the sign is f(lo)'s,
    // stops on an exact zero or when the interval it keeps is no wider than atol
    const bool right = (f < 0.0) == neg0;
    const double lo_k = f == 0.0 ? x : right ? x : a, hi_k = f == 0.0 ? x : right ? b : x;
    const unsigned long long m_right = __ballot(in_tree && right);
    const unsigned long long m_stop = __ballot(in_tree && (f == 0.0 || hi_k - lo_k <= p.atol));
    uint32_t k = 1u, last = 0u;  // the walk
    for (uint32_t lvl = 0; lvl < depth; ++lvl) {
      if (!done) {
        if ((int)iters >= p.bisect_maxiter) {
          done = true;
        } else {
          ++iters;
          last = k;
          const uint32_t at = base + k;
          if ((m_stop >> at) & 1ull) done = true;
          k = 2u * k + (uint32_t)((m_right >> at) & 1ull);
        }
      }
    }
    const double lo_n = __shfl(lo_k, (int)(base + last), 64), hi_n = __shfl(hi_k, (int)(base + last), 64);
    if (last != 0u) {
      lo = lo_n;
      hi = hi_n;
    }
    if (!done && (int)iters >= p.bisect_maxiter) done = true;
    first_turn = false;
    if (__ballot(!done) == 0ull) break;
  }
  if (gactive && node == 0u) {
    sh.iv[owner] = maxg ? hi : 0.5 * (lo + hi);  // (no step was made when max_guess stands: hi still holds it)
    sh.res[owner] = iters | (maxg ? 1u << 31 : 0u);
  }
}

// The same ladder by the failed lane alone, as the reference writes it (a test build, HH_BK_SERIAL_LADDER)
__device__ __forceinline__ void lane_ladder(const BkArgs& p, const double* coef, const double* col, LadderShared& sh) {
  const uint32_t tid = threadIdx.x;
  const int j_stop = sh.j_stop[tid];
  const double h = sh.h[tid], u = sh.u[tid], max_guess = sh.max_guess[tid];
  double t[kRegTerms];
  load_terms(col, p.cache_stride, j_stop, t);
  double unused = 0.0;
  auto cdf = [&](double x) { return cdf_cached(t, col, coef, p.cache_stride, j_stop, h, x, unused); };
  double fa = cdf(0.0) - u;
  const double fb = cdf(max_guess) - u;
  if (fa * fb > 0.0) {  // sample_from_cf.jl:124-126
    sh.iv[tid] = max_guess;
    sh.res[tid] = 1u << 31;
    return;
  }
  double lo_x = 0.0, hi_x = max_guess;
  uint32_t iters = 0;
  for (int i = 0; i < p.bisect_maxiter; ++i) {
    const double mid = 0.5 * (lo_x + hi_x);
    const double fm = cdf(mid) - u;
    ++iters;
    if (fm == 0.0) {
      lo_x = hi_x = mid;
      break;
    }
    if ((fm < 0.0) == (fa < 0.0)) {
      lo_x = mid;
      fa = fm;
    } else {
      hi_x = mid;
    }
    if (hi_x - lo_x <= p.atol) break;
  }
  sh.iv[tid] = 0.5 * (lo_x + hi_x);
  sh.res[tid] = iters;
}

// Inversion phase: the secant iteration of inverse_cdf on the cached series, for the trajectories it fails on the
// bisection ladder (wave_ladder: one wave for the tile's failures), then log S_T and the payoff.  A trajectory whose series did not fit the cache is
// flagged in a per-wave ballot and left to bk_tail_kernel.  A lane's payoff stays in its own place of the
// workgroup's sum, so the record is bit-reproducible.  Called by every thread of the workgroup.
template <int ORD>
__device__ __forceinline__ void invert_phase(const BkArgs& p, const double* coef, uint32_t tile, uint32_t tid, uint64_t path,
                                             bool live, const double* col, LadderShared& sh, double h, int j_stop) {
  double acc[6] = {0, 0, 0, 0, 0, 0};  // Σp, Σp², newton_fail, bisect, maxguess, cf_terms
  bool failed = false, too_long = false;
  uint32_t dec = 0;
  double root = 0.0;  // the trajectory's ∫V: the secant's, or the ladder's behind it
  // log S_T, the payoff and the trajectory's decision word, once its ∫V is known.  (ONE call, behind the ladder: with a
  // call for the trajectories the secant finished and another for the ladder's, three waves in four ran the payoff
  // code twice — 0.6 % of the chain, profiles/r06_v_bk_secant_hand_over.txt)
  auto finish = [&](double IV) {
    const uint64_t src = pair_of<ORD>(p, path);
    p.diag[path] = dec;
    const double V0 = p.in_var ? p.in_var[src] : p.V0;
    const double logS0 = p.in_spot ? fm::log(p.in_spot[src]) : p.logS0;  // heston.jl:84, :289
    const double pay = bk_finish(p, logS0, V0, p.draws[3 * p.draw_stride + src], p.draws[src], IV, src);
    acc[0] = pay;
    acc[1] = pay * pay;
  };
  if (live) {
    p.diag[p.draw_stride + path] = (uint32_t)j_stop;
    if (j_stop == 0) {
      too_long = true;
      p.diag[path] = kDecLongSeries;
    } else {
      const double u = p.draws[p.draw_stride + pair_of<ORD>(p, path)];
      double t[kRegTerms];
      load_terms(col, p.cache_stride, j_stop, t);
      double n_terms = 0.0;
      failed = !secant_inverse(
          [&](double x) { return cdf_cached(t, col, coef, p.cache_stride, j_stop, h, x, n_terms); }, u, sh.guess[tid],
          p.atol, p.newton_maxiter, root, dec);
      acc[5] = n_terms;
      if (failed) {  // to the ladder: what the wave needs of this trajectory (max_guess is there already)
        sh.h[tid] = h;
        sh.u[tid] = u;
        sh.j_stop[tid] = j_stop;
        const uint32_t entry = atomicAdd(&sh.stash_n, 1u);
        sh.stash_of[tid] = (unsigned char)(entry < (uint32_t)kLadderStash ? entry : 0xffu);
        if (entry < (uint32_t)kLadderStash) {
#pragma unroll
          for (int j = 0; j < kRegTerms; ++j) sh.stash[entry][j] = t[j];
        }
      }
    }
  }
  const unsigned long long m_fail = __ballot(failed);
  if ((tid & 63u) == 0u) sh.fail[tid >> 6] = m_fail;
  // what a failed lane takes back from the ladder: its ∫V, its counters, the rest of its decision word
  auto after_ladder = [&]() {
    const uint32_t res = sh.res[tid], iters = res & 0x7fffffffu;
    acc[2] = 1.0;
    // the CDF evaluations the sequential ladder makes: its two end points (one below zero returns before it
    // counts: cdf_cached) and its midpoints, j_stop terms each
    acc[5] += (double)sh.j_stop[tid] * (double)(1u + (sh.max_guess[tid] < 0.0 ? 0u : 1u) + iters);
    if (res >> 31) {
      acc[4] = 1.0;
      dec |= kDecMaxGuess;
    } else {
      acc[3] = 1.0;
      dec |= kDecBisect | ((iters & 0xffu) << kDecItersShift);
    }
    root = sh.iv[tid];
  };
#ifdef HH_BK_LADDER_VARIANTS  // A/B builds (tools/variants/bk_ladder_variants.inc): other organisations of the same walk
#include "bk_ladder_variants.inc"
#else
  if (__syncthreads_or(failed)) {  // (uniform) nearly every tile
#if HH_BK_SERIAL_LADDER
    if (failed) lane_ladder(p, coef, col, sh);
#else
    uint32_t n_fail = 0;
#pragma unroll
    for (int w = 0; w < kTile / 64; ++w) n_fail += (uint32_t)__popcll(sh.fail[w]);
    for (uint32_t b0 = (tid >> 6) * 8u; b0 < n_fail; b0 += (uint32_t)(kTile / 64) * 8u)  // (wave-uniform) batch b to wave b mod 4
      wave_ladder(p, coef, col, sh, b0, n_fail - b0 < 8u ? n_fail - b0 : 8u);
#endif
    __syncthreads();
    if (failed) after_ladder();
  }
#endif
  if (live && !too_long) finish(root);
  const unsigned long long m_long = __ballot(too_long);
  if ((tid & 63) == 0) p.long_mask[(size_t)tile * (kTile / 64) + (tid >> 6)] = m_long;
  bk_store_record(acc, p.records + (size_t)tile * kRecStride, p.counters, too_long ? 1.0 : 0.0);
}

// A workgroup's slot of the term cache.
//  * Chains of at most kSlots tiles: slot = tile, nothing shared, nothing to take.
//  * Longer chains: the slots are REUSED, and a reused column must stay inside one XCD — the L2s of the
//    eight XCDs are write-back and not coherent with each other, so a dirty line of a column's previous
//    owner on another XCD could be written back over the new owner's terms (seen: 1 trajectory in 10^6
//    moved when it was allowed).  Each XCD therefore owns kSlots / 8 slots (its 32 CUs hold 5 workgroups
//    each = 160 of the 192), as a bitmap of three 64-bit words in a 128-byte line of its own.  Wave 0 reads the XCD's
//    words, lane 0 claims a free bit with an atomic OR (which returns the word's current state when
//    somebody else was faster), the slot goes to the other waves through LDS and is given back (atomic
//    AND) at the end.  A slot's holder never waits for anything, so a workgroup that finds every slot
//    taken only waits for another one to finish: no deadlock however the hardware places workgroups.
constexpr int kXcds = 8, kSlotsPerXcd = kSlots / kXcds, kSlotLineWords = 16;  // 128-byte line per XCD
static_assert(kSlotsPerXcd % 64 == 0 && kSlotsPerXcd / 64 <= kSlotLineWords, "whole bitmap words, one line per XCD");
__device__ __forceinline__ uint32_t take_slot(const BkArgs& p, uint32_t own) {
  if (p.static_slots) return own;  // uniform
  __shared__ uint32_t slot_sh;
  if (threadIdx.x == 0) {
    // HW_REG_XCC_ID (hwreg 20), bits 3:0: the XCD this workgroup runs on
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) % (uint32_t)kXcds;
    unsigned long long* bm = reinterpret_cast<unsigned long long*>(p.slot_busy) + (size_t)xcc * kSlotLineWords;
    constexpr uint32_t kWords = kSlotsPerXcd / 64;
    // Start at the bit this workgroup would own if the slots were dealt out in order (workgroups go to the
    // XCDs round-robin by their index): the first kSlotsPerXcd workgroups of an XCD then start on
    // kSlotsPerXcd different bits — picking the LOWEST free bit instead makes them all claim bit 0, then
    // bit 1, …: O(n²) serialised atomics on one line when the grid starts (0.1 ms at 128 per XCD).
    const uint32_t h = (own / (uint32_t)kXcds) % (uint32_t)kSlotsPerXcd;
    uint32_t wi = h >> 6, rot = h & 63u, slot = 0;
    unsigned long long wv = __hip_atomic_load(bm + wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
      if (~wv == 0ull) {  // this word is full: look at the next one
        wi = wi + 1 == kWords ? 0u : wi + 1;
        rot = 0;
        __builtin_amdgcn_s_sleep(2);
        wv = __hip_atomic_load(bm + wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      const unsigned long long fr = ~wv;                                   // free bits
      const unsigned long long rr = rot ? (fr >> rot) | (fr << (64u - rot)) : fr;  // … seen from bit `rot`
      const uint32_t bit = (rot + (uint32_t)__ffsll((long long)rr) - 1u) & 63u;
      const unsigned long long old = atomicOr(bm + wi, 1ull << bit);
      if (((old >> bit) & 1ull) == 0ull) {
        slot = xcc * (uint32_t)kSlotsPerXcd + wi * 64u + bit;
        break;
      }
      wv = old | (1ull << bit);
    }
    slot_sh = slot;
  }
  __syncthreads();
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)slot_sh);  // (uniform, and the compiler may know it)
}
__device__ __forceinline__ void give_slot(const BkArgs& p, uint32_t slot) {
  if (p.static_slots) return;
  __syncthreads();  // every lane is done with its column
  if (threadIdx.x == 0) {
    const uint32_t xcc = slot / (uint32_t)kSlotsPerXcd, r = slot % (uint32_t)kSlotsPerXcd;
    atomicAnd(reinterpret_cast<unsigned long long*>(p.slot_busy) + (size_t)xcc * kSlotLineWords + (r >> 6),
              ~(1ull << (r & 63u)));
  }
}

// The CF kernel: series phase then inversion phase per trajectory, one tile per workgroup.
// The inversion re-reads the terms the lane has just written (L2 hits) while other waves of the CU are
// still in their series phase, so its memory latency — 2/3 of a stand-alone inversion kernel's time —
// hides behind their arithmetic; the series state is dead by then, so the register count is the series
// phase's.
// (tabs: the Bessel tables, a `const __restrict__` kernel argument of its own so that the compiler
// knows them read-only and un-aliased: their uniform-index reads are then scalar loads, as they were
// from the argument block; staged in LDS instead, the per-order scalars and the coefficients in
// flight sit in VGPRs and the kernel needs 192)
// (one tile per workgroup, NOT a grid-stride loop over the tiles: with a loop around this body the compiler
// hoists loop-invariant table values into 215-247 registers — measured — and halves the occupancy)
// (occupancy: at its 95 registers five workgroups are resident per CU; capped at 4 / 3 / 2 by an LDS pad the chain
// takes +2 % / +12 % / +41 % at 10^6 trajectories and +4.5 % / +17 % / +53 % at 10^7 — profiles/r06_o_bk_occupancy.txt;
// at 6 waves per SIMD it would have to spill 66 registers: 1.5 x the time, round 4)
#ifndef HH_BK_CF_WAVES
#define HH_BK_CF_WAVES 0
#endif
#if HH_BK_CF_WAVES
#define HH_BK_CF_OCC __attribute__((amdgpu_waves_per_eu(HH_BK_CF_WAVES, HH_BK_CF_WAVES)))
#else
#define HH_BK_CF_OCC
#endif
// (DRAW: the trajectory's three draws are made (1: GENERATE) or read from the caller's buffer (2: REPLAY) here — the
// one-shot law: a tile needs only its own draws, so a launch in front bought nothing but its own fill and drain.
// 0: they are where a kernel in front left them — the variance kernel of a grid)
template <int ORD, int DRAW = 0>
__global__ __launch_bounds__(kTile) HH_BK_CF_OCC void bk_cf_kernel(const BkArgs p, const BkTables* __restrict__ tabs) {
  const uint32_t tile = blockIdx.x, tid = threadIdx.x;
  const uint64_t path = (uint64_t)tile * kTile + tid;
  const bool live = path < p.n_paths;
  if (tile == 0 && tid == 0) __builtin_memcpy(p.args_dev, &p, sizeof(BkArgs));  // for bk_tail_kernel
#if defined(HH_BK_TILE_STAMPS) && HH_BK_TILE_STAMPS  // diagnostic build (tools/bk_tile_timeline.py): when a tile ran, and where
  const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
#endif
  // The tiles of the LAST ROUND — the last occupants of the chip's workgroup slots — issue ahead of the older waves
  // beside them.  The hardware issues oldest-first, so at the end of a chain the final five waves of a SIMD finish one
  // after the other, the last of them alone at a quarter of the issue rate: a drain of 100 µs behind the last start
  // (tools/bk_tile_timeline.py).  Raised, they take 75 instead of 96 µs each while the round before them gives way:
  // -1 … -3.5 % of the chain at every size from 1.1 to 6 rounds, -0.3 % at 30 (profiles/r06_ab_bk_last_round_priority.txt;
  // two rounds raised: +6 %, half a round: a third of the gain, priorities graded by start order or falling with a
  // tile's progress: no better; EVERY tile's priority falling with its progress: +9 % — oldest-first is right until the end).
  if (tile >= p.drain_tile) __builtin_amdgcn_s_setprio(3);
  if (DRAW != 0 && live) {
    double Z, u, VT;
    if constexpr (DRAW == 2) {
      VT = fmax(p.replay[path], 0x1p-1000);
      u = p.replay[p.n_paths + path];
      Z = p.replay[2 * p.n_paths + path];
    } else {
      draw_transition(p, p.seeds[0], p.path_offset + path, p.V0, Z, u, VT);
    }
    store_draws(p, path, Z, u, VT);
  }
  const BesselTable* bt = tabs->t;
  const uint32_t slot = take_slot(p, tile);
  double* col = p.phi_cache + (size_t)slot * kTile + tid;
  __shared__ LadderShared sh;
  if (tid == 0) sh.stash_n = 0u;  // (a barrier lies between this and the first failed lane: the series phase is
  __syncthreads();                //  tens of microseconds long, but say so to the compiler and the hardware)
  double h = 0.0;
  int j_stop = 0;
  // (another reading of the reference's root searches was asked for: every trajectory runs whole in the tail kernel)
  if (live && (p.root_form | p.bracket_form | p.caps) == 0) series_phase<ORD>(p, bt, path, col, p.cache_stride, sh, h, j_stop);
  invert_phase<ORD>(p, tabs->coef, tile, tid, path, live, col, sh, h, j_stop);
  give_slot(p, slot);
#if defined(HH_BK_TILE_STAMPS) && HH_BK_TILE_STAMPS
  // (in place of the series lengths of the tile's last three trajectories: start, end in 10 ns ticks, XCD | CU id)
  __syncthreads();
  if (tid == 0 && (uint64_t)(tile + 1) * kTile <= p.n_paths) {
    uint32_t* out = p.diag + p.draw_stride + (size_t)tile * kTile + (kTile - 3);
    out[0] = (uint32_t)stamp0;
    out[1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    out[2] = ((uint32_t)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) << 16) |  // XCC_ID
             ((uint32_t)__builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4) & 0xffffu);  // HW_ID: wave, simd, cu, sh, se
  }
#endif
}

// Tail kernel, kHeavyGrid workgroups behind the CF kernel.
//
// (a) Trajectories whose series did not fit the cache (cf_tol far below the reference's default) run whole here —
// secant, then the ladder if it fails — evaluating the terms beyond the cache on every use, as the reference does
// with all of them.  The CF kernel counted them (BkArgs::counters[0]); with the reference's controls the count is
// zero and this part is one scalar load.  Otherwise workgroup b takes the tiles b, b + kHeavyGrid, … that have such
// trajectories (found 256 tiles at a time), a lane its own trajectory of the tile: a fixed assignment, so the
// workgroup's record is reproducible (and bk_refinish_kernel can rebuild it).
// Every CDF evaluation of the root search — the two starting points of the secant, its iterates, the two
// ends of the ladder, the bisection's midpoints — goes through ONE call of cdf_from_cf below, the search
// itself being a small state machine around it (the same statements, in the same order, as
// secant_inverse and lane_ladder).  With the CDF inlined at four places the kernel took
// 248 registers and kept the trajectory's state in scratch (the calls that did not inline took it by
// reference); with one place it is ~160 registers and nothing in scratch.  (Capped at 128 the allocator
// spills 8-30 registers whatever part of the search state is parked in LDS: the pressure is inside the one
// CF evaluation, which this kernel runs with its full series every time.)
//
// (b) The records of the chain — the CF kernel's tiles, and this kernel's workgroups when (a) had work — are added
// into `accum` by the first kRecStride workgroups, one accumulator slot each, in reduce_records_kernel's order
// (sum_slot).  When (a) had work they wait for every workgroup's record first: kHeavyGrid workgroups are resident
// together on any device this library runs on, each arrives once, none waits before it has arrived.
// (Also the home of the OTHER READINGS of the reference's two `find_zero` calls — hh_config.bk_root_form,
// bk_bracket_form, bk_caps; include/hedgehog_mc.h has the table: Roots' Order2 as a Steffensen step guarded by a secant
// step, its bisection over the bit patterns of the ends, the `maxeval` keyword ignored.  A chain run with any of them
// sends EVERY trajectory here — they are a parity seam, to be decided by one Julia run, not a pricing path — so the
// CF kernel carries none of it.)
__device__ __forceinline__ double roots_middle(double a, double b) {  // Roots.jl `_middle` for 0 <= a < b, finite
  const unsigned long long ia = (unsigned long long)__double_as_longlong(a), ib = (unsigned long long)__double_as_longlong(b);
  return __longlong_as_double((long long)((ia + ib) >> 1));
}
__device__ __forceinline__ int sign_of(double x) { return (x > 0.0) - (x < 0.0); }

__device__ __forceinline__ void tail_whole_trajectory(const BkArgs& p, const BesselTable* bt, const double* coef,
                                                      uint64_t path, double (&acc)[6]) {
  PathSetup s;
  bk_setup(p, bt, path, s);
  double n_terms = 0.0;
  enum Stage { kSecantFirst, kSecant, kSteffProbe, kLadderLo, kLadderHi, kBisect };
  const bool order2 = p.root_form == HH_BK_ROOT_ORDER2, roots_bisect = p.bracket_form == HH_BK_BRACKET_ROOTS;
  const bool own_caps = p.caps == HH_BK_CAPS_ROOTS_DEFAULT;
  // evaluations (secant) / steps (Order2) of the first search, iterations of the bisection
  const int cap_first = own_caps ? (order2 ? kRootsMaxIters : 2 + kRootsMaxIters) : p.newton_maxiter;
  const int cap_bisect = own_caps ? 4096 : p.bisect_maxiter;
  // secant_inverse: Order2 restated as the secant iteration from (x0 + dx, x0), dx = h + |x0| h², h = eps^(1/3).
  // The secant's (x0, f0), (x1, f1) and the ladder's (lo, f(lo)), (hi, f(hi)) share their registers: xa, fa, xb, fb
  const double hs = 6.0554544523933395e-06, eps = 0x1p-52;
  double xb = s.initial_guess;                   // secant: x1            ladder: hi
  double xa = xb + hs + fabs(xb) * hs * hs;      // secant: x0            ladder: lo
  double fa = 0.0, fb = 0.0;                     // secant: f0, f1        ladder: f(lo), f(hi)
  uint32_t dec = (p.root_form | p.bracket_form | p.caps) ? 0u : kDecLongSeries, evals = 1, iters = 0;
  int stage = kSecantFirst, steps = 0, sgn = 0;
  bool out_set = false;  // the bisection met an exact zero: x is the answer
  double x = xa;
  for (;;) {
    const double f = cdf_from_cf(p, bt, coef, s.cf, x, s.h, s.cache, n_terms) - s.u;
    if (stage == kSecantFirst) {
      fa = f;
      x = xb;
      stage = kSecant;
      continue;
    }
    if (stage == kSecant || stage == kSteffProbe) {
      ++evals;
      bool ok = false, go_on = false;
      if (stage == kSteffProbe) {  // f = f(x1 - sgn f1): the Steffensen step itself
        const double d = f != fb ? -(double)sgn * fb * fb / (f - fb) : __builtin_inf();
        if (isfinite(d)) {
          xa = xb;
          fa = fb;
          x = xb = xb - d;
          ++steps;
          stage = kSecant;
          continue;
        }
        x = xb;  // the search ends where it stood
      } else if (!order2) {
        fb = f;
        ok = fabs(fb) <= p.atol;
        if (!ok && !((int)evals >= cap_first || fb == fa)) {
          const double x2 = xb - fb * (xb - xa) / (fb - fa);
          if (isfinite(x2)) {
            xa = xb;
            fa = fb;
            x = xb = x2;
            go_on = true;
          }
        }
      } else {  // Roots.Order2: assess_convergence, then a guarded step
        fb = f;
        const double tol = fmax(p.atol, fabs(xb) * 4.0 * eps);
        if (isfinite(xb) && isfinite(fb)) {
          if (fabs(fb) <= tol) {
            ok = true;
          } else if (fabs(xb - xa) <= fmax(eps, fmax(fabs(xb), fabs(xa)) * eps)) {
            ok = fabs(fb) <= cbrt(tol);
          } else if (steps < cap_first) {
            if (1000.0 * fabs(fb) > fmax(1.0, fabs(xb))) {  // f is large: a secant step
              const double d = fb != fa ? fb * (xb - xa) / (fb - fa) : __builtin_inf();
              if (isfinite(d)) {
                xa = xb;
                fa = fb;
                x = xb = xb - d;
                ++steps;
                go_on = true;
              }
            } else {  // a Steffensen step: first the probe at x1 - sgn·f1
              sgn = xb != xa ? sign_of((fb - fa) / (xb - xa)) : 0;
              x = xb - (double)sgn * fb;
              stage = kSteffProbe;
              go_on = true;
            }
          }
        }
      }
      if (go_on) continue;
      dec |= evals < 0xffu ? evals : 0xffu;
      x = xb;                            // the first search's answer, should it stand
      if (ok && !(xb < 0.0 || fabs(fb) > p.atol)) break;
      acc[2] += 1.0;  // the fall-back ladder (sample_from_cf.jl:123-133)
      x = xa = 0.0;
      xb = s.max_guess;
      stage = kLadderLo;
      continue;
    }
    if (stage == kLadderLo) {
      fa = f;
      x = xb;
      stage = kLadderHi;
      continue;
    }
    if (stage == kLadderHi) {
      fb = f;
      if (fa * fb > 0.0) {
        acc[4] += 1.0;
        dec |= kDecMaxGuess;  // x = max_guess (sample_from_cf.jl:124-126)
        break;
      }
      acc[3] += 1.0;
      dec |= kDecBisect;
      x = roots_bisect ? roots_middle(xa, xb) : 0.5 * (xa + xb);
      if (cap_bisect <= 0 || (roots_bisect && !(xa < x && x < xb))) break;
      stage = kBisect;
      continue;
    }
    // kBisect: f is the CDF residual at the midpoint x
    ++iters;
    bool stop;
    if (roots_bisect) {  // Roots.Bisection: to the last bit, the end with the smaller residual
      if (f == 0.0) {
        out_set = true;
      } else if (sign_of(fa) * sign_of(f) < 0) {
        xb = x;
        fb = f;
      } else {
        xa = x;
        fa = f;
      }
      const double mid = roots_middle(xa, xb);
      stop = out_set || (int)iters >= cap_bisect || !(xa < mid && mid < xb);
      if (!out_set) x = stop ? mid : mid;
    } else {
      if (f == 0.0) {
        xa = xb = x;
      } else if ((f < 0.0) == (fa < 0.0)) {
        xa = x;
        fa = f;
      } else {
        xb = x;
      }
      x = 0.5 * (xa + xb);
      stop = f == 0.0 || xb - xa <= p.atol || (int)iters >= cap_bisect;
    }
    if (stop) {
      dec |= (iters & 0xffu) << kDecItersShift;
      break;
    }
  }
  if (roots_bisect && (dec & kDecBisect) && !out_set) x = fabs(fa) < fabs(fb) ? xa : xb;
  const double IV = x;
  p.diag[path] = dec;
  p.diag[p.draw_stride + path] = (uint32_t)s.cache.j_stop;
  acc[5] += n_terms;
  const double pay = bk_finish_path(p, path, IV);
  acc[0] += pay;
  acc[1] = fma(pay, pay, acc[1]);
}

// The tiles of tail workgroup b that hold a trajectory for it, in increasing order: fn(tile) for each.  256 tiles
// are looked at per turn (a lane reads the four ballots of ONE tile), the hits handed on through LDS.
template <class Fn>
__device__ __forceinline__ void for_long_tiles(const unsigned long long* __restrict__ long_mask, uint32_t n_tiles,
                                               uint32_t b, Fn&& fn) {
  __shared__ unsigned char hit[kTile];
  for (uint32_t t0 = b; t0 < n_tiles; t0 += (uint32_t)kHeavyGrid * kTile) {  // (uniform)
    const uint32_t t = t0 + threadIdx.x * (uint32_t)kHeavyGrid;
    unsigned long long any = 0ull;
    if (t < n_tiles) {
#pragma unroll
      for (int w = 0; w < kTile / 64; ++w) any |= long_mask[(size_t)t * (kTile / 64) + w];
    }
    __syncthreads();  // the turn before is done with `hit`
    hit[threadIdx.x] = any != 0ull;
    __syncthreads();
    for (uint32_t i = 0; i < (uint32_t)kTile; ++i)
      if (hit[i]) fn(t0 + i * (uint32_t)kHeavyGrid);  // (uniform)
  }
}

__global__ __launch_bounds__(kTile) void bk_tail_kernel(const BkArgs* __restrict__ args, const BkTables* __restrict__ tabs,
                                                        uint32_t n_tiles, double n_paths, double* __restrict__ accum,
                                                        uint32_t* __restrict__ cnt, const double* __restrict__ records) {
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  __shared__ double sm[257];
  // The common case first, in ONE round trip: the count of too-long trajectories and — workgroup b < 16: slot b of —
  // the CF tiles' records are asked for together (`cnt`, `records` are kernel arguments: no load of the argument
  // block in front); the count is looked at when the sums are there.
  const uint32_t n_long_v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double out = 0.0;
  if (b < (uint32_t)kRecStride) out = sum_slot(records, n_tiles, (int)b, sm);
  const uint32_t n_long = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_long_v);  // (uniform)
  uint32_t n_rec = n_tiles;
  if (n_long != 0u) {
    const BkArgs& p = *args;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for_long_tiles(p.long_mask, n_tiles, b, [&](uint32_t tile) {
      const uint64_t path = (uint64_t)tile * kTile + tid;
      if ((p.long_mask[(size_t)tile * (kTile / 64) + (tid >> 6)] >> (tid & 63u)) & 1ull)
        tail_whole_trajectory(p, tabs->t, tabs->coef, path, acc);
    });
    bk_store_record(acc, p.records + (size_t)(n_tiles + b) * kRecStride);
    // every workgroup has read the count and left its record before a reducer goes on: all arrive, the reducers
    // wait.  (Plain stores, an agent-scope release behind the workgroup's barrier, an agent-scope acquire behind
    // the wait: MI355X_MICROARCH.md, inter-workgroup visibility.  The path of a tolerance nobody prices with.)
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b >= (uint32_t)kRecStride) return;
    if (tid == 0) {
      while (__hip_atomic_load(cnt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)kHeavyGrid)
        __builtin_amdgcn_s_sleep(8);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the last reducer through leaves the line as the next chain expects it
      if (__hip_atomic_fetch_add(cnt + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)kRecStride - 1u) {
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cnt + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cnt + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    n_rec = n_tiles + (uint32_t)kHeavyGrid;
    out = sum_slot(records, n_rec, (int)b, sm);  // again, with the tail workgroups' records behind the tiles'
  } else if (b >= (uint32_t)kRecStride) {
    return;
  }
  if (tid == 0) {
    accum[b] = b == (uint32_t)HH_ACC_NPATHS ? n_paths : out;
    if (b == 0u) cnt[1] = n_rec;
  }
}

// ---- a second model on a chain that has run ------------------------------------------------------------------------
// solve(GreekProblem, FiniteDifference) on a Broadie–Kaya problem is two or three full solves on the same seeds
// (greeks_problem.jl:279-329, 360-422) — and when the bump is of the spot, the rate, ρ or the strike, nothing the
// variance process sees has moved: the same V_T, the same characteristic function, the same inversion, the same ∫V for
// every trajectory.  The chain of the first model leaves ∫V per trajectory (BkArgs::iv_keep); this kernel finishes
// another model from it: log S_T, payoff, and the SAME records — workgroup b of the two kernels of the chain
// (CF tiles | tail workgroups) sums the payoffs of the same trajectories in the same order through the same
// tree (bk_store_record), so the sums are those of a chain of its own, bit for bit; the counters, which are the
// chain's, are copied from the first model's records.
__global__ __launch_bounds__(kTile) void bk_refinish_kernel(const BkArgs p, const double* __restrict__ rec0, uint32_t n_tiles) {
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  if (b >= p.counters[1]) return;  // (uniform) the chain's tail kernel had no trajectories: no records of it
  double acc[6] = {0, 0, 0, 0, 0, 0};
  if (b < n_tiles) {  // a tile of the CF kernel: every trajectory but the too-long ones
    const uint64_t path = (uint64_t)b * kTile + tid;
    const bool left = (p.long_mask[(size_t)b * (kTile / 64) + (tid >> 6)] >> (tid & 63u)) & 1ull;  // finished by the tail kernel
    if (path < p.n_paths && !left) {
      const double pay = bk_finish_path(p, path, p.iv_keep[path]);
      acc[0] = pay;
      acc[1] = pay * pay;
    }
  } else {  // a workgroup of the tail kernel: its tiles, in its order
    for_long_tiles(p.long_mask, n_tiles, b - n_tiles, [&](uint32_t tile) {
      const uint64_t path = (uint64_t)tile * kTile + tid;
      if ((p.long_mask[(size_t)tile * (kTile / 64) + (tid >> 6)] >> (tid & 63u)) & 1ull) {
        const double pay = bk_finish_path(p, path, p.iv_keep[path]);
        acc[0] += pay;
        acc[1] = fma(pay, pay, acc[1]);
      }
    });
  }
  double* rec = p.records + (size_t)b * kRecStride;
  bk_store_record(acc, rec);
  if (tid == 0) {  // (the thread that wrote the record) the chain's counters
    const double* r0 = rec0 + (size_t)b * kRecStride;
    rec[HH_ACC_BK_NEWTON_FAIL] = r0[HH_ACC_BK_NEWTON_FAIL];
    rec[HH_ACC_BK_BISECT] = r0[HH_ACC_BK_BISECT];
    rec[HH_ACC_BK_MAXGUESS] = r0[HH_ACC_BK_MAXGUESS];
    rec[HH_ACC_BK_CF_TERMS] = r0[HH_ACC_BK_CF_TERMS];
  }
}

// The host-made tables into device memory: ONE lane, constant indices (a lane-indexed read of the
// by-value argument would again send the block through scratch), the compiler batches the scalar loads.
__global__ __launch_bounds__(64) void bk_tables_kernel(const BkBessel t, BkTables* __restrict__ dst, double kappa,
                                                        double sigma2, double T, double eta_k, double zeta_k) {
  for (int j = (int)threadIdx.x; j <= kCoefTerms; j += 64) dst->coef[j] = kTwoOverPi / (double)j;  // ([0] is not read)
  if (threadIdx.x != 0) return;
  dst->zero = chf_zero(kappa, sigma2, T, eta_k, zeta_k);
  const double* src = reinterpret_cast<const double*>(&t);
  double* out = reinterpret_cast<double*>(dst->t);
  static_assert(sizeof(BkBessel) == sizeof(dst->t), "the Bessel tables of BkTables");
#pragma unroll
  for (size_t i = 0; i < sizeof(BkBessel) / sizeof(double); ++i) out[i] = src[i];
}

__global__ __launch_bounds__(256) void fill_rows_kernel(double* __restrict__ spot0,
                                                        double* __restrict__ var0, uint64_t n,
                                                        double S0, double V0) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    spot0[i] = S0;
    var0[i] = V0;
  }
}

// ---- a grid chain's pairs in the order of their Bessel arguments -------------------------------------------------
// A date of a grid is a SHORT transition: the Bessel argument ν_γ ∝ sqrt(V0·V_T)/dt is large and spread over
// orders of magnitude, lanes of one wave sit in different regimes of the Bessel function (power series / Hankel
// expansion) with series of different lengths, and the CF kernel runs at 0.53 active lanes per instruction
// (profiles/r05_d_bk_refill_experiment.txt).  The same draws sorted by V_T run the chain 32 % faster
// (profiles/r05_e_bk_sorted_short_T.txt).  So the pairs of a chain are ordered by a coarse key — exponent and
// three mantissa bits of V0·V_T (grid_order_key, made by the variance kernel), a stable radix sort of (key, pair) —
// and the chain runs its positions in that order, reading and writing each pair where it lies (BkArgs::order).
// Nothing a pair computes depends on its neighbours, and the chain's counters are whole numbers: the grid is the
// same, bit for bit.
// The sort is a counting sort of ONE 8-bit digit, three small kernels in this file (until round 5: a general radix
// sort from a library, three kernels of its own and 42 µs for 2.4·10^6 pairs): a run of kSortRun consecutive pairs per
// WAVE — (1) its 256-bin histogram (LDS atomics), (2) per digit, the exclusive prefix sums of the runs' counts, (3) the
// scatter: a run's pairs in 16 rounds of 64, a lane's place = its digit's base + the run's offset + the pairs of the
// same digit in front of it in the run (the lanes with its digit found by eight ballots, the running count per digit
// in LDS).  Stable: equal keys stay in pair order — the order a stable radix sort of (key, pair) gives.
// (measured, 2.4·10^6 pairs: runs of 2048 the same 31 µs, 512: 47, 256: 86 — the [digit][run] table of counts is what costs)
constexpr int kSortRun = 1024;                           // consecutive pairs per wave
constexpr int kSortWaves = kTile / 64;
static_assert(kSortRun % 64 == 0, "whole rounds of a wave");

__global__ __launch_bounds__(kTile) void grid_sort_count_kernel(const uint8_t* __restrict__ keys, uint32_t n, uint32_t n_runs,
                                                                uint32_t* __restrict__ counts) {
  __shared__ uint32_t hist[kSortWaves][256];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, run = blockIdx.x * kSortWaves + wave;
  for (uint32_t d = lane; d < 256u; d += 64u) hist[wave][d] = 0u;
  __syncthreads();
  if (run < n_runs) {
    const uint32_t base = run * (uint32_t)kSortRun;
    constexpr int kRounds = kSortRun / 64;
    uint32_t key[kRounds];  // all of a lane's keys first: sixteen loads in flight, not sixteen round trips in a row
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const uint32_t i = base + (uint32_t)r * 64u + lane;
      key[r] = i < n ? (uint32_t)keys[i] : 256u;
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r)
      if (key[r] < 256u) atomicAdd(&hist[wave][key[r]], 1u);
  }
  __syncthreads();
  if (run < n_runs)
    for (uint32_t d = lane; d < 256u; d += 64u) counts[(size_t)d * n_runs + run] = hist[wave][d];
}

// workgroup d: counts[d][0 … n_runs) -> their exclusive prefix sums in place, the digit's total in totals[d]
__global__ __launch_bounds__(kTile) void grid_sort_scan_kernel(uint32_t* __restrict__ counts, uint32_t n_runs,
                                                               uint32_t* __restrict__ totals) {
  __shared__ uint32_t wsum[kTile / 64];
  uint32_t* c = counts + (size_t)blockIdx.x * n_runs;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t per = (n_runs + kTile - 1) / kTile;  // a thread's stretch of consecutive runs
  const uint32_t r0 = min(tid * per, n_runs), r1 = min(r0 + per, n_runs);
  uint32_t mine = 0;
  for (uint32_t r = r0; r < r1; ++r) mine += c[r];
  uint32_t inc = mine;  // inclusive scan of the stretch totals inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(inc, off, 64);
    if (lane >= (uint32_t)off) inc += v;
  }
  if (lane == 63u) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < kTile / 64; ++w) {
    if (w < wave) before += wsum[w];
    total += wsum[w];
  }
  uint32_t run_at = before + inc - mine;
  for (uint32_t r = r0; r < r1; ++r) {
    const uint32_t v = c[r];
    c[r] = run_at;
    run_at += v;
  }
  if (tid == 0) totals[blockIdx.x] = total;
}

__global__ __launch_bounds__(kTile) void grid_sort_scatter_kernel(const uint8_t* __restrict__ keys, uint32_t n, uint32_t n_runs,
                                                                  const uint32_t* __restrict__ offs,
                                                                  const uint32_t* __restrict__ totals,
                                                                  uint32_t* __restrict__ order) {
  __shared__ uint32_t at[kSortWaves][256];  // where the next pair of digit d of this wave's run goes
  __shared__ uint32_t wsum[kTile / 64];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, run = blockIdx.x * kSortWaves + wave;
  // the digits' bases: exclusive prefix sums of the 256 totals (thread d: digit d)
  const uint32_t tot = totals[tid];
  uint32_t inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(inc, off, 64);
    if (lane >= (uint32_t)off) inc += v;
  }
  if (lane == 63u) wsum[wave] = inc;
  __syncthreads();
  uint32_t base_d = inc - tot;
#pragma unroll
  for (uint32_t w = 0; w < kTile / 64; ++w)
    if (w < wave) base_d += wsum[w];
#pragma unroll
  for (uint32_t w = 0; w < (uint32_t)kSortWaves; ++w) {
    const uint32_t rw = blockIdx.x * kSortWaves + w;
    at[w][tid] = rw < n_runs ? base_d + offs[(size_t)tid * n_runs + rw] : 0u;
  }
  __syncthreads();
  if (run >= n_runs) return;  // (wave-uniform; no barrier below)
  const uint32_t first = run * (uint32_t)kSortRun;
  constexpr int kRounds = kSortRun / 64;
  uint32_t keyr[kRounds];  // (as in the counting kernel: the loads first)
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const uint32_t i = first + (uint32_t)r * 64u + lane;
    keyr[r] = i < n ? (uint32_t)keys[i] : 256u;
  }
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const uint32_t i = first + (uint32_t)r * 64u + lane;
    const bool valid = keyr[r] < 256u;
    const uint32_t key = keyr[r] & 255u;
    unsigned long long peers = __ballot(valid);  // the lanes of this round with this lane's digit
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool set = (key >> bit) & 1u;
      const unsigned long long bl = __ballot(set);
      peers &= set ? bl : ~bl;
    }
    const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    uint32_t place = 0;
    if (valid && rank == 0u) {  // the first of its digit: takes the places of all of them
      place = __hip_atomic_load(&at[wave][key], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(&at[wave][key], place + (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    place = (uint32_t)__shfl((int)place, valid ? __ffsll((long long)peers) - 1 : (int)lane, 64);
    if (valid) order[place + rank] = i;
  }
}

// below this many pairs the sort costs more than the order saves (round 5, with the library's sort: 2.4·10^5 pairs
// +8 %, 2.4·10^6 -25 %)
constexpr uint64_t kGridOrderMinPairs = 1ull << 20;

}  // namespace

constexpr size_t kSlotBitmapBytes = 8 * 128 + 128;  // one 128-byte line per XCD (kXcds = 8) + the line of BkArgs::counters

// series terms cached per column: term_cache (HH_OPT_BK_TERM_CACHE; 0 = kBkTermCacheDefault).  The columns
// belong to workgroup slots, so the cache does not grow with the ensemble.
static int phi_cache_cap(int term_cache) { return term_cache > 0 ? term_cache : kBkTermCacheDefault; }

// The head of the scratch buffer:
//   ballots of the too-long trajectories [n_tiles][4] | (128-byte boundary) slot bitmaps, 8 lines | counters, 1 line |
//   (256-byte boundary) device copy of the argument block | (256) Bessel tables and ϕ(0) constants
static size_t bk_masks_bytes(size_t n_tiles) { return n_tiles * (kTile / 64) * sizeof(unsigned long long); }
static size_t bk_lines_offset(size_t n_tiles) { return (bk_masks_bytes(n_tiles) + 127) & ~(size_t)127; }
static size_t bk_args_offset(size_t n_tiles) { return (bk_lines_offset(n_tiles) + kSlotBitmapBytes + 255) & ~(size_t)255; }
static size_t bk_tables_offset(size_t n_tiles) { return bk_args_offset(n_tiles) + ((sizeof(BkArgs) + 255) & ~(size_t)255); }
static size_t bk_flags_bytes(size_t n_tiles) { return bk_tables_offset(n_tiles) + ((sizeof(BkTables) + 255) & ~(size_t)255); }
// columns of cached terms: one per lane of a workgroup slot (fewer slots than tiles are never needed)
static size_t bk_cache_columns(size_t n_tiles) {
  const size_t slots = n_tiles < (size_t)kHeavyGrid ? (size_t)kHeavyGrid : n_tiles < (size_t)kSlots ? n_tiles : (size_t)kSlots;
  return slots * kTile;
}

int launch_fill_rows(double* spot0, double* var0, uint64_t n, double S0, double V0, hipStream_t s) {
  hipLaunchKernelGGL(fill_rows_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, spot0,
                     var0, n, S0, V0);
  return (int)hipGetLastError();
}

uint32_t bk_record_count(uint64_t n_paths) {
  return tiles_for(n_paths) + (uint32_t)kHeavyGrid;  // CF tiles | the tail kernel's workgroups
}
// … of which a chain fills the first *bk_live_records(): the tail kernel's workgroups leave records only when
// they had trajectories to run (BkArgs::counters[1], written by the tail kernel)
const uint32_t* bk_live_records(const void* scratch, uint64_t n_paths) {
  const unsigned char* base = static_cast<const unsigned char*>(scratch);
  return reinterpret_cast<const uint32_t*>(base + bk_lines_offset(tiles_for(n_paths)) + 8 * 128) + 1;
}

size_t bk_scratch_bytes(uint64_t n_paths, int term_cache) {
  const size_t n_tiles = tiles_for(n_paths);
  // head | cached series terms [cap][columns] (per workgroup SLOT: independent of n_paths) |
  // per trajectory: draws [4] | ∫V [1] | decision word, series length (2 x uint32)
  return bk_flags_bytes(n_tiles) + bk_cache_columns(n_tiles) * (size_t)phi_cache_cap(term_cache) * sizeof(double) +
         n_tiles * kTile * sizeof(double) * 6;
}

namespace {

struct BkLayout {
  uint32_t n_tiles;
  BkTables* tabs_dev;
};

// the argument block of a chain over n_chain trajectories (or (date, trajectory) pairs) and where its pieces
// of the scratch buffer are
int bk_prepare(const hh_model& m, const hh_config& c, const DevicePtrs& ptr, uint64_t n_chain, BkArgs& a,
               BkLayout& L) {
  a.kappa = m.kappa; a.theta = m.theta; a.sigma = m.sigma; a.sigma2 = m.sigma * m.sigma; a.inv_sigma2 = 1.0 / a.sigma2;
  a.rho = m.rho; a.V0 = m.V0; a.T = m.T; a.logS0 = log(m.S0); a.r = m.r_drift;
  a.strike = m.strike; a.cp = m.cp;
  const double em1 = -expm1(-m.kappa * m.T);  // 1 − e^{−κT}
  a.d = 4.0 * m.kappa * m.theta / a.sigma2;                                   // heston.jl:128
  a.lam_num = 4.0 * m.kappa * exp(-m.kappa * m.T);                            // heston.jl:129
  a.lam_den = a.sigma2 * em1;
  a.cscale = a.sigma2 * em1 / (4.0 * m.kappa);                               // heston.jl:130
  a.nu = 0.5 * a.d - 1.0;                                                    // heston.jl:168
  a.zeta_k = em1 / m.kappa;                                                  // heston.jl:170
  a.eta_k = m.kappa * (1.0 + exp(-m.kappa * m.T)) / em1;                     // heston.jl:171
  a.nuk_factor = 4.0 * m.kappa * exp(-0.5 * m.kappa * m.T) / a.sigma2 / em1;  // heston.jl:172
  if (!(a.d > 0.0) || !std::isfinite(a.d) || !(a.lam_num * m.V0 / a.lam_den >= 0.0))
    return (int)hipErrorInvalidValue;
  a.n_int = a.nu >= 1.0 ? (int)floor(a.nu) : 0;
  a.n_sigma = c.bk_n_sigma > 0.0 ? c.bk_n_sigma : 5.0;
  a.cf_tol = c.bk_cf_tol > 0.0 ? c.bk_cf_tol : 1e-3;
  a.atol = c.bk_atol > 0.0 ? c.bk_atol : 1e-4;
  a.moment_h = c.bk_moment_h > 0.0 ? c.bk_moment_h : 1e-2;
  a.newton_maxiter = c.bk_newton_maxiter > 0 ? c.bk_newton_maxiter : 10;
  a.bisect_maxiter = c.bk_bisect_maxiter > 0 ? c.bk_bisect_maxiter : 100;
  a.root_form = c.bk_root_form;
  a.bracket_form = c.bk_bracket_form;
  a.caps = c.bk_caps;
  a.n_paths = n_chain;
  a.path_offset = c.path_offset;
  a.seeds = ptr.seeds;
  a.terminal = ptr.terminal;
  a.records = ptr.records;
  const uint32_t n_tiles = tiles_for(n_chain);
  const size_t lanes = (size_t)n_tiles * kTile;
  unsigned char* base = reinterpret_cast<unsigned char*>(ptr.bk_scratch);
  a.long_mask = reinterpret_cast<unsigned long long*>(base);
  L.n_tiles = n_tiles;
  a.slot_busy = reinterpret_cast<uint32_t*>(base + bk_lines_offset(n_tiles));  // 128-byte lines
  a.counters = a.slot_busy + 8 * 32;                                            // the line behind the eight slot bitmaps
  a.args_dev = base + bk_args_offset(n_tiles);
  L.tabs_dev = reinterpret_cast<BkTables*>(base + bk_tables_offset(n_tiles));
  a.tabs_dev = L.tabs_dev;
  a.n_tiles = n_tiles;
  a.cache_cap = phi_cache_cap(ptr.bk_term_cache);
  a.phi_cache = reinterpret_cast<double*>(base + bk_flags_bytes(n_tiles));
  a.cache_stride = bk_cache_columns(n_tiles);
  a.static_slots = n_tiles <= (uint32_t)kSlots ? 1u : 0u;
  a.draws = a.phi_cache + a.cache_stride * (size_t)a.cache_cap;
  a.iv_store = a.draws + 4 * lanes;  // ∫V per pair of a grid chain / per trajectory of the one-shot law (iv_keep)
  a.diag = reinterpret_cast<uint32_t*>(a.iv_store + lanes);  // 2 x uint32 per lane
  a.draw_stride = lanes;
  return 0;
}

// the Bessel tables depend on ν alone, the ϕ(0) constants beside them on κ, σ², T: repeated solves of one model
// (and the dates of a grid) find them in place
int bk_tables(const BkArgs& a, const BkLayout& L, const DevicePtrs& ptr, hipStream_t s, bool upload_tables) {
  // (where they lie depends on the chain's size: a grid's last, shorter batch of dates has its own place)
  if (ptr.bk_table_key)
    upload_tables = !(ptr.bk_table_key->where == L.tabs_dev && ptr.bk_table_key->nu == a.nu &&
                      ptr.bk_table_key->kappa == a.kappa && ptr.bk_table_key->sigma2 == a.sigma2 &&
                      ptr.bk_table_key->T == a.T);
  if (upload_tables) {
    // … and the slot bitmaps and the counters beside them start from zero; after that every chain leaves them so
    // (a workgroup gives its slot back, bk_tail_kernel resets what it counted with): no 5 µs fill per solve
    (void)hipMemsetAsync(a.slot_busy, 0, kSlotBitmapBytes, s);
    BkBessel tabs;
    if (!bessel_table(a.nu, tabs.t[0]) || !bessel_table(a.nu - a.n_int, tabs.t[1]))
      return (int)hipErrorInvalidValue;  // the series table of hh_bessel.h does not reach |z| = 13: not for ν > -1
    hipLaunchKernelGGL(bk_tables_kernel, dim3(1), dim3(64), 0, s, tabs, L.tabs_dev, a.kappa, a.sigma2, a.T, a.eta_k,
                       a.zeta_k);
    if (ptr.bk_table_key) *ptr.bk_table_key = BkTableKey{L.tabs_dev, a.nu, a.kappa, a.sigma2, a.T};
  }
  return 0;
}

// the chain behind the draws: the CF kernel (draw: 0 the draws are in place, 1 made there, 2 the caller's, read
// there), then the tail kernel, which leaves the sums of the chain's records in `accum` (slot HH_ACC_NPATHS: n_acc)
// workgroups of the CF kernel the current device holds at once (its four instantiations have the same registers and LDS)
uint32_t cf_resident_workgroups() {
  static std::atomic<uint32_t> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0u;
  uint32_t v = cache[dev].load(std::memory_order_relaxed);
  if (v == 0u) {
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bk_cf_kernel<0, 1>, kTile, 0) != hipSuccess || cus <= 0 || per_cu <= 0)
      return 0u;
    v = (uint32_t)cus * (uint32_t)per_cu;
    cache[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

void bk_chain(const BkArgs& a0, const BkLayout& L, hipStream_t s, double* accum, double n_acc, int draw = 0) {
  BkArgs a = a0;
  const uint32_t resident = cf_resident_workgroups();
  a.drain_tile = resident != 0u && L.n_tiles > resident ? L.n_tiles - resident : 0xffffffffu;
  const dim3 b(kTile), g(L.n_tiles);
  const BkTables* tabs = static_cast<const BkTables*>(L.tabs_dev);
  if (a.order)
    hipLaunchKernelGGL(bk_cf_kernel<1>, g, b, 0, s, a, tabs);
  else if (draw == 1)
    hipLaunchKernelGGL((bk_cf_kernel<0, 1>), g, b, 0, s, a, tabs);
  else if (draw == 2)
    hipLaunchKernelGGL((bk_cf_kernel<0, 2>), g, b, 0, s, a, tabs);
  else
    hipLaunchKernelGGL(bk_cf_kernel<0>, g, b, 0, s, a, tabs);
  hipLaunchKernelGGL(bk_tail_kernel, dim3(kHeavyGrid), b, 0, s, static_cast<const BkArgs*>(a.args_dev), tabs, L.n_tiles,
                     n_acc, accum, a.counters, static_cast<const double*>(a.records));
}

}  // namespace

int launch_bk(const hh_model& m, const hh_config& c, const DevicePtrs& ptr, hipStream_t s,
              const BkTransition* tr, bool upload_tables) {
  if (!ptr.accum) return (int)hipErrorInvalidValue;  // the chain adds its own records
  BkArgs a{};
  BkLayout L{};
  if (tr) {
    a.in_spot = tr->in_spot; a.in_var = tr->in_var;
    a.out_spot = tr->out_spot; a.out_var = tr->out_var;
    a.step = tr->step;
  }
  int rc = bk_prepare(m, c, ptr, c.n_paths, a, L);
  if (rc) return rc;
  a.replay = c.noise_mode == HH_NOISE_REPLAY ? ptr.replay : nullptr;
  if (!tr) a.iv_keep = a.iv_store;  // the one-shot law: ∫V per trajectory stays, for launch_bk_refinish
  if ((rc = bk_tables(a, L, ptr, s, upload_tables))) return rc;
  int draw = a.replay ? 2 : 1;  // the terminal law: drawn, or read, by the CF kernel
  if (a.in_var) {               // a transition of a grid: its own kernel
    if (a.replay) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bk_draw_kernel, dim3(L.n_tiles), dim3(kTile), 0, s, a);
    draw = 0;
  }
  bk_chain(a, L, s, ptr.accum, (double)c.n_paths, draw);
  return (int)hipGetLastError();
}

bool bk_same_chain(const hh_model& a, const hh_model& b) {
  return a.kappa == b.kappa && a.theta == b.theta && a.sigma == b.sigma && a.V0 == b.V0 && a.T == b.T;
}

int launch_bk_refinish(const hh_model& m, const hh_config& c, const DevicePtrs& ptr, const double* rec0, hipStream_t s) {
  BkArgs a{};
  BkLayout L{};
  const int rc = bk_prepare(m, c, ptr, c.n_paths, a, L);  // the same places in the same scratch as the chain's own block
  if (rc) return rc;
  a.iv_keep = a.iv_store;
  hipLaunchKernelGGL(bk_refinish_kernel, dim3(L.n_tiles + (uint32_t)kHeavyGrid), dim3(kTile), 0, s, a, rec0, L.n_tiles);
  return (int)hipGetLastError();
}

uint32_t bk_grid_dates_per_chain(uint64_t n_paths, uint32_t n_steps, int term_cache) {
  // The term cache no longer grows with the pairs of a chain; what does is 48 bytes per pair (draws, ∫V,
  // decision words).  2^22 pairs fill the chip more than twenty times over and keep that at 0.2 GB: a
  // longer grid is cut into several chains, the dates spread evenly over them.
  (void)term_cache;
  const uint64_t pairs_max = (uint64_t)1 << 22;
  const uint64_t per = n_paths ? pairs_max / n_paths : 1;
  if (per <= 1) return 1;
  const uint64_t chains = (n_steps + per - 1) / per;
  return (uint32_t)((n_steps + chains - 1) / chains);
}

// where a chain over n_paths trajectories left its decision words / series lengths inside `scratch`
void bk_diag_ptrs(const void* scratch, uint64_t n_paths, int term_cache, const uint32_t** decisions,
                  const uint32_t** series_len) {
  const size_t n_tiles = tiles_for(n_paths), lanes = n_tiles * kTile;
  const unsigned char* base = reinterpret_cast<const unsigned char*>(scratch);
  const double* cache = reinterpret_cast<const double*>(base + bk_flags_bytes(n_tiles));
  const double* draws = cache + bk_cache_columns(n_tiles) * (size_t)phi_cache_cap(term_cache);
  const uint32_t* diag = reinterpret_cast<const uint32_t*>(draws + 5 * lanes);
  *decisions = diag;
  *series_len = diag + lanes;
}

// device scratch of the ordered form of a grid chain over n_chain pairs:
//   the order [lanes] x uint32 | counts [256][n_runs] x uint32 | totals [256] x uint32 | keys [lanes] x uint8
static uint32_t grid_sort_runs(uint64_t n_chain) { return (uint32_t)((n_chain + kSortRun - 1) / kSortRun); }
size_t bk_grid_sort_bytes(uint64_t n_chain) {
  const size_t lanes = (size_t)tiles_for(n_chain) * kTile;
  return lanes * sizeof(uint32_t) + ((size_t)256 * grid_sort_runs(n_chain) + 256) * sizeof(uint32_t) + lanes + 256;
}

int launch_bk_grid(const hh_model& m, const hh_config& c, const DevicePtrs& ptr, hipStream_t s,
                   double* spot_rows, double* var_rows, uint32_t k0, uint32_t n_dates, bool upload_tables) {
  if (!ptr.accum) return (int)hipErrorInvalidValue;
  BkArgs a{};
  BkLayout L{};
  const uint64_t n_row = c.n_paths, n_chain = n_row * n_dates;
  int rc = bk_prepare(m, c, ptr, n_chain, a, L);
  if (rc) return rc;
  a.in_var = var_rows;  // pair i = b·n_row + trajectory starts from var_rows[i]
  a.step = k0;
  a.iv_out = a.iv_store;
  if ((rc = bk_tables(a, L, ptr, s, upload_tables))) return rc;
  const dim3 rows(tiles_for(n_row)), b(kTile);
  const bool ordered = ptr.bk_sort && n_chain >= kGridOrderMinPairs && n_chain < (1ull << 31);
  const size_t lanes = (size_t)L.n_tiles * kTile;
  const uint32_t n_runs = grid_sort_runs(n_chain);
  uint32_t* perm = reinterpret_cast<uint32_t*>(ptr.bk_sort);
  uint32_t* counts = perm + lanes;
  uint32_t* totals = counts + (size_t)256 * n_runs;
  uint8_t* keys = reinterpret_cast<uint8_t*>(totals + 256);
  hipLaunchKernelGGL(bk_draw_grid_kernel, rows, b, 0, s, a, n_row, k0, n_dates, var_rows, ordered ? keys : nullptr);
  if (ordered) {  // the pairs in the order of their Bessel arguments (grid_order_key)
    const dim3 sort_grid((n_runs + kSortWaves - 1) / kSortWaves);
    hipLaunchKernelGGL(grid_sort_count_kernel, sort_grid, b, 0, s, static_cast<const uint8_t*>(keys), (uint32_t)n_chain,
                       n_runs, counts);
    hipLaunchKernelGGL(grid_sort_scan_kernel, dim3(256), b, 0, s, counts, n_runs, totals);
    hipLaunchKernelGGL(grid_sort_scatter_kernel, sort_grid, b, 0, s, static_cast<const uint8_t*>(keys), (uint32_t)n_chain,
                       n_runs, static_cast<const uint32_t*>(counts), static_cast<const uint32_t*>(totals), perm);
    // The chain reads each pair's draws and start variance, and writes its ∫V, THROUGH the order (BkArgs::order): a
    // pair's 48 bytes, fetched at random by a kernel that then computes for 0.3 µs per pair, hide behind the other
    // waves' arithmetic.  Copying them into the order first (a gather by destination 0.19 ms; by source through the
    // inverse permutation 0.10 ms) and scattering ∫V back (0.035) were passes the chain does not need.
    BkArgs o = a;
    o.order = perm;
    bk_chain(o, L, s, ptr.accum, (double)n_chain);
  } else {
    bk_chain(a, L, s, ptr.accum, (double)n_chain);
  }
  hipLaunchKernelGGL(bk_grid_spots_kernel, rows, b, 0, s, a, n_row, n_dates,
                     static_cast<const double*>(var_rows), spot_rows);
  return (int)hipGetLastError();
}

}  // namespace hh
