/*
 * hedgehog_mc.h — C-ABI of libhedgehog_mc.so, the MI355X (gfx950) Monte Carlo engine that stands in
 * for the hot path of Hedgehog.jl's
 *
 *     solve(prob::PricingProblem{VanillaOption{…,European,C,Spot},I}, method::MonteCarlo)
 *                                             (reference: src/pricing_methods/montecarlo.jl:478-493)
 *
 * and for the ForwardDiff-dual Greeks that run through it (src/greeks/greeks_problem.jl:249-262,
 * 559-568).  The reference has no FFI seam of its own: its extension point is Julia multiple
 * dispatch on `solve`.  The entry points below are what a Julia `ccall` — or the
 * Python host mirror in this repository — binds to replace that method body.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (0 = HH_OK, negative = error) and
 *     never throws or aborts across the boundary; text via hh_last_error().
 *   - all arithmetic is IEEE fp64 (the reference computes in Float64 throughout).
 *   - the caller owns every buffer it passes, for the duration of the call only — also with the
 *     ASYNCHRONOUS entry points: whatever they read from HOST memory (seeds, increments, strikes) has
 *     been read when they return (the call waits for its staging copies, not for its kernels; pinned
 *     host memory included); DEVICE operands of an asynchronous call are the exception, see below.
 *     The library owns whatever it allocates inside hh_ctx and releases it in hh_ctx_destroy().
 *   - a hh_ctx is bound to ONE device and ONE HIP stream; its entry points serialise on an
 *     internal mutex, so sharing one between threads is safe but gains nothing — use one ctx per
 *     host thread / per GPU.  Multi-GPU = either ONE call on a hh_mgpu (one ctx per device, the
 *     shards enqueued concurrently and the all-reduce inside the library: hh_mgpu_solve), or one process (or
 *     thread) per GPU, each with its own ctx, exchanging only the HH_ACC_LEN-double accumulator
 *     vector (hh_mc_accumulate + the caller's all-reduce + hh_mc_finalize).
 *   - DEVICE buffers the caller passes (seeds / replay / terminal `_on_device`, grids) are read and
 *     written on the ctx's stream, which does not wait for any other stream: data produced on another
 *     stream (PyTorch's, say) must be complete before the call — synchronize it, or lend that stream
 *     to the ctx with hh_ctx_set_stream — and results of the ASYNCHRONOUS entry points
 *     (hh_mc_accumulate, hh_wiener_fill, hh_replay_pack, the hh_lsm_shard_* phases) are ready only after
 *     hh_ctx_synchronize.  The synchronous ones (hh_mc_solve, hh_lsm_solve, …) return with their
 *     device outputs complete.
 *   - there is NO CPU fallback in this library: without a HIP device every compute entry point
 *     fails with HH_ERR_HIP.
 */
#ifndef HEDGEHOG_MC_H
#define HEDGEHOG_MC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HH_ABI_VERSION 6
#define HH_MAX_PARTIALS 8   /* max. number of dual-number partials carried through one solve    */
#define HH_TILE_PATHS 256   /* paths per tile of the tile-major REPLAY layout (see below)        */
#define HH_ACC_LEN 16       /* doubles in the accumulator vector exchanged between GPUs          */
#define HH_MAX_MODELS 16    /* models one hh_mc_solve_multi call prices on the same draws          */

/* accumulator vector slots (all are plain sums, so one SUM all-reduce combines shards) */
#define HH_ACC_SUM 0        /* Σ payoff (undiscounted; antithetic: Σ of pair averages)           */
#define HH_ACC_SUMSQ 1      /* Σ payoff²                                                         */
#define HH_ACC_DSUM 2       /* [2, 2+HH_MAX_PARTIALS): Σ ∂payoff/∂θ_k                             */
#define HH_ACC_NPATHS 10    /* number of trajectories accumulated                                */
#define HH_ACC_BK_NEWTON_FAIL 11
#define HH_ACC_BK_BISECT 12
#define HH_ACC_BK_MAXGUESS 13
#define HH_ACC_BK_CF_TERMS 14 /* Σ characteristic-function series terms evaluated (Broadie–Kaya) */

enum hh_status {
  HH_OK = 0,
  HH_ERR_INVALID = -1,     /* bad argument (NULL, sizes, ranges)                                  */
  HH_ERR_UNSUPPORTED = -2, /* combination the reference itself cannot run (e.g. BK + antithetic)  */
  HH_ERR_HIP = -3,         /* HIP runtime error, or no HIP device                                 */
  HH_ERR_NOMEM = -4,
  HH_ERR_RCCL = -5,        /* RCCL missing or failed where the caller REQUIRED it (HH_MGPU_RCCL)  */
  HH_ERR_DEVICE_TIMEOUT = -6 /* the record reduction inside a simulation kernel gave up waiting for a
                                workgroup's record (HH_OPT_FINISH_SPIN_TICKS): that solve's sums are lost,
                                the context has been reset and is usable again (hh_ctx_check_last)    */
};

/* montecarlo.jl:8-22 */
enum hh_dynamics { HH_LOGNORMAL = 0, HH_HESTON = 1 };
/* montecarlo.jl:86-115: EulerMaruyama / BlackScholesExact / HestonBroadieKaya */
enum hh_strategy { HH_EULER_MARUYAMA = 0, HH_EXACT_LAW = 1, HH_BROADIE_KAYA = 2 };
enum hh_noise_mode { HH_NOISE_GENERATE = 0, HH_NOISE_REPLAY = 1 };
enum hh_replay_layout { HH_REPLAY_TILE_MAJOR = 0, HH_REPLAY_PATH_MAJOR = 1 };

/* readings of find_zero(func, x0, Order2(); atol, maxeval) / find_zero(func, (0, max_guess); xtol, maxeval) */
enum hh_bk_root_form { HH_BK_ROOT_SECANT = 0,    /* the secant iteration from (x0 + dx, x0)                        */
                       HH_BK_ROOT_ORDER2 = 1 };  /* Roots' Order2: a Steffensen step, a secant step while f is large */
enum hh_bk_bracket_form { HH_BK_BRACKET_MIDPOINT = 0, /* arithmetic bisection to width <= atol                      */
                          HH_BK_BRACKET_ROOTS = 1 };  /* Roots' Bisection: midpoints of the BIT PATTERNS, to the last bit */
enum hh_bk_caps { HH_BK_CAPS_AS_WRITTEN = 0,     /* `maxeval` caps evaluations / iterations as the author meant    */
                  HH_BK_CAPS_ROOTS_DEFAULT = 1 };/* `maxeval` is no keyword of Roots 2 and is ignored: 40 steps / none */

typedef struct hh_ctx hh_ctx; /* opaque: device, stream, scratch */

/*
 * Model scalars after the host has resolved dates and curves exactly as the reference does:
 *   T        = yearfrac(referenceDate, expiry)                      (montecarlo.jl:173,197)
 *   r_drift  = zero_rate(rate, 0.0) for Euler (montecarlo.jl:176,200),
 *              zero_rate(rate, expiry) for the exact laws (montecarlo.jl:299,318)
 *   discount = df(rate, expiry)                                     (montecarlo.jl:489)
 *   cp       = +1 call / -1 put                                     (payoffs.jl:76-87)
 * Lognormal: sigma is the flat vol; V0, kappa, theta, rho are ignored.
 * d* are the dual-number seeds: each is NULL or points to n_partials doubles, direction k of
 * parameter θ being dθ[k].  (Spot enters as x0 = log S0, the library applies dx0 = dS0/S0.)
 * Cost: the library carries per trajectory one derivative per PARAMETER with a non-zero seed among
 * V0, kappa, theta, sigma (at most 4, whatever n_partials is) and assembles the n_partials
 * directions from them; seeds on S0, r_drift, discount and strike cost nothing per trajectory.
 */
typedef struct hh_model {
  double S0, V0, kappa, theta, sigma, rho;
  double r_drift, discount, T, strike, cp;
  const double *dS0, *dV0, *dkappa, *dtheta, *dsigma, *dr_drift, *ddiscount, *dstrike;
} hh_model;

typedef struct hh_config {
  int32_t dynamics;          /* enum hh_dynamics                                                  */
  int32_t strategy;          /* enum hh_strategy                                                  */
  int32_t antithetic;        /* 0 NoVarianceReduction, 1 Antithetic (montecarlo.jl:29-43)         */
  int32_t em_split;          /* Euler diffusion evaluated at K=u+dt·f(u) (1) or at u (0)          */
  int32_t compat_sqrt_alpha; /* exact lognormal mean uses (r-σ²/2)·√T as montecarlo.jl:302 (1)
                                or the correct ·T (0); identical at T = 1                         */
  int32_t noise_mode;        /* enum hh_noise_mode                                                */
  int32_t replay_layout;     /* enum hh_replay_layout (REPLAY only)                               */
  int32_t seeds_on_device;   /* seeds / replay / terminal point to device memory of ctx's device  */
  int32_t replay_on_device;
  int32_t terminal_on_device;
  uint32_t n_steps;          /* SimulationConfig.steps (montecarlo.jl:60); exact laws ignore it   */
  uint32_t n_partials;       /* 0..HH_MAX_PARTIALS                                                */
  uint64_t n_paths;          /* SimulationConfig.trajectories of THIS shard: 1 .. 2^32 - 256 per
                                call (one 256-thread workgroup per 256 trajectories, < 2^32
                                threads per launch); larger ensembles are sharded by path_offset.
                                n_steps: Euler at most 262 140; LSM / exact grid at most 65 534   */
  uint64_t path_offset;      /* global index of this shard's first trajectory (exact laws draw by
                                global index from ONE key, montecarlo.jl:456, so results do not
                                depend on the sharding)                                           */
  const uint64_t* seeds;     /* Euler: ≥ n_paths per-trajectory seeds (montecarlo.jl:331);
                                exact laws: seeds[0] only (montecarlo.jl:456)                     */
  const double* replay;      /* REPLAY: Wiener increments, see hh_replay_elems()                  */
  /* Broadie–Kaya controls, defaults of sample_from_cf.jl:27,50,75,105-113 when 0 */
  double bk_n_sigma;         /* n = 5                                                             */
  double bk_cf_tol;          /* cf_tol = 1e-3                                                     */
  double bk_atol;            /* atol = 1e-4                                                       */
  double bk_moment_h;        /* h = 1e-2                                                          */
  int32_t bk_newton_maxiter; /* 10                                                                */
  int32_t bk_bisect_maxiter; /* 100                                                               */
  /* How the two `find_zero` calls of inverse_cdf (sample_from_cf.jl:118,128) are READ — Roots.jl is not part of the
     reference's tree, and three things about those calls cannot be told from its text (the enums above name them,
     DESIGN.md says why; julia/parity_replay.jl's `bk_root_probe` + tools/check_reference_replay.py decide them on a
     Julia host).
     All 0 = what this library prices with.  Any other value sends the whole ensemble through the whole-trajectory
     kernel (a parity seam, ~10x slower), with the same draws.                                              */
  int32_t bk_root_form;      /* enum hh_bk_root_form                                              */
  int32_t bk_bracket_form;   /* enum hh_bk_bracket_form                                           */
  int32_t bk_caps;           /* enum hh_bk_caps                                                   */
  int32_t reserved0;         /* 0                                                                 */
  /* Optional operand-shape checks (0 = not checked): the number of ELEMENTS behind `seeds` and
     `replay`.  When given, a buffer shorter than what the kernels will index is rejected with
     HH_ERR_INVALID on the host instead of faulting on the device.                               */
  uint64_t seeds_len;
  uint64_t replay_len;
} hh_config;

typedef struct hh_result {
  double price;               /* discount · mean(payoff)            (montecarlo.jl:489-490)       */
  double std_error;           /* build extension; the reference computes none                     */
  double sum_payoff, sumsq_payoff;
  double dprice[HH_MAX_PARTIALS]; /* partials of price, direction k                              */
  uint64_t n_paths_done;
  uint64_t bk_newton_fail, bk_bisect_fallback, bk_maxguess_fallback, bk_cf_terms;
  double kernel_ms;           /* HIP-event time of everything the call enqueued: staging copies
                                 of host seeds / increments, simulation, record reduction; the
                                 FIRST call of a size also grows the ctx's scratch buffers
                                 (hipMalloc / hipFree synchronise) inside this bracket — use
                                 hh_ctx_enable_timing for the simulation kernel alone             */
  double total_ms;            /* host wall time of the call                                       */
} hh_result;

int hh_abi_version(void);

/* Context. device_id is a HIP device ordinal of this process. */
int hh_ctx_create(hh_ctx** out, int device_id);
void hh_ctx_destroy(hh_ctx* ctx);
/* Borrow an external hipStream_t (e.g. PyTorch's current stream).  NULL is a valid handle: the
 * device's default (null) stream — which is what PyTorch uses unless told otherwise.
 * hh_ctx_reset_stream goes back to the ctx's own non-blocking stream.  A switch orders the new
 * stream behind whatever the ctx still has queued on the old one (the asynchronous entry points
 * share the ctx's scratch buffers); the borrowed stream must stay alive until then. */
int hh_ctx_set_stream(hh_ctx* ctx, void* hip_stream);
int hh_ctx_reset_stream(hh_ctx* ctx);
const char* hh_last_error(const hh_ctx* ctx); /* NUL-terminated, owned by ctx (or static if NULL) */
/* Build options of a context (none changes a result).  HH_OPT_LSM_FORM: how hh_lsm_solve /
 * hh_lsm_solve_grid run the backward induction.
 *   HH_LSM_FORM_PERSISTENT  ONE launch that keeps every trajectory's stopping state in registers,
 *                           reads each row of the grid once and exchanges the per-date sums between
 *                           its workgroups in the kernel; applies to ensembles of up to 2^21
 *                           trajectories (256 chunks); a COOPERATIVE launch — refused by the runtime, and
 *                           replaced by the other form, when its workgroups cannot all be resident;
 *   HH_LSM_FORM_PER_DATE    one launch per exercise date;
 *   HH_LSM_FORM_AUTO        (default) the persistent form whenever it applies (since round 3 it is the
 *                           faster one at every size: 2·10^6 x 100 dates 1.6 vs 2.9 ms, 2.6·10^5 x 100
 *                           0.59 vs 0.71 ms, 4·10^3 x 100 0.45 vs 0.50 ms), a launch per date beyond
 *                           2^21 trajectories.
 * All forms give bit-identical prices and stopping decisions.
 *
 * HH_OPT_BK_TERM_CACHE: how many CDF-series terms Re ϕ(h·j) the Broadie–Kaya kernels keep per column
 * (8 … 1024, default 256).  A column belongs to a resident workgroup slot of the kernel, not to a
 * trajectory, so the cache is 8 bytes x terms x 393 216 columns (0.81 GB at 256 terms) whatever the
 * ensemble size; beside it a solve keeps 48 bytes per trajectory.  Series that fit are evaluated once
 * and inverted on the cached terms; a trajectory whose series is longer re-evaluates the terms beyond
 * the cache in every CDF call (as the reference does with all of them), in a separate, slower kernel.
 * With the reference's controls the series has 10–15 terms (60 at short maturities, 100–250 for
 * d = 4κθ/σ² ≪ 1 or cf_tol ≪ 1e-3).
 *
 * HH_OPT_GRID_FORM: how hh_heston_exact_grid (and LSM on those paths) runs the dates of a grid:
 *   HH_GRID_FORM_BATCHED   (default) the variances of all dates first, then ONE kernel chain over every
 *                          (date, trajectory) pair — given the variances the CF inversions of different
 *                          dates are independent — then the spot rows; a chain takes at most 2^22 pairs
 *                          (48 bytes each; longer grids run as several chains);
 *   HH_GRID_FORM_PER_DATE  one kernel chain per date.
 * Both forms give bit-identical grids.
 *
 * HH_OPT_LSM_SPIN_TICKS: bound of every wait inside the persistent LSM launch, in ticks of the 100 MHz
 * constant clock (default -1 = 10^8 = one second).  The launch is cooperative (hipLaunchCooperativeKernel:
 * a grid that cannot be resident as a whole is refused at launch and the launch-per-date form runs), so
 * the bound is only the last guard; 0 makes every workgroup give up at once, which is how the tests
 * force the give-up-and-redo path (hh_lsm_result.persistent_fallbacks counts it; same result).
 *
 * HH_OPT_GRID_ORDER: 1 (default) a batched chain of hh_heston_exact_grid (HH_GRID_FORM_BATCHED) runs its (date,
 * trajectory) pairs sorted by a coarse key of V0·V_T — the size of their Bessel argument — so that the lanes of a
 * wave share a regime of the Bessel function and series of similar length (short transitions spread both widely:
 * 0.53 active lanes per instruction unsorted; the same draws sorted run the chain 32 % faster); 0: in their
 * natural order.  The same grid either way, bit for bit.
 *
 * HH_OPT_FUSE_REDUCE: who adds the workgroups' partial sums of a single-payoff solve (mean(payoffs),
 * montecarlo.jl:490): 0 a second kernel (reduce_records_kernel), 1 the simulation kernel itself (the last
 * tile's workgroup, once every record has arrived), 2 (default) by what was measured — the simulation kernel
 * up to 512 records (a launch saved is 2 % of a small solve) and for Euler runs of 32 steps or more (the
 * reducer's wait disappears behind the other workgroups: -0.3 % at 10^6 x 252), the second kernel for short
 * kernels with thousands of records (the exact law at 10^6 - 10^7 trajectories: 1-4 % faster that way).  The
 * same additions in the same order every way: bit-identical results.
 *
 * HH_OPT_FINISH_SPIN_TICKS: how long the workgroup that adds the records INSIDE a simulation kernel
 * (HH_OPT_FUSE_REDUCE) waits for a record that has not arrived, in ticks of the 100 MHz constant clock (default
 * -1 = 5·10^8 = five seconds).  Every workgroup needs only a slot of its own to finish, so the bound is the last
 * guard against a workgroup that died or a queue preempted for that long.  When it is passed the solve's sums are
 * NaN, and — because the missing record may still land in the buffer the next launch reads — every solve queued
 * behind it on this context gives NaN too, until the host has reset the buffer: hh_mc_solve and its kin do that
 * themselves and return HH_ERR_DEVICE_TIMEOUT; callers of the asynchronous entry points ask with
 * hh_ctx_check_last.  0 makes the reducer give up on the first record it does not find, which — with
 * HH_OPT_FINISH_TILE_FIRST = 1: the FIRST tile's workgroup adds the records instead of the last one's, so nearly
 * none is there when it looks — is how the tests force that path. */
enum hh_option { HH_OPT_LSM_FORM = 1, HH_OPT_BK_TERM_CACHE = 2, HH_OPT_GRID_FORM = 3, HH_OPT_LSM_SPIN_TICKS = 4,
                 HH_OPT_FUSE_REDUCE = 5, HH_OPT_GRID_ORDER = 6, HH_OPT_FINISH_SPIN_TICKS = 7,
                 HH_OPT_FINISH_TILE_FIRST = 8 };
enum hh_grid_form { HH_GRID_FORM_PER_DATE = 0, HH_GRID_FORM_BATCHED = 1 };
enum hh_lsm_form { HH_LSM_FORM_PER_DATE = 0, HH_LSM_FORM_PERSISTENT = 1, HH_LSM_FORM_AUTO = 2 };
int hh_ctx_set_option(hh_ctx* ctx, int32_t option, int64_t value);
/*
 * For callers of the ASYNCHRONOUS entry points (hh_mc_accumulate and its kin, whose sums stay on the device): waits
 * for the context's stream, then HH_OK — or HH_ERR_DEVICE_TIMEOUT when a record reduction inside a simulation kernel
 * gave up since the last check (HH_OPT_FINISH_SPIN_TICKS): the accumulators written since then hold NaN, the
 * context's record buffer has been reset, the context is usable again.  Call it where the sums are read back (a NaN
 * in slot HH_ACC_NPATHS says why), or once per batch of solves.
 */
int hh_ctx_check_last(hh_ctx* ctx);

/*
 * Replaces the body of solve(prob, ::MonteCarlo) (montecarlo.jl:478-493): simulate, payoff,
 * discount·mean.  Synchronous.  `terminal` (nullable) receives the samples at expiry that the
 * reference keeps in MonteCarloSolution.ensemble (pricing_solutions.jl:22-27): n_paths doubles,
 * followed by n_paths mirrored samples when antithetic.
 */
int hh_mc_solve(hh_ctx* ctx, const hh_model* model, const hh_config* cfg, hh_result* out,
                double* terminal);

/*
 * Split form for path-sharded multi-GPU runs: enqueue simulation + reduction on the ctx stream and
 * leave the HH_ACC_LEN accumulator doubles in DEVICE memory `accum_dev`, so the caller can all-reduce
 * them (RCCL) and then finalize on the host.  No wait for the kernels; host seeds / increments in cfg
 * have been read when the call returns (it waits for their staging copies only).
 */
int hh_mc_accumulate(hh_ctx* ctx, const hh_model* model, const hh_config* cfg, double* accum_dev,
                     double* terminal);
/* Pure host arithmetic: accumulator vector (HOST memory) -> price, std_error, dprice. */
int hh_mc_finalize(const hh_model* model, const hh_config* cfg, const double* accum_host,
                   hh_result* out);

/*
 * SEVERAL MODELS ON THE SAME DRAWS, in one pass — what the reference's bumped Greeks are made of:
 * compute_fd_derivative (greeks_problem.jl:279-303) solves 2 problems that differ in one number, the
 * second-order stencils (:396-422) 3 or 4, each a full solve(prob, method) with the seeds of the SAME
 * SimulationConfig (common random numbers).  Here the n_models problems — same payoff type, same cfg: dynamics,
 * strategy, trajectories, steps, seeds or increments, variance reduction — are simulated TOGETHER: a lane
 * carries one state per model and steps them all on each draw, so the normals (GENERATE) or the 16 bytes of
 * increments (REPLAY) are paid once per path-step, not once per model and path-step.  Result k is what
 * hh_mc_solve(ctx, &models[k], cfg, …) returns, bit for bit (same arithmetic per model, same summation order).
 *   models     n_models hh_model (1 .. HH_MAX_MODELS); anything may differ between them (spot, variance
 *              parameters, ρ, rate, discount, expiry time T, strike, call/put); d* seeds are ignored
 *   cfg        as for hh_mc_solve; n_partials must be 0 (HH_ERR_UNSUPPORTED otherwise)
 *   out        n_models results;   accum_dev  n_models x HH_ACC_LEN doubles, model-major (all-reducible as ONE
 *              vector over path shards, then hh_mc_finalize(&models[k], cfg, accum + k·HH_ACC_LEN, &out[k]))
 *   terminals  NULL, or n_models pointers (each nullable): model k's samples at expiry, as `terminal` of hh_mc_solve
 * Up to 4 models share a pass (more run as ⌈n/4⌉ passes); Euler–Maruyama and the exact lognormal law share
 * their draws.  Broadie–Kaya shares a whole CHAIN between models that differ in nothing its variance process
 * sees (κ, θ, σ, V0, T equal: a bumped spot, rate, ρ or strike — the finite-difference delta, gamma, rho): the
 * first such model runs the chain, the others are finished from the ∫V it sampled (~10 µs each at 10^6
 * trajectories; a central delta 0.355 ms for the reference's two solves of 0.336); a bumped κ, θ, σ, V0 or T is
 * a chain of its own.
 * Path-major REPLAY increments are repacked to the tile-major layout first.
 */
int hh_mc_solve_multi(hh_ctx* ctx, const hh_model* models, uint32_t n_models, const hh_config* cfg,
                      hh_result* out, double* const* terminals);
int hh_mc_accumulate_multi(hh_ctx* ctx, const hh_model* models, uint32_t n_models, const hh_config* cfg,
                           double* accum_dev, double* const* terminals);

/*
 * Diagnostics of the last HH_BROADIE_KAYA solve of this context (hh_mc_solve / _accumulate with the same
 * n_paths; the scratch it reads is overwritten by the next Broadie–Kaya call): per trajectory, WHAT the
 * root search of inverse_cdf (sample_from_cf.jl:105-135) did —
 *   decisions[i]   bits 0-7   CDF evaluations of the secant iteration (find_zero(…, Order2(); maxevals))
 *                  bits 8-9   0 the secant's root was accepted, 1 bisection finished it (:127-133),
 *                             2 no sign change: max_guess (:124-126)
 *                  bits 16-23 bisection iterations;   bit 31  series longer than the term cache
 *   series_len[i]  terms of the CDF series, set by |ϕ(h j)|/j < π·cf_tol/2 (sample_from_cf.jl:88)
 * Two implementations that agree on both have run the same sequence of operations on that trajectory;
 * the parity tests compare samples per trajectory where they agree and count where they do not.
 * Either output may be NULL.
 */
int hh_bk_decisions(hh_ctx* ctx, uint64_t n_paths, uint32_t* decisions, uint32_t* series_len);

/*
 * Multi-GPU solve in ONE call from ONE host thread — what solve(prob, method) (montecarlo.jl:478-493,
 * the one EnsembleProblem of :329-333,351) becomes when the trajectories are sharded over the GPUs of
 * a node.  A hh_mgpu owns one hh_ctx (own stream, own scratch) per listed device.
 * hh_mgpu_solve cuts the ensemble into contiguous ranges [g·per, min(N, (g+1)·per)), per = ⌈N/G⌉
 * (rounded up to a multiple of HH_TILE_PATHS for tile-major REPLAY data, whose tiles cannot be cut),
 * slices seeds / increments / terminal samples by the same ranges (exact laws: path_offset), enqueues
 * every shard's kernels without waiting, and combines the HH_ACC_LEN-double accumulator vectors
 *   HH_MGPU_REDUCE_RCCL  by ONE ncclAllReduce(ncclSum, ncclDouble) inside ncclGroupStart/End on the
 *                        shards' own streams (communicators from ncclCommInitAll at creation; RCCL is
 *                        bound at run time — librccl.so.1, or the path in $HEDGEHOG_MC_RCCL — so the
 *                        library itself has no link-time dependency on it), or
 *   HH_MGPU_REDUCE_HOST  by a deterministic ordered sum on the host, g = 0 … G−1 (also what is used
 *                        when RCCL is absent, fails to initialise — e.g. a device listed twice — or
 *                        fails in the call: the solve is still completed, hh_mgpu_last_error() says why).
 * Draws depend on (seed, counter) only, so a G-device result differs from the one-device result by
 * the order of that last sum alone (≤ 1e-13 relative); with one device there is no sum and the result
 * is hh_mc_solve's bit for bit.
 *   flags of hh_mgpu_create: HH_MGPU_AUTO  RCCL when n_devices > 1 and it initialises, else host sum;
 *                            HH_MGPU_HOST_SUM  never touch RCCL;
 *                            HH_MGPU_RCCL  RCCL or fail (HH_ERR_RCCL), used even for one device.
 * Enqueueing: the host-side cost of one shard (checks, events, launches) is tens of microseconds, so
 * the devices are served CONCURRENTLY — device 0 by the calling thread, every other device by a
 * library-owned thread bound to it (created on first use, joined in hh_mgpu_destroy; it polls for
 * ~150 µs after a job and sleeps otherwise, and never calls back into the host language).
 * hh_mgpu_set_option(HH_MGPU_OPT_ENQUEUE, HH_MGPU_ENQUEUE_SERIAL) enqueues one shard after the other
 * from the calling thread instead; results are identical.  hh_mgpu_enqueue_stats reports what the last
 * solve's enqueue phase cost on the host: per shard (measured in the thread that enqueued it) and as a
 * whole (wall time until every shard was enqueued).
 * A collective that FAILS in the call (ncclAllReduce refusing rank g after ranks < g were enqueued) may
 * leave kernels on the earlier ranks' streams that wait for peers which never come.  The library never
 * synchronises such a stream: it aborts the communicators (ncclCommAbort — a library without it is not used), retires every shard stream
 * for a fresh one that continues behind the shard's last kernel, and — HH_MGPU_AUTO — completes this
 * and every later solve with the host's ordered sum (hh_mgpu_reduce_mode() then says HOST); with
 * HH_MGPU_RCCL the call returns HH_ERR_RCCL once the caller's buffers are no longer read, and so does
 * every later call on that context.
 * hh_mgpu_solve takes HOST buffers in cfg (seeds, replay) and for `terminal` (n_paths doubles, then the
 * n_paths mirrored samples when antithetic — the layout of hh_mc_solve); *_on_device must be 0.  The
 * REPLAY buffer as a whole must be 16-byte aligned, as for hh_mc_solve; a shard's slice need not be
 * (odd row lengths cut at odd trajectories are staged by the library).
 * hh_mgpu_solve_shards takes one hh_config per device exactly as hh_mc_accumulate would on that device
 * (device-resident seeds / increments allowed; n_paths = 0 leaves a device idle) and optional per-shard
 * terminal pointers — for callers that keep their inputs in HBM.  hh_mgpu_ctx(mg, i) is the i-th
 * device's context (hh_device_malloc, hh_wiener_fill, hh_ctx_enable_timing … on that device); it stays
 * owned by mg.  out->kernel_ms is the LONGEST shard's HIP-event time, total_ms the host wall time.
 * After any error return nothing that reads the caller's buffers is still running (a collective that lost its
 * peers may still sit on a retired stream until ncclCommAbort has released it: hh_mgpu_destroy then leaves that
 * device's streams and buffers alone rather than wait for it).  Should a shard be impossible to move off the
 * stream the failed collective was enqueued on — a stream lent with hh_ctx_set_stream, or no new stream to be
 * had — the context is stuck: the call and every later one return HH_ERR_RCCL, whatever the flags.
 */
typedef struct hh_mgpu hh_mgpu;
enum hh_mgpu_flags { HH_MGPU_AUTO = 0, HH_MGPU_HOST_SUM = 1, HH_MGPU_RCCL = 2 };
enum hh_mgpu_reduce { HH_MGPU_REDUCE_HOST = 0, HH_MGPU_REDUCE_RCCL = 1 };
int hh_mgpu_create(hh_mgpu** out, const int* device_ids, int n_devices, int flags);
void hh_mgpu_destroy(hh_mgpu* mg);
const char* hh_mgpu_last_error(const hh_mgpu* mg);
int hh_mgpu_n_devices(const hh_mgpu* mg);
int hh_mgpu_reduce_mode(const hh_mgpu* mg);       /* enum hh_mgpu_reduce in use                    */
/* What a caller may print next to "reduce: rccl".  hh_mgpu_selftest sends a vector of ones from every device
 * through the exchange a solve uses (the grouped ncclAllReduce on the shards' streams, or the host's ordered
 * sum) and returns the count that came back — the ranks that really took part — and the mode it ran in.
 * hh_mgpu_rccl_info names the library the collective entry points were bound from (dladdr of ncclAllReduce;
 * empty when none), its ncclGetVersion (0 when unknown) and whether $HEDGEHOG_MC_RCCL chose it; HH_ERR_RCCL
 * when no usable RCCL is bound (ncclCommAbort is required: see above). */
int hh_mgpu_selftest(hh_mgpu* mg, int32_t* ranks_out, int32_t* reduce_mode_out /* nullable */);
int hh_mgpu_rccl_info(const hh_mgpu* mg, char* path_out, size_t cap, int32_t* version_out, int32_t* from_env_out);
enum hh_mgpu_option { HH_MGPU_OPT_ENQUEUE = 1 };
enum hh_mgpu_enqueue { HH_MGPU_ENQUEUE_SERIAL = 0, HH_MGPU_ENQUEUE_THREADS = 1 /* default */ };
int hh_mgpu_set_option(hh_mgpu* mg, int32_t option, int64_t value);
/* host microseconds of the last solve's enqueue phase: shard_us[n_devices] (nullable), *phase_us (nullable) */
int hh_mgpu_enqueue_stats(hh_mgpu* mg, double* shard_us, double* phase_us);
hh_ctx* hh_mgpu_ctx(hh_mgpu* mg, int i);          /* NULL when i is out of range                   */
/* shard g of an ensemble of n_paths over n_devices, as hh_mgpu_solve cuts it (tile_aligned = 1 for
 * tile-major REPLAY increments) */
void hh_mgpu_shard_range(uint64_t n_paths, int n_devices, int g, int tile_aligned, uint64_t* start,
                         uint64_t* stop);
int hh_mgpu_solve(hh_mgpu* mg, const hh_model* model, const hh_config* cfg, hh_result* out,
                  double* terminal);
int hh_mgpu_solve_shards(hh_mgpu* mg, const hh_model* model, const hh_config* shard_cfgs /* n_devices */,
                         hh_result* out, double* const* terminals /* nullable; n_devices, each nullable */);
int hh_mgpu_solve_basket(hh_mgpu* mg, const hh_model* model, const hh_config* cfg, const double* strikes,
                         const double* cps, uint32_t n_payoffs, hh_result* out /* n_payoffs */);
/* hh_mc_solve_multi with the trajectories sharded over the devices of mg: every device steps all n_models
 * models on its range's draws, ONE all-reduce of the n_models x HH_ACC_LEN accumulator block.  Host buffers
 * in cfg, as for hh_mgpu_solve; no terminal samples. */
int hh_mgpu_solve_multi(hh_mgpu* mg, const hh_model* models, uint32_t n_models, const hh_config* cfg,
                        hh_result* out /* n_models */);


/*
 * Several payoffs on ONE simulation — solve(::BasketPricingProblem, method) for payoffs that share
 * an expiry (src/calibration/basket.jl:35-38 prices them as independent solves; with the fixed
 * seeds of SimulationConfig they see the same trajectories, so the results coincide).  Payoff k is
 * max(cps[k]·(S_T − strikes[k]), 0); model->strike / model->cp are ignored.  The accumulator block
 * is n_payoffs × HH_ACC_LEN doubles, payoff-major (all-reducible as one vector); dual partials of
 * the model parameters are carried for every payoff (strike partials are not).  Works for every
 * simulation strategy, Broadie–Kaya included.
 */
int hh_mc_accumulate_basket(hh_ctx* ctx, const hh_model* model, const hh_config* cfg,
                            const double* strikes, const double* cps, uint32_t n_payoffs,
                            double* accum_dev, double* terminal);
int hh_mc_solve_basket(hh_ctx* ctx, const hh_model* model, const hh_config* cfg,
                       const double* strikes, const double* cps, uint32_t n_payoffs,
                       hh_result* out /* n_payoffs */, double* terminal);

/*
 * Carr–Madan Fourier price of a European vanilla on the device — solve(prob, ::CarrMadan)
 * (src/pricing_methods/carr_madan.jl:47-92) with the Heston marginal law (heston.jl:307-319) or the
 * lognormal one (sample_from_cf.jl:14-16; compat_sqrt_alpha as in hh_config).  It is the analytic
 * value the reference's MC tests compare against; model->r_drift = zero_rate(rate, expiry),
 * model->T = yearfrac(rate.reference_date, expiry), model->discount = df(rate, expiry).
 * alpha = damping factor, bound = integration bound (CarrMadan(α, bound, dynamics)).
 */
int hh_carr_madan(hh_ctx* ctx, const hh_model* model, int32_t dynamics, int32_t compat_sqrt_alpha,
                  double alpha, double bound, double* price_out);
/*
 * The same for a basket in ONE launch (one workgroup per payoff) — solve(::BasketPricingProblem,
 * ::CarrMadan), i.e. src/calibration/basket.jl:35-38 over carr_madan.jl:47-92: what the calibration
 * objective evaluates at every iterate (src/calibration/calibration.jl:75-88).  The payoffs share the
 * model's parameters (S0 and V0, kappa, theta, sigma, rho — or the lognormal sigma); per payoff k:
 * strikes[k], cps[k] (+1 call / -1 put), Ts[k] = yearfrac(rate.reference_date, expiry_k),
 * r_drifts[k] = zero_rate(rate, expiry_k), discounts[k] = df(rate, expiry_k).  model->strike, cp, T,
 * r_drift and discount are ignored.  1 .. 2^20 payoffs per call.
 */
int hh_carr_madan_basket(hh_ctx* ctx, const hh_model* model, int32_t dynamics,
                         int32_t compat_sqrt_alpha, double alpha, double bound, const double* strikes,
                         const double* cps, const double* Ts, const double* r_drifts,
                         const double* discounts, uint32_t n_payoffs, double* prices_out);
/*
 * Prices AND their gradient in one launch — what a ForwardDiff-differentiated calibration objective
 * (calibration.jl:75-88 with AutoForwardDiff) pushes through carr_madan.jl:47-92 and heston.jl:307-319
 * as Dual numbers.  grad_out[k][HH_CM_GRAD_LEN] = ∂price_k / ∂(S0, V0, kappa, theta, sigma, rho,
 * r_drift_k, discount_k) in that order (enum hh_cm_grad); the caller's Dual price is
 * price + Σ_j grad[j]·(partials of parameter j), r_drift_k and discount_k being whatever its rate curve
 * makes of its parameters.  Lognormal dynamics: sigma = the volatility; V0, kappa, theta, rho slots 0.
 */
enum hh_cm_grad { HH_CM_GRAD_S0 = 0, HH_CM_GRAD_V0, HH_CM_GRAD_KAPPA, HH_CM_GRAD_THETA, HH_CM_GRAD_SIGMA,
                  HH_CM_GRAD_RHO, HH_CM_GRAD_R_DRIFT, HH_CM_GRAD_DISCOUNT, HH_CM_GRAD_LEN };
int hh_carr_madan_basket_grad(hh_ctx* ctx, const hh_model* model, int32_t dynamics,
                              int32_t compat_sqrt_alpha, double alpha, double bound,
                              const double* strikes, const double* cps, const double* Ts,
                              const double* r_drifts, const double* discounts, uint32_t n_payoffs,
                              double* prices_out, double* grad_out);

/*
 * Longstaff–Schwartz American pricing on the full path grid:
 *   solve(::PricingProblem{VanillaOption{…,American,…}}, ::LSM)
 *                              (src/pricing_methods/least_squares_montecarlo.jl:99-165)
 * cfg: dynamics = HH_LOGNORMAL, strategy = HH_EXACT_LAW (the BlackScholesExact GBM-process paths the
 * reference's LSM is used with, montecarlo.jl:140-159; trajectory i keyed by seeds[i],
 * montecarlo.jl:331; antithetic = flipped σ, :270-284), n_steps, n_paths, antithetic, seeds.
 * step_discount = df(rate, referenceDate + T/nsteps) (:107); degree = LSM.degree (1..8).
 * Optional host outputs (nullable): stop_time / stop_value = the reference's stopping_info
 * ((n_paths·(1+antithetic)) entries), spot_grid = the (n_steps+1) x n_total path matrix in
 * step-major order (the transpose of extract_spot_grid's matrix, :47-85).
 */
typedef struct hh_lsm_result {
  double price, std_error;
  uint64_t n_paths_total;
  uint32_t rows_regressed, rows_skipped; /* time rows with / without an in-the-money path */
  double kernel_ms, total_ms;
  int32_t form;      /* HH_LSM_FORM_PER_DATE / _PERSISTENT: how the backward induction ran */
  int32_t persistent_fallbacks; /* 1: the one-launch form gave up waiting (nothing written) and the
                                   launch-per-date form produced this result instead */
} hh_lsm_result;
size_t hh_lsm_grid_elems(uint64_t n_paths, uint32_t n_steps, int32_t antithetic);
int hh_lsm_solve(hh_ctx* ctx, const hh_model* model, const hh_config* cfg, int32_t degree,
                 double step_discount, hh_lsm_result* out, int32_t* stop_time, double* stop_value,
                 double* spot_grid);
/*
 * The same backward induction on a spot grid the caller already holds in DEVICE memory
 * (grid[(n_steps+1)][n_paths], row k = date k·T/n_steps): the regression / stopping part of
 * least_squares_montecarlo.jl:107-134 alone, for any path source.  Uses model->strike, cp only.
 */
int hh_lsm_solve_grid(hh_ctx* ctx, const hh_model* model, const double* spot_grid_dev,
                      uint64_t n_paths, uint32_t n_steps, int32_t degree, double step_discount,
                      hh_lsm_result* out, int32_t* stop_time, double* stop_value);

/*
 * The same solve for an ensemble SHARDED over several devices (one process and one hh_ctx per
 * device, trajectories split by contiguous ranges as for hh_mc_accumulate).  The backward induction
 * needs sums over ALL trajectories at three points — the in-the-money statistics of every row, the
 * power sums of every row, and the moment sums Σ z^k y of each exercise date — so the solve is cut
 * exactly there: every call leaves a vector of LOCAL sums in device memory, the host all-reduces it
 * (SUM) over the ranks (torch.distributed / RCCL), and hands the GLOBAL vector to the next call:
 *
 *   hh_lsm_shard_begin(cfg of THIS shard)            ->  out [n_steps+1][3]
 *   hh_lsm_shard_phase(HH_LSM_PHASE_POW,  0, in, out):   in [n_steps+1][3]        out [n_steps+1][2D+1]
 *   hh_lsm_shard_phase(HH_LSM_PHASE_INIT, 0, in, out):   in [n_steps+1][2D+1]     out [D+1]  (row n_steps-1)
 *   for t = n_steps-1 .. 1:
 *   hh_lsm_shard_phase(HH_LSM_PHASE_STEP, t, in, out):   in [D+1] of row t        out [D+1] of row t-1
 *                                                                                  (nothing for t = 1)
 *   hh_lsm_shard_finish(accum_dev)                   ->  accum_dev[HH_ACC_LEN]: Σ, Σ² of the discounted
 *                                                        stopped values and the local trajectory count
 *   hh_lsm_finalize(all-reduced accumulator, host)   ->  price, std_error, n_paths_total
 *
 * in / out are device pointers (they may alias); hh_lsm_shard_xchg_elems() doubles hold the largest
 * of them.  With one rank (no all-reduce) the sequence reproduces hh_lsm_solve bit for bit.  The
 * optional host outputs of hh_lsm_shard_finish are this shard's stopping_info / spot rows.
 */
enum hh_lsm_phase { HH_LSM_PHASE_POW = 1, HH_LSM_PHASE_INIT = 2, HH_LSM_PHASE_STEP = 3 };
size_t hh_lsm_shard_xchg_elems(uint32_t n_steps, int32_t degree);
int hh_lsm_shard_begin(hh_ctx* ctx, const hh_model* model, const hh_config* cfg, int32_t degree,
                       double step_discount, double* out_dev);
int hh_lsm_shard_phase(hh_ctx* ctx, int32_t phase, uint32_t t, const double* in_dev, double* out_dev);
int hh_lsm_shard_finish(hh_ctx* ctx, double* accum_dev, int32_t* stop_time, double* stop_value,
                        double* spot_grid, uint32_t* rows_regressed, uint32_t* rows_skipped);
int hh_lsm_finalize(const double* accum_host, hh_lsm_result* out);
/* Diagnostics: the 8 phase totals (ticks of the 100 MHz constant clock) that a library built with
 * -DHH_LSM_STAMPS=1 leaves behind a persistent LSM solve of that shape; zeros from the shipped build. */
int hh_lsm_debug_read(hh_ctx* ctx, uint64_t n_paths_total, uint32_t n_steps, int32_t degree,
                      double* out8);

/*
 * hh_lsm_solve (least_squares_montecarlo.jl:99-136) with the trajectories sharded over the devices of
 * mg, in ONE call: the phased induction of hh_lsm_shard_* on every device, the vectors of local sums
 * all-reduced INSIDE the library between consecutive phases (2 + (n_steps − 1) + 1 exchanges of at most
 * (n_steps + 1)·(2·degree + 1) doubles: ncclAllReduce in place on the shards' streams, or — host-sum
 * contexts — copied back, added in the order g = 0 … G−1 and handed out again).  Unlike the European
 * solve this path exchanges IN PLACE between its phases, so the local sums go with a collective that
 * fails in the middle: the communicators are aborted and the streams retired as described above, and
 * the whole induction is run again with the host's ordered sum (HH_MGPU_AUTO) or the call returns
 * HH_ERR_RCCL (HH_MGPU_RCCL).  cfg: the whole ensemble with HOST seeds, as for hh_lsm_solve; every device must get
 * at least one trajectory.  stop_time / stop_value (nullable, host): the reference's stopping_info in
 * the whole-ensemble order of hh_lsm_solve (n_paths entries, then the n_paths mirrored ones).  Same
 * regression sums up to their summation order: identical stopping decisions except where a payoff
 * equals its fitted continuation value to rounding.
 */
int hh_mgpu_lsm_solve(hh_mgpu* mg, const hh_model* model, const hh_config* cfg, int32_t degree,
                      double step_discount, hh_lsm_result* out, int32_t* stop_time, double* stop_value);

/*
 * Per-date EXACT Heston paths: the NoiseProblem that sde_problem(::PricingProblem, ::HestonDynamics,
 * ::HestonBroadieKaya) builds (src/pricing_methods/montecarlo.jl:209-231) on the HestonNoise process
 * (src/distributions/heston.jl:82-91), stepped with dt = T/n_steps by simulate_paths (:342-353):
 * every step draws (log S, V) at t+dt from LogHestonDistribution(S_t, V_t, κ, θ, σ, ρ, r, dt) —
 * one Broadie–Kaya transition per date and trajectory.
 * cfg: dynamics = HH_HESTON, strategy = HH_BROADIE_KAYA, noise_mode = HH_NOISE_GENERATE, n_steps,
 * n_paths, seeds (ONE PER TRAJECTORY here, montecarlo.jl:331 — unlike the one-shot terminal law,
 * which reads seeds[0] only), bk_* controls; no antithetic form, no dual partials.
 * Outputs (nullable; host, or device when grids_on_device): spot_grid[(n_steps+1)][n_paths] =
 * exp(log S) rows (row 0 = S0) and var_grid of the same shape (row 0 = V0);
 * hh_lsm_grid_elems(n_paths, n_steps, 0) doubles each.  out (nullable): the bk_* counters summed
 * over all transitions, n_paths_done, kernel_ms, total_ms; price fields are zero.
 * hh_lsm_solve accepts the same (dynamics, strategy) pair and regresses on the spot rows.
 */
int hh_heston_exact_grid(hh_ctx* ctx, const hh_model* model, const hh_config* cfg, double* spot_grid,
                         double* var_grid, int32_t grids_on_device, hh_result* out);

/*
 * REPLAY increments.  Tile-major layout (what the step kernels stream):
 *     dW[tile][step][comp][HH_TILE_PATHS],  tile = path / 256, comp < ncomp (1 lognormal, 2 Heston),
 * the last tile zero-padded.  hh_replay_elems() = ceil(n_paths/256)·n_steps·ncomp·256 doubles.
 * Path-major layout (what the reference's saved noise gives per trajectory, montecarlo.jl:258,370):
 *     dW[path][step][comp]  (n_paths·n_steps·ncomp doubles), replay_layout = HH_REPLAY_PATH_MAJOR.
 * hh_mc_solve / hh_mc_accumulate STREAM such a buffer directly (no repacked copy) whenever a
 * trajectory's row is a multiple of 16 bytes — n_steps·ncomp even: every Heston shape, lognormal with an
 * even step count; the rows may start on any 16-byte boundary.  A device-resident buffer
 * (replay_on_device) is then read in place WHILE THE KERNEL RUNS: it must stay untouched until the
 * call's stream has completed (hh_mc_solve returns after that; after hh_mc_accumulate, synchronize) and
 * must not alias `terminal`.  A host buffer is copied to the ctx's staging buffer first, as always.  Only
 * the remaining case (lognormal Euler with an odd n_steps, and the one-normal-per-trajectory buffer of the
 * exact law when it is declared path-major) is repacked into the tile-major layout first, by the kernel
 * behind hh_replay_pack(), which callers may also use themselves.
 * For Heston the increments are the CORRELATED ones, cov = dt·[1 ρ; ρ 1] (heston.jl:18-20).
 * Exact lognormal law (HH_EXACT_LAW) in REPLAY mode: `replay` holds ONE standard normal per
 * trajectory, n_paths doubles (x = μ + σ̃·z, montecarlo.jl:302,413); no padding required.
 * Broadie–Kaya (HH_BROADIE_KAYA) in REPLAY mode: `replay` holds the three draws the reference makes
 * per trajectory, in its order (heston.jl:246-259) — V_T = c·rand(NoncentralChisq(d, λ)) (:125-133),
 * the uniform u of sample_from_cf (sample_from_cf.jl:29) and the normal Z of sample_log_S_T
 * (heston.jl:296) — as [V_T | u | Z], 3·n_paths doubles.  Everything downstream of the draws
 * (moments_from_cf, cdf_from_cf, inverse_cdf, log S_T) then runs on the reference's own numbers.
 */
size_t hh_replay_elems(uint64_t n_paths, uint32_t n_steps, int32_t dynamics);
int hh_replay_pack(hh_ctx* ctx, int32_t dynamics, uint64_t n_paths, uint32_t n_steps,
                   const double* src_path_major, int32_t src_on_device, double* dst_tile_major_dev);
/*
 * Fill a tile-major REPLAY buffer on the device with exactly the increments GENERATE mode would
 * draw (Philox4x32-10 keyed by seeds[i], counter = step; Box–Muller; lower-triangular correlation).
 */
int hh_wiener_fill(hh_ctx* ctx, int32_t dynamics, double rho, double T, uint32_t n_steps,
                   uint64_t n_paths, const uint64_t* seeds, int32_t seeds_on_device,
                   double* dst_tile_major_dev);

/*
 * Measurement hooks (the reference has no tracing; the build supplies its own).  When
 * enabled, every hh_mc_accumulate / hh_mc_solve brackets EVERYTHING IT ENQUEUES — the simulation
 * kernels and, where it is a kernel of its own, the record reduction; not the staging copies in front —
 * with HIP events on the ctx stream (one slot per call; hh_mc_accumulate_multi on Broadie–Kaya: one per
 * model — a chain's, then each finish pass's); hh_ctx_read_timings
 * synchronizes the stream and returns the elapsed ms of the slots recorded since the last read
 * (at most 256 are kept).
 */
int hh_ctx_enable_timing(hh_ctx* ctx, int32_t on);
int hh_ctx_read_timings(hh_ctx* ctx, double* ms, int32_t cap, int32_t* n_out);

/*
 * A SimulationConfig's seed vector in the device memory of ctx, uploaded once and kept — repeated solves on one
 * config (finite-difference Greeks, calibration loops) would otherwise move 8 MB per 10^6 trajectories at
 * every call, about a third of a GENERATE solve of that size.  The reference reads seeds[i] at every solve
 * (montecarlo.jl:331), so the cache is CONTENT-addressed: an entry is found by the vector's length and a
 * fingerprint of ALL its elements (hh_seeds_fingerprint, ~1 ms per 10^6 seeds) — a vector changed in place, or
 * another vector at the same address, is simply another key and is uploaded; the same numbers at another
 * address hit.  `fingerprint` = 0 lets the call compute it; a host whose vector cannot change (the Python
 * mirror freezes its copy) computes it once and passes it.  At most HH_SEED_CACHE_ENTRIES vectors are kept per
 * context, the least recently used one goes first, all go with the context.  *dev_out stays valid until a
 * later hh_seeds_cache call of this context MISSES (an eviction frees it): ask right before every solve,
 * pass the pointer as cfg->seeds with seeds_on_device = 1, do not keep it.  The upload is synchronous — the
 * host vector is free to change when the call returns.  The entry the call BEFORE a miss returned is never the one
 * evicted, so two host threads that share a context and each look up, then solve, cannot free each other's vector
 * between the two steps; more than two such threads serialise lookup + solve themselves (a context is one stream:
 * they gain nothing from running side by side).
 */
#define HH_SEED_CACHE_ENTRIES 8
uint64_t hh_seeds_fingerprint(const uint64_t* seeds, uint64_t n);  /* pure host arithmetic; never 0 */
int hh_seeds_cache(hh_ctx* ctx, const uint64_t* seeds_host, uint64_t n, uint64_t fingerprint,
                   const uint64_t** dev_out);
int hh_seeds_cache_stats(hh_ctx* ctx, uint64_t* hits, uint64_t* uploads, uint64_t* evictions); /* each nullable */

/* Device memory helpers for hosts without another allocator (the Julia wrapper). */
int hh_device_malloc(hh_ctx* ctx, size_t bytes, void** out_dev);
int hh_device_free(hh_ctx* ctx, void* dev);
int hh_memcpy_h2d(hh_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int hh_memcpy_d2h(hh_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int hh_ctx_synchronize(hh_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* HEDGEHOG_MC_H */
