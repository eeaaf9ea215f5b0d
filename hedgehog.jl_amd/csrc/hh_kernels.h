// Kernel argument blocks and launchers shared between the C-ABI translation unit (hh_api.hip) and
// the kernel translation units.  Everything here is internal; the public surface is
// include/hedgehog_mc.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hedgehog_mc.h"

namespace hh {

constexpr int kTile = HH_TILE_PATHS;  // paths per tile == paths per workgroup
constexpr int kRecStride = HH_ACC_LEN;
// internal record slots of the simulation / basket kernels (never part of the public vector)
// what every word of the self-reducing launches' record buffer holds between launches (hh_sim.h, finish_records):
// a quiet NaN no arithmetic produces; both 32-bit halves alike, so that hipMemsetD32 can write it
constexpr unsigned long long kPoison = 0x7FF8C0DE7FF8C0DEull;
constexpr unsigned long long kFinishSpinTicksDefault = 500000000ull;  // 5 s of the 100 MHz clock
constexpr int kRecItmS = 11;  // Σ 1[itm]·cp·S   (kRecItmS + 1: Σ 1[itm]·cp)

// value + P partials (forward-mode dual number; ForwardDiff.Dual{Tag,Float64,P} on the reference
// side, greeks_problem.jl:260).  P = 0 keeps a 1-element dummy that is never touched.
template <int P>
struct DualT {
  double v;
  double d[P > 0 ? P : 1];
};

// What is carried per path and how the requested directions are assembled from it.  Derivative
// propagation is linear in the seeds, so a requested direction k splits into
//  * its components along the parameters that reach the variance / diffusion — V0, κ, θ, σ for the
//    Heston Euler scheme, σ alone otherwise.  Only these BASIS derivatives are carried per path
//    (unit seed each, at most 4 slots whatever n_partials is): Σ∂p_k += w[k][j] · Σ(∂p/∂basis_j);
//  * its passive part (spot, drift rate, strike): ∂x_T is one constant for all paths, finished in
//    closed form:  Σ∂p_k += xdT_k · Σ 1[itm]·cp·S  −  dK_k · Σ 1[itm]·cp.
constexpr int kMaxBasis = 4;
enum { kBasisV0 = 0, kBasisKappa = 1, kBasisTheta = 2, kBasisSigma = 3 };
struct PartialMap {
  int sim;          // 1: records of a simulation/basket kernel (slots kRecItmS.. are internal)
  int n;            // n_partials of the call
  int n_active;     // carried basis derivatives
  int basis[kMaxBasis];                  // carried slot j -> parameter
  double w[HH_MAX_PARTIALS][kMaxBasis];  // seed of direction k on the parameter of slot j
  double xdT[HH_MAX_PARTIALS];           // passive part: ∂ log S_T / ∂θ_k from the spot and rate seeds
  double dK[HH_MAX_PARTIALS];            // strike seed of direction k
};
// Model/problem block passed BY VALUE as the kernel argument (every field is wave-uniform, so it
// is read through scalar loads).
template <int P>
struct SimArgs {
  DualT<P> x0;      // log S0                     (montecarlo.jl:182,201)
  DualT<P> v0;      // V0
  DualT<P> kappa, theta, sigma;
  DualT<P> r;       // drift rate
  DualT<P> gdrift;  // lognormal: r - sigma^2/2   (heston.jl:35)
  DualT<P> strike;
  DualT<P> law_mu, law_sd;  // exact lognormal law x = mu + sd·z (montecarlo.jl:302)
  double dt, sqrt_dt, rho, rho_c, cp;
  uint64_t n_paths;
  uint64_t path_offset;
  uint32_t n_steps;
  uint32_t n_tiles;       // workgroups of the launch = records it leaves (sim_records)
  uint32_t tail_from;     // REPLAY: tiles from here on pipeline deeper (the grid's tail)
  const uint64_t* seeds;  // device
  const double* replay;   // device, tile-major
  double* terminal;       // device or nullptr
  double* terminal_d;     // device or nullptr: dS_T/dθ_k at [k * n_total + index]
  double* records;        // device, [n_tiles][kRecStride]
  // the record reduction folded into this launch (hh_sim.h, finish_records): the last tile's workgroup
  // leaves the reduced HH_ACC_LEN-double accumulator vector in `accum`.  nullptr: records only.
  double* accum;          // device; then `records` is a buffer that holds kPoison between launches
  double acc_n_paths;     // what goes into slot HH_ACC_NPATHS
  // … its reducer: the workgroup of tile `reducer_tile` (the last one; the first under HH_OPT_FINISH_TILE_FIRST),
  // which waits at most `finish_spin_ticks` of the 100 MHz clock for a record and, when it gives up, says so in
  // finish_state[0] — a word of the context that stays set until the host has put the record buffer right again
  unsigned int* finish_state;
  unsigned long long finish_spin_ticks;
  uint32_t reducer_tile;
  PartialMap map;         // how the requested directions come out of the carried ones (map.n = n_partials)
};

struct DevicePtrs {
  const uint64_t* seeds;
  const double* replay;
  double* terminal;
  double* terminal_d;
  double* records;
  double* accum;          // non-NULL: the simulation kernel also reduces its records into this vector, and
                          // `records` is a poisoned buffer (hh_sim.h, finish_records)
  unsigned int* finish_state;       // … with the context's give-up word (SimArgs::finish_state), its wait bound in
  long long finish_spin_ticks = -1; //   ticks (< 0: kFinishSpinTicksDefault) and which tile's workgroup reduces
  int finish_tile_first = 0;        //   (0: the last)
  void* bk_scratch;  // Broadie–Kaya: bk_scratch_bytes() of device memory
  // Broadie–Kaya: which Bessel tables bk_scratch holds (NULL = unknown, always uploaded): the owner of
  // bk_scratch keeps one BkTableKey next to it, zero-initialised
  struct BkTableKey* bk_table_key;
  int bk_term_cache;  // Broadie–Kaya: cached series terms per trajectory at most (0 = default)
  void* bk_sort;      // exact grid: bk_grid_sort_bytes(pairs of a chain) of device memory — the chain then runs its
                      // pairs in the order of their Bessel arguments (NULL: in their natural order; same grid)
  // Euler REPLAY: `replay` is the reference's own layout dW[path][step][comp] (rows of a multiple of
  // 16 bytes: replay_direct_path_major()) and is consumed as it stands by euler_pm_kernel
  bool replay_path_major;
};
// can path-major increments of this shape be streamed without the repack pass?
inline bool replay_direct_path_major(uint32_t n_steps, int ncomp) { return ((uint64_t)n_steps * ncomp) % 2 == 0; }
struct BkTableKey {
  const void* where;        // address of the tables inside the scratch buffer
  double nu;                // Bessel order they were made for
  double kappa, sigma2, T;  // … and the model constants of the ϕ(0) block beside them (hh_bk.hip, CfZero)
};

// several payoffs on ONE set of terminal samples (basket.jl:35-38, same-expiry payoffs)
#ifndef HH_BASKET_CHUNK
#define HH_BASKET_CHUNK 4096
#endif
constexpr int kBasketChunk = HH_BASKET_CHUNK;  // trajectories per workgroup of basket_payoff_kernel
struct BasketArgs {
  const double* terminal;    // [n_paths] (+ [n_paths] mirrored)
  const double* terminal_d;  // [P][n_total] or nullptr
  const double* strikes;     // [n_payoffs]
  const double* cps;         // [n_payoffs]
  uint64_t n_paths;
  uint32_t n_chunks;
  int antithetic;
  double* records;           // [n_payoffs][n_chunks][kRecStride]
};
inline uint32_t basket_chunks(uint64_t n_paths) {
  return (uint32_t)((n_paths + kBasketChunk - 1) / kBasketChunk);
}

inline uint32_t tiles_for(uint64_t n_paths) { return (uint32_t)((n_paths + kTile - 1) / kTile); }
// The exact-law kernels (one normal, one exp per trajectory — no time loop to amortise a workgroup's reduction
// over) give a lane ONE pair of trajectories in a small ensemble and kExactPairs / kExactPairsHuge pairs in a
// large / huge one: a workgroup of 256 lanes then leaves ONE record per 512 / 4096 / 32768 trajectories, the
// chip always has >= 2048 workgroups to run and the reducer at most a few thousand records to add.  The form is
// a function of the shard's n_paths alone, so a result is reproducible for a given (n_paths, sharding) as
// everywhere else.
#ifndef HH_EXACT_PAIRS_SMALL
#define HH_EXACT_PAIRS_SMALL 4
#endif
constexpr int kExactPairs = 8, kExactPairsHuge = 64;
constexpr int kExactPairsSmall = HH_EXACT_PAIRS_SMALL;  // below 2048·512·8 trajectories (hh_kernels.hip says why)
// (defined in hh_kernels.hip: $HEDGEHOG_MC_EXACT_PAIRS overrides it for a measurement — tools/exact_pairs_ab.py)
int exact_pairs_per_lane(uint64_t n_paths);
inline uint32_t exact_records(uint64_t n_paths) {
  const uint64_t per = 512ull * (uint64_t)exact_pairs_per_lane(n_paths);
  return (uint32_t)((n_paths + per - 1) / per);
}
// records a simulation launch of this shape leaves (what its in-kernel reducer, or reduce_records_kernel, adds)
inline uint32_t sim_records(const hh_config& c) {
  return c.strategy == HH_EXACT_LAW ? exact_records(c.n_paths) : tiles_for(c.n_paths);
}
// carried basis derivatives (at most 4: V0, κ, θ, σ) -> instantiated kernel width
inline int pad_partials(uint32_t p) { return p <= 4 ? (int)p : 4; }

// All launchers return a hipError_t as int (0 = success) and only enqueue work on `s`.
int launch_simulation(const hh_model& m, const hh_config& c, const DevicePtrs& p, hipStream_t s);
// The same for n_models (2, 3 or 4) models stepped on the SAME draws in one pass (hh_multi.hip): p[k] carries
// model k's outputs (records, accum, terminal); seeds / replay are read from p[0].  c.n_partials must be 0;
// Euler–Maruyama and the exact lognormal law (Broadie–Kaya: one chain per model, by the caller).
int launch_simulation_multi(const hh_model* models, int n_models, const hh_config& c, const DevicePtrs* p,
                            hipStream_t s);
constexpr int kMaxModelsPerPass = 4;
// kernel argument block of a model that carries no derivatives (shared with hh_multi.hip)
SimArgs<0> make_args0(const hh_model& m, const hh_config& c, const DevicePtrs& p);
// One transition of the per-date exact Heston grid (HestonNoise, heston.jl:82-91): start state of
// every trajectory read from in_*, end state written to out_* (n_paths doubles each, device);
// m.T is the length of the transition.  NULL = the one-shot terminal law of launch_bk.
struct BkTransition {
  const double* in_spot;
  const double* in_var;
  double* out_spot;
  double* out_var;
  uint32_t step;
};
// upload_tables = false: the Bessel tables of this (κ, θ, σ) are already in p.bk_scratch from an earlier
// launch_bk on the SAME scratch buffer and trajectory count (the dates of one exact grid)
// Two models whose one-shot Broadie–Kaya chains are the same chain: nothing the variance process and the inversion
// see differs (κ, θ, σ, V0, T) — a bumped spot, rate, ρ or strike.  launch_bk_refinish finishes such a model from the
// ∫V the chain of the other one left in the SAME scratch (same n_paths, same draws), with records that are those of a
// chain of its own bit for bit; rec0 = the records of the model whose chain ran (the counters are copied from there).
bool bk_same_chain(const hh_model& a, const hh_model& b);
int launch_bk_refinish(const hh_model& m, const hh_config& c, const DevicePtrs& p, const double* rec0, hipStream_t s);
int launch_bk(const hh_model& m, const hh_config& c, const DevicePtrs& p, hipStream_t s,
              const BkTransition* tr = nullptr, bool upload_tables = true);
// Dates k0 … k0 + n_dates of an exact grid in ONE chain: the variance rows first (sequential in the date, cheap),
// then the CF inversions of all (date, trajectory) pairs at once — they are independent given the variances —
// then the spot rows chained date by date.  spot_rows / var_rows = row k0 of the grids ([date][c.n_paths]);
// rows k0+1 … k0+n_dates are written.  Scratch: bk_scratch_bytes(c.n_paths · n_dates); records:
// bk_record_count(c.n_paths · n_dates) (payoff sums zero, the counters of all pairs).  Same draws, same
// arithmetic per pair as n_dates launch_bk transitions: bit-identical rows.
int launch_bk_grid(const hh_model& m, const hh_config& c, const DevicePtrs& p, hipStream_t s,
                   double* spot_rows, double* var_rows, uint32_t k0, uint32_t n_dates, bool upload_tables);
// dates per chain for a grid of n_steps dates: all of them unless the pairs' term cache would pass its budget
uint32_t bk_grid_dates_per_chain(uint64_t n_paths, uint32_t n_steps, int term_cache);
size_t bk_grid_sort_bytes(uint64_t n_chain);
constexpr int kBkTermCacheDefault = 256;
size_t bk_scratch_bytes(uint64_t n_paths, int term_cache = 0);
// where the last chain over n_paths trajectories left, per trajectory, its decision word (BkDecision bits:
// secant evaluations | branch << 8 | bisection iterations << 16 | long-series bit 31) and its series length
void bk_diag_ptrs(const void* scratch, uint64_t n_paths, int term_cache, const uint32_t** decisions,
                  const uint32_t** series_len);
uint32_t bk_record_count(uint64_t n_paths);  // records the Broadie–Kaya chain can write (the CF kernel's tiles + the tail kernel's workgroups)
// device word that holds, once the chain over n_paths trajectories (or pairs) has run in `scratch`, how many of them
// it DID write — the reduction reads that many (launch_reduce_records, n_records_dev)
const uint32_t* bk_live_records(const void* scratch, uint64_t n_paths);
// the four Broadie–Kaya counter slots of `src` (HH_ACC_LEN doubles) into each of n_groups accumulators
int launch_copy_bk_counters(const double* src, double* accum, uint32_t n_groups, hipStream_t s);
// row 0 of the exact Heston grid: spot0[i] = S0, var0[i] = V0
int launch_fill_rows(double* spot0, double* var0, uint64_t n, double S0, double V0, hipStream_t s);
// m, c given: finish the dual partials (active slots re-ordered, passive ones in closed form)
// n_records_dev (device, optional): the launch reads min(*n_records_dev, n_records) records instead of n_records
int launch_reduce_records(const double* records, uint32_t n_records, double n_paths, double* accum,
                          hipStream_t s, uint32_t n_groups = 1, const hh_model* m = nullptr,
                          const hh_config* c = nullptr, bool basket = false,
                          const uint32_t* n_records_dev = nullptr);
int count_active_partials(const hh_model& m, const hh_config& c);
int launch_basket_payoffs(const BasketArgs& b, uint32_t n_payoffs, uint32_t n_active_partials,
                          hipStream_t s);
// Carr–Madan on the device (hh_fourier.hip)
int launch_carr_madan(const hh_model& m, int dynamics, int compat_sqrt_alpha, double alpha,
                      double bound, double* out_dev, hipStream_t s);
// n_payoffs Carr–Madan integrals in one launch (one workgroup each); per_payoff_dev = [4][n_payoffs]:
// log K, T, r_drift, discount; out_dev[n_payoffs] receives the CALL prices
// the same with the gradient: out_dev[n_payoffs][8] = call price, then its partials along
// S0, V0, kappa, theta, sigma, rho, r_drift
int launch_carr_madan_grad(const hh_model& m, int dynamics, int compat_sqrt_alpha, double alpha,
                           double bound, const double* per_payoff_dev, uint32_t n_payoffs,
                           double* out_dev, hipStream_t s);
int launch_carr_madan_basket(const hh_model& m, int dynamics, int compat_sqrt_alpha, double alpha,
                             double bound, const double* per_payoff_dev, uint32_t n_payoffs,
                             double* out_dev, hipStream_t s);
// LSM (hh_lsm.hip)
uint32_t lsm_chunks(uint64_t ntot);
size_t lsm_scratch_doubles(uint64_t ntot, uint32_t n_steps, int degree);
// The induction cut where it needs GLOBAL sums, for ensembles sharded over several devices: between
// two phases the host all-reduces (SUM) the device vector the phase left in vec_out and hands it to
// the next phase as vec_in.  Stats: - -> [rows][3];  Pow: [rows][3] -> [rows][2D+1];
// Init: [rows][2D+1] -> [D+1] (moment sums of row n_steps-1);  Step(t): [D+1] of row t -> [D+1] of
// row t-1 (nothing for t = 1);  Final: per-workgroup records of the discounted stopped values.
enum { kLsmPhaseStats = 0, kLsmPhasePow = 1, kLsmPhaseInit = 2, kLsmPhaseStep = 3, kLsmPhaseFinal = 4 };
int launch_lsm_phase(int phase, uint32_t t, const double* grid, uint64_t ntot, uint32_t n_steps,
                     double strike, double cp, double step_discount, int degree, int32_t* tau,
                     double* val, double* scratch, double* records, const double* vec_in,
                     double* vec_out, hipStream_t s);
int launch_gbm_grid(const uint64_t* seeds_dev, uint64_t n_paths, uint32_t n_steps, double S0,
                    double r, double sigma, double T, int anti, double* grid, hipStream_t s);
// form: kLsmFormPersistent = the whole backward induction in ONE launch when the ensemble fits the
// chip (else it falls back by itself), kLsmFormPerDate = one launch per exercise date.  *form_used
// says which was enqueued; after a persistent launch the caller synchronises and reads the word at
// lsm_persistent_status(scratch): non-zero = a workgroup gave up waiting (the grid was not
// co-resident) and nothing was written — run the per-date form instead.  Both forms give
// bit-identical results.
enum { kLsmFormPerDate = 0, kLsmFormPersistent = 1, kLsmFormAuto = 2 };  // auto: persistent above 2^18 trajectories
int launch_lsm(const double* grid, uint64_t ntot, uint32_t n_steps, double strike, double cp,
               double step_discount, int degree, int32_t* tau, double* val, double* scratch,
               double* records, hipStream_t s, int form, int* form_used,
               unsigned long long spin_ticks = 100000000ull /* 1 s of the 100 MHz constant clock */);
const unsigned int* lsm_persistent_status(const double* scratch);
constexpr int kLsmStampSlotsApi = 8;  // doubles behind the two row counters at the end of the scratch
int launch_wiener_fill(int dynamics, double rho, double sqrt_dt, uint32_t n_steps, uint64_t n_paths,
                       const uint64_t* seeds_dev, double* dst, hipStream_t s);
int launch_replay_pack(int ncomp, uint64_t n_paths, uint32_t n_steps, const double* src_dev,
                       double* dst_dev, hipStream_t s);

}  // namespace hh
