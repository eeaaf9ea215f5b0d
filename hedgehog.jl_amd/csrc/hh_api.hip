// C-ABI of libhedgehog_mc.so (include/hedgehog_mc.h): context, argument checking, staging of
// caller buffers, kernel sequencing.  No arithmetic of the pricing path happens on the host except
// hh_mc_finalize's discount·mean (montecarlo.jl:489-490) on the reduced accumulator vector.
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "hh_kernels.h"

#include "hh_ctx.h"

namespace {

const char* const kNoCtx = "hedgehog_mc: no context";

int ncomp_of(int dynamics) { return dynamics == HH_HESTON ? 2 : 1; }

// launch limits (documented in hedgehog_mc.h): threads per launch < 2^32, grid.y <= 65535
constexpr uint64_t kMaxPaths = (1ull << 32) - 256;
constexpr uint32_t kMaxEulerSteps = 4u * 65535u;  // replay_pack_kernel: 4 steps per grid.y
constexpr uint32_t kMaxGridSteps = 65534u;        // LSM / exact grid: one grid.y per row, n_steps + 1 rows

// degrees of freedom of the noncentral chi-squared law of V_T (heston.jl:128): the Bessel order is
// d/2 - 1, which must stay strictly above -1 in fp64 and within what the tables of hh_bessel.h cover
// … and the constants of a transition of length dt (heston.jl:129-130, :170-172) must come out finite:
// kappa·dt beyond ±700 overflows e^{∓κ dt}
static bool bk_law_ok(const hh_model* m, double dt) {
  const double s2 = m->sigma * m->sigma;
  const double d = 4.0 * m->kappa * m->theta / s2;
  if (!(d >= 1e-8 && d <= 1e6)) return false;
  const double em1 = -std::expm1(-m->kappa * dt);
  const double lam_per_v = 4.0 * m->kappa * std::exp(-m->kappa * dt) / (s2 * em1);  // λ / V0
  const double cscale = s2 * em1 / (4.0 * m->kappa);
  const double nuk = 4.0 * m->kappa * std::exp(-0.5 * m->kappa * dt) / s2 / em1;
  return std::isfinite(lam_per_v) && lam_per_v >= 0.0 && std::isfinite(cscale) && cscale > 0.0 &&
         std::isfinite(nuk) && nuk > 0.0 && std::isfinite(m->kappa * (1.0 + std::exp(-m->kappa * dt)) / em1);
}

int validate(hh_ctx* ctx, const hh_model* m, const hh_config* c) {
  if (!m || !c) return fail(ctx, HH_ERR_INVALID, "model/config is NULL");
  if (c->n_paths == 0) return fail(ctx, HH_ERR_INVALID, "n_paths must be >= 1");
  // one 256-thread workgroup per 256 trajectories: HIP rejects a launch of 2^32 threads or more
  if (c->n_paths > kMaxPaths)
    return fail(ctx, HH_ERR_INVALID, "n_paths too large (max 2^32 - 256 per call; shard the ensemble)");
  // grid.y of the REPLAY repack / fill kernels (4 and 16 steps per workgroup) is limited to 65535
  if (c->strategy == HH_EULER_MARUYAMA && c->n_steps > kMaxEulerSteps)
    return fail(ctx, HH_ERR_INVALID, "n_steps too large (max %u)", kMaxEulerSteps);
  if (c->n_partials > HH_MAX_PARTIALS)
    return fail(ctx, HH_ERR_INVALID, "n_partials %u > HH_MAX_PARTIALS", c->n_partials);
  const bool logn = c->dynamics == HH_LOGNORMAL, hest = c->dynamics == HH_HESTON;
  if (!logn && !hest) return fail(ctx, HH_ERR_INVALID, "unknown dynamics %d", c->dynamics);
  // the (dynamics, strategy) pairs that have a sde_problem / marginal_law method in the reference
  // (montecarlo.jl:140-231, 293-320); anything else is a MethodError there
  const bool ok = (logn && (c->strategy == HH_EULER_MARUYAMA || c->strategy == HH_EXACT_LAW)) ||
                  (hest && (c->strategy == HH_EULER_MARUYAMA || c->strategy == HH_BROADIE_KAYA));
  if (!ok)
    return fail(ctx, HH_ERR_UNSUPPORTED, "no simulation method for dynamics %d with strategy %d",
                c->dynamics, c->strategy);
  if (!(m->S0 > 0.0) || !std::isfinite(m->S0)) return fail(ctx, HH_ERR_INVALID, "S0 must be > 0");
  if (!(m->T > 0.0) || !std::isfinite(m->T)) return fail(ctx, HH_ERR_INVALID, "T must be > 0");
  if (m->cp != 1.0 && m->cp != -1.0) return fail(ctx, HH_ERR_INVALID, "cp must be +1 or -1");
  if (hest && !(std::fabs(m->rho) <= 1.0)) return fail(ctx, HH_ERR_INVALID, "|rho| must be <= 1");
  // the scalars the chosen dynamics reads (the reference would carry a NaN through to the price; here it
  // is an argument error, before any launch)
  if (!std::isfinite(m->sigma) || !std::isfinite(m->r_drift) || !std::isfinite(m->discount) ||
      !std::isfinite(m->strike) ||
      (hest && (!std::isfinite(m->V0) || !std::isfinite(m->kappa) || !std::isfinite(m->theta))))
    return fail(ctx, HH_ERR_INVALID, "model scalars must be finite");
  if (c->strategy == HH_EULER_MARUYAMA && c->n_steps == 0)
    return fail(ctx, HH_ERR_INVALID, "n_steps must be >= 1 for EulerMaruyama");
  if (c->strategy == HH_BROADIE_KAYA) {
    // final_sample(law, sample, ::Antithetic) needs mean(law), which LogHestonDistribution does
    // not define (montecarlo.jl:387); dual numbers do not pass rand! (heston.jl:261-276)
    if (c->antithetic)
      return fail(ctx, HH_ERR_UNSUPPORTED, "HestonBroadieKaya has no antithetic form");
    if (c->n_partials)
      return fail(ctx, HH_ERR_UNSUPPORTED, "HestonBroadieKaya does not carry dual partials");
    if (m->sigma == 0.0 || m->kappa == 0.0 || !(m->V0 > 0.0))
      return fail(ctx, HH_ERR_INVALID, "HestonBroadieKaya needs sigma != 0, kappa != 0, V0 > 0");
    // d = 4κθ/σ² degrees of freedom of the noncentral chi-squared law (heston.jl:128), λ >= 0 (:129)
    if ((c->bk_root_form != HH_BK_ROOT_SECANT && c->bk_root_form != HH_BK_ROOT_ORDER2) ||
        (c->bk_bracket_form != HH_BK_BRACKET_MIDPOINT && c->bk_bracket_form != HH_BK_BRACKET_ROOTS) ||
        (c->bk_caps != HH_BK_CAPS_AS_WRITTEN && c->bk_caps != HH_BK_CAPS_ROOTS_DEFAULT))
      return fail(ctx, HH_ERR_INVALID, "bk_root_form, bk_bracket_form, bk_caps: 0 or 1 each (hedgehog_mc.h)");
    if (!bk_law_ok(m, m->T))
      return fail(ctx, HH_ERR_INVALID, "HestonBroadieKaya needs 1e-8 <= d = 4 kappa theta / sigma^2 <= 1e6 "
                                       "(Bessel order d/2 - 1 strictly above -1, tables up to order 5e5) and "
                                       "finite transition constants (|kappa T| < 700)");
  }
  const bool euler = c->strategy == HH_EULER_MARUYAMA;
  if (c->noise_mode == HH_NOISE_REPLAY) {
    if (!c->replay) return fail(ctx, HH_ERR_INVALID, "REPLAY needs a replay buffer");
    if (((uintptr_t)c->replay & 15u) != 0)
      return fail(ctx, HH_ERR_INVALID, "replay buffer must be 16-byte aligned");
    if (c->replay_len) {  // operand shape: what the kernels (or the packer) will index
      uint64_t need = c->n_paths;  // exact law: one normal per trajectory
      if (c->strategy == HH_BROADIE_KAYA) need = 3 * c->n_paths;  // V_T | u | Z
      if (euler)
        need = c->replay_layout == HH_REPLAY_PATH_MAJOR
                   ? c->n_paths * (uint64_t)c->n_steps * (uint64_t)ncomp_of(c->dynamics)
                   : (uint64_t)hh::tiles_for(c->n_paths) * c->n_steps * ncomp_of(c->dynamics) *
                         hh::kTile;
      if (c->replay_len < need)
        return fail(ctx, HH_ERR_INVALID, "replay buffer holds %llu elements, %llu needed",
                    (unsigned long long)c->replay_len, (unsigned long long)need);
    }
  } else if (c->noise_mode == HH_NOISE_GENERATE) {
    if (!c->seeds) return fail(ctx, HH_ERR_INVALID, "GENERATE needs seeds");
    const uint64_t need = euler ? c->n_paths : 1;  // montecarlo.jl:65-66, 331, 456
    if (c->seeds_len && c->seeds_len < need)
      return fail(ctx, HH_ERR_INVALID, "Number of seeds (%llu) must be >= number of trajectories (%llu)",
                  (unsigned long long)c->seeds_len, (unsigned long long)need);
  } else {
    return fail(ctx, HH_ERR_INVALID, "unknown noise_mode %d", c->noise_mode);
  }
  return HH_OK;
}

// the record buffer of the launches that reduce their own records: every word kPoison (hh_sim.h) from the
// allocation on — each launch's reducer leaves it that way again
int ensure_poisoned(hh_ctx* ctx, size_t need) {
  if (need <= ctx->frecords_cap) return HH_OK;
  int rc = ensure(ctx, ctx->frecords, ctx->frecords_cap, need);
  if (rc) return rc;
  static_assert((hh::kPoison >> 32) == (hh::kPoison & 0xffffffffull), "hipMemsetD32 writes the pattern");
  HH_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->frecords, (int)(hh::kPoison & 0xffffffffull),
                                ctx->frecords_cap * 2, ctx->stream));
  return HH_OK;
}

// HH_OPT_FUSE_REDUCE = 2 (default), by what was measured (profiles/r05_c_fuse_ab.txt, same process, same box):
// the simulation kernel reduces its own records when they are few (the whole grid is resident at once, the
// reducer finds them all on its first look: -2 % on a 27 µs solve) and when the kernel runs long enough for
// the reducer's wait to disappear behind the other workgroups' work (Euler, 10^6 x 252: -0.3 % in both noise
// modes); a SHORT kernel with thousands of records (the exact law at 10^6-10^7 trajectories, 14-43 µs) is 1-4 %
// faster with reduce_records_kernel behind it, whose 16 workgroups read the records in parallel.
constexpr uint32_t kFuseAutoMaxRecords = 512, kFuseAutoMinSteps = 32;
bool fuse_for(const hh_ctx* ctx, const hh_config* c) {
  if (ctx->fuse_reduce != 2) return ctx->fuse_reduce == 1;
  return hh::sim_records(*c) <= kFuseAutoMaxRecords || (c->strategy == HH_EULER_MARUYAMA && c->n_steps >= kFuseAutoMinSteps);
}

// what a launch that reduces its own records needs of the context beside the poisoned buffer
void fused_controls(const hh_ctx* ctx, hh::DevicePtrs& p) {
  p.finish_state = ctx->finish_state;
  p.finish_spin_ticks = ctx->finish_spin_ticks;
  p.finish_tile_first = ctx->finish_tile_first;
}

// A reducer inside a simulation kernel gave up waiting for a record (hh_sim.h): the record buffer may hold that
// record now, where the next launch would take it for its own.  With the stream idle: every word back to the
// poison, the give-up word cleared.  Returns HH_OK when there was nothing to recover from.
int recover_finish(hh_ctx* ctx) {
  unsigned int state = 0;
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  HH_HIP(ctx, hipMemcpy(&state, ctx->finish_state, sizeof(state), hipMemcpyDeviceToHost));
  if (state == 0u) return HH_OK;
  if (ctx->frecords_cap)
    HH_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->frecords, (int)(hh::kPoison & 0xffffffffull),
                                  ctx->frecords_cap * 2, ctx->stream));
  HH_HIP(ctx, hipMemsetAsync(ctx->finish_state, 0, 2 * sizeof(unsigned int), ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const long long ticks = ctx->finish_spin_ticks < 0 ? (long long)hh::kFinishSpinTicksDefault : ctx->finish_spin_ticks;
  return fail(ctx, HH_ERR_DEVICE_TIMEOUT,
              "the record reduction inside a simulation kernel gave up: a workgroup's record had not arrived after %lld "
              "ticks of the 100 MHz clock (a queue preempted for that long, or a workgroup that died); the sums of that "
              "solve, and of solves queued behind it, are lost — the context's record buffer has been reset and the "
              "context can be used again", ticks);
}

// hh_mc_finalize refused an accumulator: because a reducer gave up (the named status), or for what it says
int finalize_failed(hh_ctx* ctx, int rc) {
  const int rf = recover_finish(ctx);
  return rf ? rf : fail(ctx, rc, "finalize failed: the accumulator holds no trajectories");
}

size_t replay_elems(uint64_t n_paths, uint32_t n_steps, int dynamics) {
  return (size_t)hh::tiles_for(n_paths) * n_steps * ncomp_of(dynamics) * hh::kTile;
}

}  // namespace

extern "C" {

int hh_abi_version(void) { return HH_ABI_VERSION; }

int hh_ctx_create(hh_ctx** out, int device_id) {
  if (!out) return HH_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return HH_ERR_HIP;  // no CPU fallback
  if (device_id < 0 || device_id >= n) return HH_ERR_INVALID;
  hh_ctx* ctx = new (std::nothrow) hh_ctx();
  if (!ctx) return HH_ERR_NOMEM;
  ctx->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_switch, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_stage, hipEventDisableTiming) != hipSuccess ||
      hipMalloc((void**)&ctx->accum, HH_ACC_LEN * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&ctx->finish_state, 2 * sizeof(unsigned int)) != hipSuccess ||
      hipMemset(ctx->finish_state, 0, 2 * sizeof(unsigned int)) != hipSuccess ||
      hipHostMalloc((void**)&ctx->accum_host, (HH_ACC_LEN + 8) * sizeof(double), hipHostMallocDefault) !=
          hipSuccess) {
    hh_ctx_destroy(ctx);
    return HH_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  *out = ctx;
  return HH_OK;
}

void hh_ctx_destroy(hh_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  // a borrowed stream may be gone already: wait for the device (hipFree below synchronises anyway)
  if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
  (void)hipDeviceSynchronize();
  if (ctx->records) (void)hipFree(ctx->records);
  if (ctx->seeds) (void)hipFree(ctx->seeds);
  if (ctx->replay) (void)hipFree(ctx->replay);
  if (ctx->replay_src) (void)hipFree(ctx->replay_src);
  if (ctx->terminal) (void)hipFree(ctx->terminal);
  if (ctx->terminal_d) (void)hipFree(ctx->terminal_d);
  if (ctx->payoffs) (void)hipFree(ctx->payoffs);
  if (ctx->basket_records) (void)hipFree(ctx->basket_records);
  if (ctx->basket_accum) (void)hipFree(ctx->basket_accum);
  if (ctx->bk_scratch) (void)hipFree(ctx->bk_scratch);
  if (ctx->bk_sort) (void)hipFree(ctx->bk_sort);
  if (ctx->lsm_grid) (void)hipFree(ctx->lsm_grid);
  if (ctx->heston_var) (void)hipFree(ctx->heston_var);
  if (ctx->lsm_val) (void)hipFree(ctx->lsm_val);
  if (ctx->lsm_tau) (void)hipFree(ctx->lsm_tau);
  if (ctx->lsm_scratch) (void)hipFree(ctx->lsm_scratch);
  for (auto& e : ctx->seed_cache)
    if (e.dev) (void)hipFree(e.dev);
  if (ctx->accum) (void)hipFree(ctx->accum);
  if (ctx->frecords) (void)hipFree(ctx->frecords);
  if (ctx->finish_state) (void)hipFree(ctx->finish_state);
  if (ctx->accum_host) (void)hipHostFree(ctx->accum_host);
  for (auto& pr : ctx->tev)
    for (auto& e : pr)
      if (e) (void)hipEventDestroy(e);
  if (ctx->ev_switch) (void)hipEventDestroy(ctx->ev_switch);
  if (ctx->ev_stage) (void)hipEventDestroy(ctx->ev_stage);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

// Work already queued on the stream being left still uses the ctx's scratch buffers (records, staged
// seeds / increments): the stream taken up next waits for it (an event, no host synchronisation).
static int switch_stream(hh_ctx* ctx, hipStream_t next) {
  if (next == ctx->stream) return HH_OK;
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hipEventRecord(ctx->ev_switch, ctx->stream));
  HH_HIP(ctx, hipStreamWaitEvent(next, ctx->ev_switch, 0));
  ctx->stream = next;
  return HH_OK;
}

int hh_ctx_set_stream(hh_ctx* ctx, void* hip_stream) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  return switch_stream(ctx, (hipStream_t)hip_stream);  // NULL is the device's default (null) stream
}

int hh_ctx_reset_stream(hh_ctx* ctx) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  return switch_stream(ctx, ctx->own_stream);
}

const char* hh_last_error(const hh_ctx* ctx) { return ctx ? ctx->err : kNoCtx; }

int hh_ctx_set_option(hh_ctx* ctx, int32_t option, int64_t value) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  switch (option) {
    case HH_OPT_LSM_FORM:
      if (value != HH_LSM_FORM_PER_DATE && value != HH_LSM_FORM_PERSISTENT && value != HH_LSM_FORM_AUTO)
        return fail(ctx, HH_ERR_INVALID, "HH_OPT_LSM_FORM: 0 (launch per date), 1 (one launch) or 2 (auto)");
      ctx->lsm_form = (int)value;
      return HH_OK;
    case HH_OPT_BK_TERM_CACHE:
      if (value < 8 || value > 1024)
        return fail(ctx, HH_ERR_INVALID, "HH_OPT_BK_TERM_CACHE: 8 .. 1024 series terms per trajectory");
      ctx->bk_term_cache = (int)value;
      return HH_OK;
    case HH_OPT_LSM_SPIN_TICKS:
      if (value < -1) return fail(ctx, HH_ERR_INVALID, "HH_OPT_LSM_SPIN_TICKS: ticks of the 100 MHz clock, 0 = give up at once, -1 = default");
      ctx->lsm_spin_ticks = value;
      return HH_OK;
    case HH_OPT_GRID_FORM:
      if (value != HH_GRID_FORM_PER_DATE && value != HH_GRID_FORM_BATCHED)
        return fail(ctx, HH_ERR_INVALID, "HH_OPT_GRID_FORM: 0 (one chain per date) or 1 (dates batched)");
      ctx->grid_form = (int)value;
      return HH_OK;
    case HH_OPT_GRID_ORDER:
      if (value != 0 && value != 1)
        return fail(ctx, HH_ERR_INVALID, "HH_OPT_GRID_ORDER: 0 (a chain's pairs in their natural order) or 1 (sorted by their Bessel arguments)");
      ctx->grid_order = (int)value;
      return HH_OK;
    case HH_OPT_FINISH_SPIN_TICKS:
      if (value < -1) return fail(ctx, HH_ERR_INVALID, "HH_OPT_FINISH_SPIN_TICKS: ticks of the 100 MHz clock, 0 = give up at once, -1 = default");
      ctx->finish_spin_ticks = value;
      return HH_OK;
    case HH_OPT_FINISH_TILE_FIRST:
      if (value != 0 && value != 1) return fail(ctx, HH_ERR_INVALID, "HH_OPT_FINISH_TILE_FIRST: 0 (the last tile's workgroup adds the records) or 1 (the first one's)");
      ctx->finish_tile_first = (int)value;
      return HH_OK;
    case HH_OPT_FUSE_REDUCE:
      if (value < 0 || value > 2)
        return fail(ctx, HH_ERR_INVALID, "HH_OPT_FUSE_REDUCE: 0 (a second kernel reduces the records), 1 (the simulation kernel does) or 2 (by size)");
      ctx->fuse_reduce = (int)value;
      return HH_OK;
    default:
      return fail(ctx, HH_ERR_INVALID, "unknown option %d", option);
  }
}

int hh_ctx_synchronize(hh_ctx* ctx) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return HH_OK;
}

size_t hh_replay_elems(uint64_t n_paths, uint32_t n_steps, int32_t dynamics) {
  return replay_elems(n_paths, n_steps, dynamics);
}

int hh_replay_pack(hh_ctx* ctx, int32_t dynamics, uint64_t n_paths, uint32_t n_steps,
                   const double* src, int32_t src_on_device, double* dst) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!src || !dst || n_paths == 0 || n_steps == 0 || n_paths > kMaxPaths || n_steps > kMaxEulerSteps)
    return fail(ctx, HH_ERR_INVALID, "hh_replay_pack: bad arguments");
  HH_HIP(ctx, hipSetDevice(ctx->device));
  const int nc = ncomp_of(dynamics);
  const double* src_dev = src;
  if (!src_on_device) {
    const size_t n = (size_t)n_paths * n_steps * nc;
    int rc = ensure(ctx, ctx->replay_src, ctx->replay_src_cap, n);
    if (rc) return rc;
    HH_HIP(ctx, hipMemcpyAsync(ctx->replay_src, src, n * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
    if ((rc = note_host_copy(ctx))) return rc;
    src_dev = ctx->replay_src;
  }
  HH_HIP(ctx, hh::launch_replay_pack(nc, n_paths, n_steps, src_dev, dst, ctx->stream));
  return release_host_operands(ctx);
}

int hh_wiener_fill(hh_ctx* ctx, int32_t dynamics, double rho, double T, uint32_t n_steps,
                   uint64_t n_paths, const uint64_t* seeds, int32_t seeds_on_device, double* dst) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!seeds || !dst || n_paths == 0 || n_steps == 0 || !(T > 0.0) || !(std::fabs(rho) <= 1.0) ||
      n_paths > kMaxPaths || n_steps > kMaxEulerSteps)
    return fail(ctx, HH_ERR_INVALID, "hh_wiener_fill: bad arguments");
  HH_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t* seeds_dev = seeds;
  if (!seeds_on_device) {
    int rc = ensure(ctx, ctx->seeds, ctx->seeds_cap, (size_t)n_paths);
    if (rc) return rc;
    HH_HIP(ctx, hipMemcpyAsync(ctx->seeds, seeds, n_paths * sizeof(uint64_t),
                               hipMemcpyHostToDevice, ctx->stream));
    if ((rc = note_host_copy(ctx))) return rc;
    seeds_dev = ctx->seeds;
  }
  const double dt = T / (double)n_steps;
  HH_HIP(ctx, hh::launch_wiener_fill(dynamics, rho, std::sqrt(dt), n_steps, n_paths, seeds_dev, dst,
                                     ctx->stream));
  return release_host_operands(ctx);
}

// Shared body of hh_mc_accumulate / hh_mc_accumulate_basket: stage caller buffers, run the
// simulation kernel (which also reduces the payoff of m->strike / m->cp into ctx->records) and
// leave the terminal samples in device memory when asked.
// The noise a simulation reads, in device memory: p.seeds (GENERATE) or p.replay (REPLAY), staged from the
// caller's host buffers where needed.  tile_major_only: path-major Euler increments are repacked even where the
// one-model kernel could stream them as they are (the several-model kernel reads the tile-major layout only).
static int stage_noise(hh_ctx* ctx, const hh_config* c, hh::DevicePtrs& p, bool tile_major_only) {
  int rc = HH_OK;
  const bool bk = c->strategy == HH_BROADIE_KAYA;
  // seeds: per-trajectory for Euler (montecarlo.jl:331), seeds[1] only for the exact laws (:456)
  if (c->noise_mode == HH_NOISE_GENERATE) {
    const size_t need = (c->strategy == HH_EULER_MARUYAMA) ? (size_t)c->n_paths : 1;
    if (c->seeds_on_device) {
      p.seeds = c->seeds;
    } else {
      rc = ensure(ctx, ctx->seeds, ctx->seeds_cap, need);
      if (rc) return rc;
      HH_HIP(ctx, hipMemcpyAsync(ctx->seeds, c->seeds, need * sizeof(uint64_t),
                                 hipMemcpyHostToDevice, ctx->stream));
      if ((rc = note_host_copy(ctx))) return rc;
      p.seeds = ctx->seeds;
    }
  } else if (bk) {
    // the trajectory's three draws [V_T | u | Z], n_paths each (heston.jl:246-259 order)
    const size_t n3 = (size_t)3 * c->n_paths;
    if (c->replay_on_device) {
      p.replay = c->replay;
    } else {
      rc = ensure(ctx, ctx->replay, ctx->replay_cap, n3);
      if (rc) return rc;
      HH_HIP(ctx, hipMemcpyAsync(ctx->replay, c->replay, n3 * sizeof(double), hipMemcpyHostToDevice,
                                 ctx->stream));
      if ((rc = note_host_copy(ctx))) return rc;
      p.replay = ctx->replay;
    }
  } else {
    const uint32_t steps = (c->strategy == HH_EULER_MARUYAMA) ? c->n_steps : 1;
    const int dyn = (c->strategy == HH_EULER_MARUYAMA) ? c->dynamics : HH_LOGNORMAL;
    const size_t tile_elems = replay_elems(c->n_paths, steps, dyn);
    if (c->replay_layout == HH_REPLAY_PATH_MAJOR && c->strategy == HH_EULER_MARUYAMA && !tile_major_only &&
        hh::replay_direct_path_major(steps, ncomp_of(dyn))) {
      // the reference's layout, streamed as it stands (euler_pm_kernel): no repack pass
      if (c->replay_on_device) {
        p.replay = c->replay;
      } else {
        const size_t n = (size_t)c->n_paths * steps * ncomp_of(dyn);
        rc = ensure(ctx, ctx->replay_src, ctx->replay_src_cap, n);
        if (rc) return rc;
        HH_HIP(ctx, hipMemcpyAsync(ctx->replay_src, c->replay, n * sizeof(double), hipMemcpyHostToDevice,
                                   ctx->stream));
        if ((rc = note_host_copy(ctx))) return rc;
        p.replay = ctx->replay_src;
      }
      p.replay_path_major = true;
    } else if (c->replay_layout == HH_REPLAY_PATH_MAJOR) {
      rc = ensure(ctx, ctx->replay, ctx->replay_cap, tile_elems);
      if (rc) return rc;
      rc = hh_replay_pack(ctx, dyn, c->n_paths, steps, c->replay, c->replay_on_device, ctx->replay);
      if (rc) return rc;
      p.replay = ctx->replay;
    } else if (c->replay_layout == HH_REPLAY_TILE_MAJOR) {
      if (c->replay_on_device) {
        p.replay = c->replay;
      } else {
        // exact law: one standard normal per trajectory, n_paths doubles (no tile padding needed:
        // the kernel guards the tail); Euler: the padded tile-major buffer of hh_replay_elems()
        const size_t host_elems =
            (c->strategy == HH_EULER_MARUYAMA) ? tile_elems : (size_t)c->n_paths;
        rc = ensure(ctx, ctx->replay, ctx->replay_cap, tile_elems);
        if (rc) return rc;
        HH_HIP(ctx, hipMemcpyAsync(ctx->replay, c->replay, host_elems * sizeof(double),
                                   hipMemcpyHostToDevice, ctx->stream));
        if ((rc = note_host_copy(ctx))) return rc;
        p.replay = ctx->replay;
      }
    } else {
      return fail(ctx, HH_ERR_INVALID, "unknown replay_layout %d", c->replay_layout);
    }
  }

  return HH_OK;
}

// accum_dev != NULL (and the ctx's HH_OPT_FUSE_REDUCE): the simulation kernel reduces its own records into
// accum_dev — *reduced says whether it did.  The Broadie–Kaya chain always adds its own records (its tail kernel):
// into accum_dev, or into ctx->accum when the caller has no use for the sums.  The timing slot opened here is
// closed by end_timing() once the caller has enqueued its last kernel.
static int run_simulation(hh_ctx* ctx, const hh_model* m, const hh_config* c, double* terminal,
                          bool need_terminal_dev, double** terminal_dev_out, double* accum_dev = nullptr,
                          bool* reduced = nullptr) {
  int rc = HH_OK;
  const uint32_t n_tiles = hh::tiles_for(c->n_paths);
  const bool bk = c->strategy == HH_BROADIE_KAYA;
  rc = ensure(ctx, ctx->records, ctx->records_cap,
              (size_t)(bk ? hh::bk_record_count(c->n_paths) : n_tiles) * hh::kRecStride);
  if (rc) return rc;

  hh::DevicePtrs p{};
  p.records = ctx->records;
  const bool fuse = accum_dev && !bk && fuse_for(ctx, c);
  if (fuse) {  // records in the buffer that holds the poison pattern between launches (hh_sim.h)
    if ((rc = ensure_poisoned(ctx, (size_t)n_tiles * hh::kRecStride))) return rc;
    p.records = ctx->frecords;
    p.accum = accum_dev;
    fused_controls(ctx, p);
  }
  if (reduced) *reduced = fuse || bk;
  if (bk) {
    rc = ensure_bk_scratch(ctx, hh::bk_scratch_bytes(c->n_paths, ctx->bk_term_cache));
    if (rc) return rc;
    p.accum = accum_dev ? accum_dev : ctx->accum;
    p.bk_scratch = ctx->bk_scratch;
    p.bk_table_key = &ctx->bk_table_key;
    p.bk_term_cache = ctx->bk_term_cache;
    ctx->bk_last_n = c->n_paths;
    ctx->bk_last_cache = ctx->bk_term_cache;
  }

  if ((rc = stage_noise(ctx, c, p, false))) return rc;

  const size_t n_term = (size_t)c->n_paths * (c->antithetic ? 2 : 1);
  if (terminal && c->terminal_on_device) {
    p.terminal = terminal;
  } else if (terminal || need_terminal_dev) {
    rc = ensure(ctx, ctx->terminal, ctx->terminal_cap, n_term);
    if (rc) return rc;
    p.terminal = ctx->terminal;
  }
  const int n_active = c->n_partials ? hh::count_active_partials(*m, *c) : 0;
  if (need_terminal_dev && n_active) {
    rc = ensure(ctx, ctx->terminal_d, ctx->terminal_d_cap,
                n_term * (size_t)hh::pad_partials((uint32_t)n_active));
    if (rc) return rc;
    p.terminal_d = ctx->terminal_d;
  }
  if (terminal_dev_out) *terminal_dev_out = p.terminal;

  if (ctx->timing) HH_HIP(ctx, hipEventRecord(ctx->tev[ctx->t_count % hh_ctx::kTimingSlots][0], ctx->stream));
  if (c->strategy == HH_BROADIE_KAYA)
    HH_HIP(ctx, hh::launch_bk(*m, *c, p, ctx->stream));
  else
    HH_HIP(ctx, hh::launch_simulation(*m, *c, p, ctx->stream));
  return HH_OK;
}

// the call's last kernel has been enqueued: close the timing slot run_simulation opened
static int end_timing(hh_ctx* ctx) {
  if (ctx->timing) {
    HH_HIP(ctx, hipEventRecord(ctx->tev[ctx->t_count % hh_ctx::kTimingSlots][1], ctx->stream));
    ++ctx->t_count;
  }
  return HH_OK;
}

static int copy_back_terminal(hh_ctx* ctx, const hh_config* c, double* terminal) {
  if (terminal && !c->terminal_on_device) {
    const size_t n_term = (size_t)c->n_paths * (c->antithetic ? 2 : 1);
    HH_HIP(ctx, hipMemcpyAsync(terminal, ctx->terminal, n_term * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
    HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return HH_OK;
}

int hh_mc_accumulate(hh_ctx* ctx, const hh_model* m, const hh_config* c, double* accum_dev,
                     double* terminal) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  int rc = validate(ctx, m, c);
  if (rc) return rc;
  if (!accum_dev) return fail(ctx, HH_ERR_INVALID, "accum_dev is NULL");
  HH_HIP(ctx, hipSetDevice(ctx->device));
  bool reduced = false;
  rc = run_simulation(ctx, m, c, terminal, false, nullptr, accum_dev, &reduced);
  if (rc) return rc;
  if (!reduced)
    HH_HIP(ctx, hh::launch_reduce_records(ctx->records, hh::sim_records(*c), (double)c->n_paths, accum_dev, ctx->stream, 1, m, c));
  if ((rc = end_timing(ctx))) return rc;
  if ((rc = release_host_operands(ctx))) return rc;
  return copy_back_terminal(ctx, c, terminal);
}

// sizes of the passes n_models are run in: at most kMaxModelsPerPass models per launch, never one alone
static int next_pass(int remaining) {
  if (remaining <= hh::kMaxModelsPerPass) return remaining;
  return remaining == hh::kMaxModelsPerPass + 1 ? hh::kMaxModelsPerPass - 1 : hh::kMaxModelsPerPass;
}

int hh_mc_accumulate_multi(hh_ctx* ctx, const hh_model* models, uint32_t n_models, const hh_config* c,
                           double* accum_dev, double* const* terminals) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!models || n_models == 0 || n_models > HH_MAX_MODELS)
    return fail(ctx, HH_ERR_INVALID, "hh_mc_accumulate_multi: 1 .. %d models", HH_MAX_MODELS);
  if (!accum_dev) return fail(ctx, HH_ERR_INVALID, "accum_dev is NULL");
  int rc;
  for (uint32_t k = 0; k < n_models; ++k)
    if ((rc = validate(ctx, &models[k], c))) return rc;
  if (c->n_partials)
    return fail(ctx, HH_ERR_UNSUPPORTED, "several models in one pass carry no dual partials (n_partials must be 0)");
  if (n_models == 1) return hh_mc_accumulate(ctx, models, c, accum_dev, terminals ? terminals[0] : nullptr);
  HH_HIP(ctx, hipSetDevice(ctx->device));  // before anything is allocated: the caller's thread may be on another device
  if (c->strategy == HH_BROADIE_KAYA) {
    // One chain per set of models that the variance process cannot tell apart (hh::bk_same_chain: a bumped spot, rate,
    // ρ, strike — the finite-difference delta / gamma / rho of greeks_problem.jl:279-329, 360-422): the first of a set
    // runs the chain, the others are finished from the ∫V it left (bk_refinish_kernel, ~10 µs each), with the records —
    // hence the sums — of a chain of their own.  A bumped κ, θ, σ, V0 or T is another chain.
    const uint32_t count = hh::bk_record_count(c->n_paths);
    if ((rc = ensure(ctx, ctx->records, ctx->records_cap, 2 * (size_t)count * hh::kRecStride))) return rc;  // BEFORE a chain runs
    std::vector<int> leader(n_models, -1);
    for (uint32_t k = 0; k < n_models; ++k)
      for (uint32_t j = 0; j < k; ++j)
        if (leader[j] < 0 && hh::bk_same_chain(models[j], models[k])) {
          leader[k] = (int)j;
          break;
        }
    for (uint32_t k = 0; k < n_models; ++k) {
      if (leader[k] >= 0) continue;
      if ((rc = hh_mc_accumulate(ctx, &models[k], c, accum_dev + (size_t)k * HH_ACC_LEN, terminals ? terminals[k] : nullptr)))
        return rc;
      for (uint32_t f = k + 1; f < n_models; ++f) {
        if (leader[f] != (int)k) continue;
        double* terminal = terminals ? terminals[f] : nullptr;
        hh::DevicePtrs p{};
        p.records = ctx->records + (size_t)count * hh::kRecStride;
        p.bk_scratch = ctx->bk_scratch;
        p.bk_term_cache = ctx->bk_term_cache;
        // (the draws, the ballots and ∫V are where the chain left them: no seeds, no noise here)
        if (terminal && c->terminal_on_device) {
          p.terminal = terminal;
        } else if (terminal) {
          if ((rc = ensure(ctx, ctx->terminal, ctx->terminal_cap, (size_t)c->n_paths))) return rc;
          p.terminal = ctx->terminal;
        }
        if (ctx->timing) HH_HIP(ctx, hipEventRecord(ctx->tev[ctx->t_count % hh_ctx::kTimingSlots][0], ctx->stream));
        HH_HIP(ctx, hh::launch_bk_refinish(models[f], *c, p, ctx->records, ctx->stream));
        HH_HIP(ctx, hh::launch_reduce_records(p.records, count, (double)c->n_paths, accum_dev + (size_t)f * HH_ACC_LEN,
                                              ctx->stream, 1, &models[f], c, false,
                                              hh::bk_live_records(ctx->bk_scratch, c->n_paths)));
        if ((rc = end_timing(ctx))) return rc;
        if ((rc = copy_back_terminal(ctx, c, terminal))) return rc;
      }
    }
    return HH_OK;
  }
  const uint32_t n_tiles = hh::tiles_for(c->n_paths);
  const size_t rec_elems = (size_t)n_tiles * hh::kRecStride;
  const bool fuse = fuse_for(ctx, c);
  if (fuse) rc = ensure_poisoned(ctx, (size_t)n_models * rec_elems);
  else rc = ensure(ctx, ctx->records, ctx->records_cap, (size_t)n_models * rec_elems);
  if (rc) return rc;
  hh::DevicePtrs base{};
  if (fuse) fused_controls(ctx, base);
  if ((rc = stage_noise(ctx, c, base, /*tile_major_only=*/true))) return rc;
  // terminal samples: a model's own device buffer, or a slice of the ctx's staging buffer for a host buffer
  const size_t n_term = (size_t)c->n_paths * (c->antithetic ? 2 : 1);
  bool any_host_terminal = false;
  if (terminals && !c->terminal_on_device) {
    for (uint32_t k = 0; k < n_models; ++k) any_host_terminal = any_host_terminal || terminals[k] != nullptr;
    if (any_host_terminal && (rc = ensure(ctx, ctx->terminal, ctx->terminal_cap, (size_t)n_models * n_term))) return rc;
  }
  std::vector<hh::DevicePtrs> p(n_models, base);
  for (uint32_t k = 0; k < n_models; ++k) {
    p[k].records = (fuse ? ctx->frecords : ctx->records) + (size_t)k * rec_elems;
    p[k].accum = fuse ? accum_dev + (size_t)k * HH_ACC_LEN : nullptr;
    double* t = terminals ? terminals[k] : nullptr;
    p[k].terminal = !t ? nullptr : c->terminal_on_device ? t : ctx->terminal + (size_t)k * n_term;
  }
  if (ctx->timing) HH_HIP(ctx, hipEventRecord(ctx->tev[ctx->t_count % hh_ctx::kTimingSlots][0], ctx->stream));
  for (uint32_t k0 = 0; k0 < n_models;) {
    const int take = next_pass((int)(n_models - k0));
    HH_HIP(ctx, hh::launch_simulation_multi(models + k0, take, *c, p.data() + k0, ctx->stream));
    if (!fuse)
      for (int k = 0; k < take; ++k)
        HH_HIP(ctx, hh::launch_reduce_records(p[k0 + k].records, hh::sim_records(*c), (double)c->n_paths,
                                              accum_dev + (size_t)(k0 + k) * HH_ACC_LEN, ctx->stream, 1, &models[k0 + k], c));
    k0 += (uint32_t)take;
  }
  if ((rc = end_timing(ctx))) return rc;
  if ((rc = release_host_operands(ctx))) return rc;
  if (any_host_terminal) {
    for (uint32_t k = 0; k < n_models; ++k)
      if (terminals[k])
        HH_HIP(ctx, hipMemcpyAsync(terminals[k], p[k].terminal, n_term * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return HH_OK;
}

int hh_mc_solve_multi(hh_ctx* ctx, const hh_model* models, uint32_t n_models, const hh_config* c, hh_result* out,
                      double* const* terminals) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!out || !models || n_models == 0 || n_models > HH_MAX_MODELS)
    return fail(ctx, HH_ERR_INVALID, "hh_mc_solve_multi: 1 .. %d models and their results", HH_MAX_MODELS);
  const auto t0 = std::chrono::steady_clock::now();
  HH_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n_acc = (size_t)n_models * HH_ACC_LEN;
  int rc = ensure(ctx, ctx->basket_accum, ctx->basket_accum_cap, n_acc);
  if (rc) return rc;
  HH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if ((rc = hh_mc_accumulate_multi(ctx, models, n_models, c, ctx->basket_accum, terminals))) return rc;
  HH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  double host[HH_MAX_MODELS * HH_ACC_LEN];
  HH_HIP(ctx, hipMemcpyAsync(host, ctx->basket_accum, n_acc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (uint32_t k = 0; k < n_models; ++k) {
    std::memset(&out[k], 0, sizeof(hh_result));
    if ((rc = hh_mc_finalize(&models[k], c, host + (size_t)k * HH_ACC_LEN, &out[k]))) return finalize_failed(ctx, rc);
    out[k].kernel_ms = ms;
    out[k].total_ms = total;
  }
  return HH_OK;
}

int hh_mc_accumulate_basket(hh_ctx* ctx, const hh_model* m, const hh_config* c,
                            const double* strikes, const double* cps, uint32_t n_payoffs,
                            double* accum_dev, double* terminal) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  int rc = validate(ctx, m, c);
  if (rc) return rc;
  if (!accum_dev || !strikes || !cps || n_payoffs == 0 || n_payoffs > 65535)
    return fail(ctx, HH_ERR_INVALID, "hh_mc_accumulate_basket: bad arguments");
  for (uint32_t k = 0; k < n_payoffs; ++k)
    if (cps[k] != 1.0 && cps[k] != -1.0) return fail(ctx, HH_ERR_INVALID, "cp must be +1 or -1");
  if (m->dstrike)
    return fail(ctx, HH_ERR_UNSUPPORTED, "strike partials are not carried through a basket");
  HH_HIP(ctx, hipSetDevice(ctx->device));
  double* term_dev = nullptr;
  rc = run_simulation(ctx, m, c, terminal, true, &term_dev);
  if (rc) return rc;

  hh::BasketArgs b{};
  b.n_paths = c->n_paths;
  b.n_chunks = hh::basket_chunks(c->n_paths);
  b.antithetic = c->antithetic;
  rc = ensure(ctx, ctx->payoffs, ctx->payoffs_cap, (size_t)2 * n_payoffs);
  if (rc) return rc;
  HH_HIP(ctx, hipMemcpyAsync(ctx->payoffs, strikes, n_payoffs * sizeof(double),
                             hipMemcpyHostToDevice, ctx->stream));
  HH_HIP(ctx, hipMemcpyAsync(ctx->payoffs + n_payoffs, cps, n_payoffs * sizeof(double),
                             hipMemcpyHostToDevice, ctx->stream));
  if ((rc = note_host_copy(ctx))) return rc;
  rc = ensure(ctx, ctx->basket_records, ctx->basket_records_cap,
              (size_t)n_payoffs * b.n_chunks * hh::kRecStride);
  if (rc) return rc;
  b.terminal = term_dev;
  const int n_active = c->n_partials ? hh::count_active_partials(*m, *c) : 0;
  b.terminal_d = n_active ? ctx->terminal_d : nullptr;
  b.strikes = ctx->payoffs;
  b.cps = ctx->payoffs + n_payoffs;
  b.records = ctx->basket_records;
  HH_HIP(ctx, hh::launch_basket_payoffs(b, n_payoffs, (uint32_t)n_active, ctx->stream));
  HH_HIP(ctx, hh::launch_reduce_records(ctx->basket_records, b.n_chunks, (double)c->n_paths,
                                        accum_dev, ctx->stream, n_payoffs, m, c, true));
  if (c->strategy == HH_BROADIE_KAYA)  // the simulation's fall-back / series counters (the chain left its sums in
    HH_HIP(ctx, hh::launch_copy_bk_counters(ctx->accum, accum_dev, n_payoffs, ctx->stream));  // ctx->accum), for every payoff
  if ((rc = end_timing(ctx))) return rc;
  if ((rc = release_host_operands(ctx))) return rc;
  return copy_back_terminal(ctx, c, terminal);
}

int hh_mc_solve_basket(hh_ctx* ctx, const hh_model* m, const hh_config* c, const double* strikes,
                       const double* cps, uint32_t n_payoffs, hh_result* out, double* terminal) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!out) return fail(ctx, HH_ERR_INVALID, "result is NULL");
  const auto t0 = std::chrono::steady_clock::now();
  HH_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n_acc = (size_t)n_payoffs * HH_ACC_LEN;
  int rc = ensure(ctx, ctx->basket_accum, ctx->basket_accum_cap, n_acc);
  if (rc) return rc;
  HH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  rc = hh_mc_accumulate_basket(ctx, m, c, strikes, cps, n_payoffs, ctx->basket_accum, terminal);
  if (rc) return rc;
  HH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  std::vector<double> host(n_acc);
  HH_HIP(ctx, hipMemcpyAsync(host.data(), ctx->basket_accum, n_acc * sizeof(double),
                             hipMemcpyDeviceToHost, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  const double total =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (uint32_t k = 0; k < n_payoffs; ++k) {
    std::memset(&out[k], 0, sizeof(hh_result));
    rc = hh_mc_finalize(m, c, host.data() + (size_t)k * HH_ACC_LEN, &out[k]);
    if (rc) return finalize_failed(ctx, rc);
    out[k].kernel_ms = ms;
    out[k].total_ms = total;
  }
  return HH_OK;
}

int hh_mc_finalize(const hh_model* m, const hh_config* c, const double* acc, hh_result* out) {
  if (!m || !c || !acc || !out) return HH_ERR_INVALID;
  const double n = acc[HH_ACC_NPATHS];
  if (!(n >= 1.0)) return HH_ERR_INVALID;
  const double mean = acc[HH_ACC_SUM] / n;
  out->sum_payoff = acc[HH_ACC_SUM];
  out->sumsq_payoff = acc[HH_ACC_SUMSQ];
  out->price = m->discount * mean;  // montecarlo.jl:489-490
  double var = 0.0;
  if (n > 1.0) var = (acc[HH_ACC_SUMSQ] - n * mean * mean) / (n - 1.0);
  if (!(var > 0.0)) var = 0.0;
  out->std_error = m->discount * std::sqrt(var / n);
  for (uint32_t k = 0; k < HH_MAX_PARTIALS; ++k) {
    double d = 0.0;
    if (k < c->n_partials) {
      const double dD = m->ddiscount ? m->ddiscount[k] : 0.0;
      d = dD * mean + m->discount * (acc[HH_ACC_DSUM + k] / n);
    }
    out->dprice[k] = d;
  }
  out->n_paths_done = (uint64_t)n;
  out->bk_newton_fail = (uint64_t)acc[HH_ACC_BK_NEWTON_FAIL];
  out->bk_bisect_fallback = (uint64_t)acc[HH_ACC_BK_BISECT];
  out->bk_maxguess_fallback = (uint64_t)acc[HH_ACC_BK_MAXGUESS];
  out->bk_cf_terms = (uint64_t)acc[HH_ACC_BK_CF_TERMS];
  return HH_OK;
}

int hh_mc_solve(hh_ctx* ctx, const hh_model* m, const hh_config* c, hh_result* out,
                double* terminal) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!out) return fail(ctx, HH_ERR_INVALID, "result is NULL");
  const auto t0 = std::chrono::steady_clock::now();
  std::memset(out, 0, sizeof(*out));
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int rc = hh_mc_accumulate(ctx, m, c, ctx->accum, terminal);
  if (rc) return rc;
  HH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HH_HIP(ctx, hipMemcpyAsync(ctx->accum_host, ctx->accum, HH_ACC_LEN * sizeof(double),
                             hipMemcpyDeviceToHost, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  rc = hh_mc_finalize(m, c, ctx->accum_host, out);
  if (rc) return finalize_failed(ctx, rc);
  float ms = 0.f;
  HH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  out->kernel_ms = ms;
  out->total_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return HH_OK;
}

int hh_ctx_check_last(hh_ctx* ctx) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  return recover_finish(ctx);
}

int hh_bk_decisions(hh_ctx* ctx, uint64_t n_paths, uint32_t* decisions, uint32_t* series_len) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!ctx->bk_scratch || ctx->bk_last_n == 0 || n_paths != ctx->bk_last_n)
    return fail(ctx, HH_ERR_INVALID, "hh_bk_decisions: the last Broadie-Kaya solve of this context had %llu trajectories",
                (unsigned long long)ctx->bk_last_n);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t *dec = nullptr, *len = nullptr;
  hh::bk_diag_ptrs(ctx->bk_scratch, n_paths, ctx->bk_last_cache, &dec, &len);
  if (decisions)
    HH_HIP(ctx, hipMemcpyAsync(decisions, dec, n_paths * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (series_len)
    HH_HIP(ctx, hipMemcpyAsync(series_len, len, n_paths * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return HH_OK;
}

int hh_carr_madan(hh_ctx* ctx, const hh_model* m, int32_t dynamics, int32_t compat_sqrt_alpha,
                  double alpha, double bound, double* price_out) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!m || !price_out) return fail(ctx, HH_ERR_INVALID, "hh_carr_madan: NULL argument");
  if (dynamics != HH_LOGNORMAL && dynamics != HH_HESTON)
    return fail(ctx, HH_ERR_INVALID, "unknown dynamics %d", dynamics);
  if (!(m->S0 > 0.0) || !(m->strike > 0.0) || !(m->T > 0.0) || !(alpha > 0.0) || !(bound > 0.0) ||
      (m->cp != 1.0 && m->cp != -1.0) || (dynamics == HH_HESTON && m->sigma == 0.0))
    return fail(ctx, HH_ERR_INVALID, "hh_carr_madan: bad scalars");
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hh::launch_carr_madan(*m, dynamics, compat_sqrt_alpha, alpha, bound, ctx->accum,
                                    ctx->stream));
  HH_HIP(ctx, hipMemcpyAsync(ctx->accum_host, ctx->accum, sizeof(double), hipMemcpyDeviceToHost,
                             ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const double call = ctx->accum_host[0];
  // parity_transform (payoffs.jl:172-193): put = call − S + K·D
  *price_out = m->cp > 0.0 ? call : call - m->S0 + m->strike * m->discount;
  return HH_OK;
}

int hh_carr_madan_basket(hh_ctx* ctx, const hh_model* m, int32_t dynamics, int32_t compat_sqrt_alpha,
                         double alpha, double bound, const double* strikes, const double* cps,
                         const double* Ts, const double* r_drifts, const double* discounts,
                         uint32_t n_payoffs, double* prices_out) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!m || !strikes || !cps || !Ts || !r_drifts || !discounts || !prices_out)
    return fail(ctx, HH_ERR_INVALID, "hh_carr_madan_basket: NULL argument");
  if (dynamics != HH_LOGNORMAL && dynamics != HH_HESTON)
    return fail(ctx, HH_ERR_INVALID, "unknown dynamics %d", dynamics);
  if (n_payoffs == 0 || n_payoffs > (1u << 20))
    return fail(ctx, HH_ERR_INVALID, "hh_carr_madan_basket: 1 .. 2^20 payoffs per call");
  if (!(m->S0 > 0.0) || !(alpha > 0.0) || !(bound > 0.0) || (dynamics == HH_HESTON && m->sigma == 0.0))
    return fail(ctx, HH_ERR_INVALID, "hh_carr_madan_basket: bad scalars");
  const size_t n = n_payoffs;
  std::vector<double> host(5 * n);  // log K | T | r_drift | discount | (out)
  for (size_t k = 0; k < n; ++k) {
    if (!(strikes[k] > 0.0) || !(Ts[k] > 0.0) || (cps[k] != 1.0 && cps[k] != -1.0) ||
        !std::isfinite(r_drifts[k]) || !(discounts[k] > 0.0))
      return fail(ctx, HH_ERR_INVALID, "hh_carr_madan_basket: payoff %zu: strike, T, discount > 0, cp = +-1", k);
    host[k] = std::log(strikes[k]);
    host[n + k] = Ts[k];
    host[2 * n + k] = r_drifts[k];
    host[3 * n + k] = discounts[k];
  }
  HH_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure(ctx, ctx->payoffs, ctx->payoffs_cap, 5 * n);
  if (rc) return rc;
  HH_HIP(ctx, hipMemcpyAsync(ctx->payoffs, host.data(), 4 * n * sizeof(double), hipMemcpyHostToDevice,
                             ctx->stream));
  HH_HIP(ctx, hh::launch_carr_madan_basket(*m, dynamics, compat_sqrt_alpha, alpha, bound, ctx->payoffs,
                                           n_payoffs, ctx->payoffs + 4 * n, ctx->stream));
  HH_HIP(ctx, hipMemcpyAsync(host.data() + 4 * n, ctx->payoffs + 4 * n, n * sizeof(double),
                             hipMemcpyDeviceToHost, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t k = 0; k < n; ++k) {  // parity_transform (payoffs.jl:172-193): put = call − S + K·D
    const double call = host[4 * n + k];
    prices_out[k] = cps[k] > 0.0 ? call : call - m->S0 + strikes[k] * discounts[k];
  }
  return HH_OK;
}

int hh_carr_madan_basket_grad(hh_ctx* ctx, const hh_model* m, int32_t dynamics,
                              int32_t compat_sqrt_alpha, double alpha, double bound,
                              const double* strikes, const double* cps, const double* Ts,
                              const double* r_drifts, const double* discounts, uint32_t n_payoffs,
                              double* prices_out, double* grad_out) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!m || !strikes || !cps || !Ts || !r_drifts || !discounts || !prices_out || !grad_out)
    return fail(ctx, HH_ERR_INVALID, "hh_carr_madan_basket_grad: NULL argument");
  if (dynamics != HH_LOGNORMAL && dynamics != HH_HESTON)
    return fail(ctx, HH_ERR_INVALID, "unknown dynamics %d", dynamics);
  if (n_payoffs == 0 || n_payoffs > (1u << 20))
    return fail(ctx, HH_ERR_INVALID, "hh_carr_madan_basket_grad: 1 .. 2^20 payoffs per call");
  if (!(m->S0 > 0.0) || !(alpha > 0.0) || !(bound > 0.0) ||
      (dynamics == HH_HESTON && (m->sigma == 0.0 || m->theta == 0.0)))
    return fail(ctx, HH_ERR_INVALID, "hh_carr_madan_basket_grad: bad scalars (Heston: sigma, theta != 0)");
  const size_t n = n_payoffs;
  std::vector<double> host(4 * n + HH_CM_GRAD_LEN * n);  // log K | T | r_drift | discount | out [n][8]
  for (size_t k = 0; k < n; ++k) {
    if (!(strikes[k] > 0.0) || !(Ts[k] > 0.0) || (cps[k] != 1.0 && cps[k] != -1.0) ||
        !std::isfinite(r_drifts[k]) || !(discounts[k] > 0.0))
      return fail(ctx, HH_ERR_INVALID, "hh_carr_madan_basket_grad: payoff %zu: strike, T, discount > 0, cp = +-1", k);
    host[k] = std::log(strikes[k]);
    host[n + k] = Ts[k];
    host[2 * n + k] = r_drifts[k];
    host[3 * n + k] = discounts[k];
  }
  HH_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure(ctx, ctx->payoffs, ctx->payoffs_cap, 4 * n + HH_CM_GRAD_LEN * n);
  if (rc) return rc;
  double* out_dev = ctx->payoffs + 4 * n;
  HH_HIP(ctx, hipMemcpyAsync(ctx->payoffs, host.data(), 4 * n * sizeof(double), hipMemcpyHostToDevice,
                             ctx->stream));
  HH_HIP(ctx, hh::launch_carr_madan_grad(*m, dynamics, compat_sqrt_alpha, alpha, bound, ctx->payoffs,
                                         n_payoffs, out_dev, ctx->stream));
  HH_HIP(ctx, hipMemcpyAsync(host.data() + 4 * n, out_dev, HH_CM_GRAD_LEN * n * sizeof(double),
                             hipMemcpyDeviceToHost, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t k = 0; k < n; ++k) {
    const double* o = host.data() + 4 * n + HH_CM_GRAD_LEN * k;  // call, then d/d(S0 V0 κ θ σ ρ r_drift)
    double* g = grad_out + HH_CM_GRAD_LEN * k;
    for (int i = 0; i < 7; ++i) g[i] = o[1 + i];
    g[HH_CM_GRAD_DISCOUNT] = o[0] / discounts[k];  // the price is linear in the discount factor
    double price = o[0];
    if (cps[k] < 0.0) {  // parity_transform (payoffs.jl:172-193): put = call − S + K·D
      price = price - m->S0 + strikes[k] * discounts[k];
      g[HH_CM_GRAD_S0] -= 1.0;
      g[HH_CM_GRAD_DISCOUNT] += strikes[k];
    }
    prices_out[k] = price;
  }
  return HH_OK;
}

size_t hh_lsm_grid_elems(uint64_t n_paths, uint32_t n_steps, int32_t antithetic) {
  return (size_t)(n_steps + 1) * n_paths * (antithetic ? 2 : 1);
}

// Backward induction on a spot grid already in device memory (rows = dates, ntot trajectories):
// launches, reduction and the copies back.  The caller recorded ctx->ev0 before producing the grid.
static int lsm_on_grid(hh_ctx* ctx, const double* grid_dev, uint64_t ntot, uint32_t n_steps,
                       const hh_model* m, int32_t degree, double step_discount, hh_lsm_result* out,
                       int32_t* stop_time, double* stop_value,
                       std::chrono::steady_clock::time_point t0) {
  const uint32_t ch = hh::lsm_chunks(ntot);
  int rc;
  if ((rc = ensure(ctx, ctx->lsm_val, ctx->lsm_val_cap, (size_t)ntot))) return rc;
  if ((rc = ensure(ctx, ctx->lsm_tau, ctx->lsm_tau_cap, (size_t)ntot))) return rc;
  const size_t nscr = hh::lsm_scratch_doubles(ntot, n_steps, degree);
  if ((rc = ensure(ctx, ctx->lsm_scratch, ctx->lsm_scratch_cap, nscr))) return rc;
  if ((rc = ensure(ctx, ctx->records, ctx->records_cap, (size_t)ch * hh::kRecStride))) return rc;
  // enqueue the induction, the final reduction and every copy back, then synchronise ONCE; the
  // one-launch form leaves a word behind when its workgroups could not all be resident together
  // (another kernel held CUs): nothing was written then, and the launch-per-date form runs instead
  int form_used = hh::kLsmFormPerDate;
  // the small results land in PINNED memory behind the accumulator (a copy to the stack is staged
  // through the runtime's own buffer and costs ~10 µs apiece)
  double* counters = ctx->accum_host + HH_ACC_LEN;
  unsigned int& gave_up = *reinterpret_cast<unsigned int*>(ctx->accum_host + HH_ACC_LEN + 2);
  counters[0] = counters[1] = 0.0;
  int32_t fallbacks = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int form = attempt == 0 ? ctx->lsm_form : hh::kLsmFormPerDate;
    HH_HIP(ctx, hh::launch_lsm(grid_dev, ntot, n_steps, m->strike, m->cp, step_discount, degree,
                               ctx->lsm_tau, ctx->lsm_val, ctx->lsm_scratch, ctx->records,
                               ctx->stream, form, &form_used,
                               ctx->lsm_spin_ticks < 0 ? 100000000ull : (unsigned long long)ctx->lsm_spin_ticks));
    HH_HIP(ctx, hh::launch_reduce_records(ctx->records, ch, (double)ntot, ctx->accum, ctx->stream));
    HH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    gave_up = 0;
    if (form_used == hh::kLsmFormPersistent)
      HH_HIP(ctx, hipMemcpyAsync(&gave_up, hh::lsm_persistent_status(ctx->lsm_scratch),
                                 sizeof(gave_up), hipMemcpyDeviceToHost, ctx->stream));
    HH_HIP(ctx, hipMemcpyAsync(ctx->accum_host, ctx->accum, HH_ACC_LEN * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
    HH_HIP(ctx, hipMemcpyAsync(counters, ctx->lsm_scratch + nscr - 2 - hh::kLsmStampSlotsApi,
                               2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (stop_time)
      HH_HIP(ctx, hipMemcpyAsync(stop_time, ctx->lsm_tau, ntot * sizeof(int32_t),
                                 hipMemcpyDeviceToHost, ctx->stream));
    if (stop_value)
      HH_HIP(ctx, hipMemcpyAsync(stop_value, ctx->lsm_val, ntot * sizeof(double),
                                 hipMemcpyDeviceToHost, ctx->stream));
    HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!gave_up) break;
    ++ctx->lsm_persistent_fallbacks;
    ++fallbacks;
  }
  const double n = (double)ntot, mean = ctx->accum_host[HH_ACC_SUM] / n;
  double var = n > 1.0 ? (ctx->accum_host[HH_ACC_SUMSQ] - n * mean * mean) / (n - 1.0) : 0.0;
  if (!(var > 0.0)) var = 0.0;
  std::memset(out, 0, sizeof(*out));
  out->price = mean;  // price = mean(discount^t * val) (least_squares_montecarlo.jl:133-134)
  out->std_error = std::sqrt(var / n);
  out->n_paths_total = ntot;
  out->rows_regressed = (uint32_t)counters[0];
  out->rows_skipped = (uint32_t)counters[1];
  out->form = form_used;
  out->persistent_fallbacks = fallbacks;
  float ms = 0.f;
  HH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  out->kernel_ms = ms;
  out->total_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return HH_OK;
}

static int lsm_check_scalars(hh_ctx* ctx, const hh_model* m, const hh_config* c, int32_t degree,
                             double step_discount) {
  if (c->n_paths == 0 || c->n_steps == 0 || degree < 1 || degree > 8)
    return fail(ctx, HH_ERR_INVALID, "LSM: n_paths, n_steps >= 1, 1 <= degree <= 8");
  if (c->n_paths > kMaxPaths / 2 || c->n_steps > kMaxGridSteps)
    return fail(ctx, HH_ERR_INVALID, "LSM: at most 2^31 - 128 trajectories (x2 antithetic) and %u steps",
                kMaxGridSteps);
  if (!(m->S0 > 0.0) || !(m->T > 0.0) || (m->cp != 1.0 && m->cp != -1.0) ||
      !(step_discount > 0.0) || !std::isfinite(step_discount))
    return fail(ctx, HH_ERR_INVALID, "LSM: bad model scalars");
  return HH_OK;
}

// seeds of the trajectories in device memory (staged if the caller's are on the host)
static int stage_path_seeds(hh_ctx* ctx, const hh_config* c, const uint64_t** out) {
  if (!c->seeds) return fail(ctx, HH_ERR_INVALID, "GENERATE needs seeds");
  if (c->seeds_len && c->seeds_len < c->n_paths)
    return fail(ctx, HH_ERR_INVALID, "Number of seeds (%llu) must be >= number of trajectories (%llu)",
                (unsigned long long)c->seeds_len, (unsigned long long)c->n_paths);
  *out = c->seeds;
  if (!c->seeds_on_device) {
    int rc = ensure(ctx, ctx->seeds, ctx->seeds_cap, (size_t)c->n_paths);
    if (rc) return rc;
    HH_HIP(ctx, hipMemcpyAsync(ctx->seeds, c->seeds, c->n_paths * sizeof(uint64_t),
                               hipMemcpyHostToDevice, ctx->stream));
    if ((rc = note_host_copy(ctx))) return rc;
    *out = ctx->seeds;
  }
  return HH_OK;
}

// The n_steps Broadie–Kaya transitions of length T/n_steps into ctx->lsm_grid (spot rows) and
// ctx->heston_var (variance rows); per-transition accumulator vectors (BK counters) into
// ctx->basket_accum[n_steps][HH_ACC_LEN].
static int run_heston_grid(hh_ctx* ctx, const hh_model* m, const hh_config* c) {
  if (c->dynamics != HH_HESTON || c->strategy != HH_BROADIE_KAYA)
    return fail(ctx, HH_ERR_UNSUPPORTED, "exact Heston grid needs HestonDynamics + HestonBroadieKaya");
  if (c->noise_mode != HH_NOISE_GENERATE || c->n_partials != 0 || c->antithetic)
    return fail(ctx, HH_ERR_UNSUPPORTED,
                "exact Heston grid: GENERATE noise, no dual partials, no antithetic form");
  if (c->n_paths == 0 || c->n_steps == 0 || c->n_paths > kMaxPaths || c->n_steps > kMaxGridSteps)
    return fail(ctx, HH_ERR_INVALID, "exact Heston grid: 1 <= n_paths <= 2^32 - 256, 1 <= n_steps <= %u",
                kMaxGridSteps);
  if (!(m->S0 > 0.0) || !(m->T > 0.0) || !std::isfinite(m->S0) || !std::isfinite(m->T) ||
      !(std::fabs(m->rho) <= 1.0) || m->sigma == 0.0 || m->kappa == 0.0 || !(m->V0 > 0.0) ||
      !std::isfinite(m->sigma) || !std::isfinite(m->kappa) || !std::isfinite(m->theta) ||
      !std::isfinite(m->V0) || !std::isfinite(m->r_drift) ||
      !bk_law_ok(m, m->T / (double)c->n_steps))
    return fail(ctx, HH_ERR_INVALID,
                "exact Heston grid: S0, T, V0 > 0, |rho| <= 1, 1e-8 <= 4 kappa theta / sigma^2 <= 1e6, all finite (|kappa dt| < 700)");
  const uint64_t n = c->n_paths;
  const size_t grid_elems = (size_t)(c->n_steps + 1) * n;
  // dates per kernel chain: given the variance rows, the CF inversions of different dates are
  // independent, so a batch of dates runs as ONE chain over its (date, trajectory) pairs
  uint32_t per_chain = ctx->grid_form == HH_GRID_FORM_BATCHED
                           ? hh::bk_grid_dates_per_chain(n, c->n_steps, ctx->bk_term_cache)
                           : 1u;
  int rc;
  if ((rc = ensure(ctx, ctx->lsm_grid, ctx->lsm_grid_cap, grid_elems))) return rc;
  if ((rc = ensure(ctx, ctx->heston_var, ctx->heston_var_cap, grid_elems))) return rc;
  // a chain's scratch is 48 bytes per (date, trajectory) pair beside the fixed term cache: on a device
  // that cannot spare it (shared with another allocator) the dates go into shorter chains, down to one
  // chain per date — same bits either way
  uint64_t n_chain = n * per_chain;
  while ((rc = ensure_bk_scratch(ctx, hh::bk_scratch_bytes(n_chain, ctx->bk_term_cache))) == HH_ERR_NOMEM &&
         per_chain > 1) {
    per_chain = (per_chain + 1) / 2;
    n_chain = n * per_chain;
  }
  if (rc) return rc;
  if ((rc = ensure(ctx, ctx->records, ctx->records_cap, (size_t)hh::bk_record_count(n_chain) * hh::kRecStride)))
    return rc;
  if ((rc = ensure(ctx, ctx->basket_accum, ctx->basket_accum_cap, (size_t)c->n_steps * HH_ACC_LEN)))
    return rc;
  hh::DevicePtrs p{};
  p.records = ctx->records;
  p.bk_scratch = ctx->bk_scratch;
  p.bk_table_key = &ctx->bk_table_key;
  p.bk_term_cache = ctx->bk_term_cache;
  if (per_chain > 1 && ctx->grid_order) {  // the batched chains run their pairs in the order of their Bessel arguments
    if ((rc = ensure(ctx, ctx->bk_sort, ctx->bk_sort_cap, hh::bk_grid_sort_bytes(n_chain)))) return rc;
    p.bk_sort = ctx->bk_sort;
  }
  if ((rc = stage_path_seeds(ctx, c, &p.seeds))) return rc;
  HH_HIP(ctx, hh::launch_fill_rows(ctx->lsm_grid, ctx->heston_var, n, m->S0, m->V0, ctx->stream));
  hh_model step_model = *m;
  step_model.T = m->T / (double)c->n_steps;  // dt of solve(NoiseProblem; dt = T / steps)
  step_model.cp = 1.0;
  hh_config step_cfg = *c;
  step_cfg.path_offset = 0;
  if (per_chain > 1) {
    HH_HIP(ctx, hipMemsetAsync(ctx->basket_accum, 0, (size_t)c->n_steps * HH_ACC_LEN * sizeof(double),
                               ctx->stream));
    for (uint32_t k = 0; k < c->n_steps; k += per_chain) {
      const uint32_t nd = std::min(per_chain, c->n_steps - k);
      p.accum = ctx->basket_accum + (size_t)k * HH_ACC_LEN;  // the counters of the batch's pairs, kept in the slot of its first date
      HH_HIP(ctx, hh::launch_bk_grid(step_model, step_cfg, p, ctx->stream, ctx->lsm_grid + (size_t)k * n,
                                     ctx->heston_var + (size_t)k * n, k, nd, /*upload_tables=*/k == 0));
    }
    return HH_OK;
  }
  for (uint32_t k = 0; k < c->n_steps; ++k) {
    const hh::BkTransition tr{ctx->lsm_grid + (size_t)k * n, ctx->heston_var + (size_t)k * n,
                              ctx->lsm_grid + (size_t)(k + 1) * n,
                              ctx->heston_var + (size_t)(k + 1) * n, k};
    p.accum = ctx->basket_accum + (size_t)k * HH_ACC_LEN;
    HH_HIP(ctx, hh::launch_bk(step_model, step_cfg, p, ctx->stream, &tr, /*upload_tables=*/k == 0));
  }
  return HH_OK;
}

// BK counters of all transitions, summed
static int heston_grid_counters(hh_ctx* ctx, uint32_t n_steps, uint64_t* newton_fail,
                                uint64_t* bisect, uint64_t* maxguess, uint64_t* cf_terms) {
  std::vector<double> acc((size_t)n_steps * HH_ACC_LEN);
  HH_HIP(ctx, hipMemcpyAsync(acc.data(), ctx->basket_accum, acc.size() * sizeof(double),
                             hipMemcpyDeviceToHost, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double f = 0, b = 0, g = 0, t = 0;
  for (uint32_t k = 0; k < n_steps; ++k) {
    const double* a = acc.data() + (size_t)k * HH_ACC_LEN;
    f += a[HH_ACC_BK_NEWTON_FAIL]; b += a[HH_ACC_BK_BISECT];
    g += a[HH_ACC_BK_MAXGUESS]; t += a[HH_ACC_BK_CF_TERMS];
  }
  *newton_fail = (uint64_t)f; *bisect = (uint64_t)b; *maxguess = (uint64_t)g; *cf_terms = (uint64_t)t;
  return HH_OK;
}

int hh_heston_exact_grid(hh_ctx* ctx, const hh_model* m, const hh_config* c, double* spot_grid,
                         double* var_grid, int32_t grids_on_device, hh_result* out) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!m || !c) return fail(ctx, HH_ERR_INVALID, "hh_heston_exact_grid: NULL argument");
  const auto t0 = std::chrono::steady_clock::now();
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int rc = run_heston_grid(ctx, m, c);
  if (rc) return rc;
  HH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  const size_t bytes = (size_t)(c->n_steps + 1) * c->n_paths * sizeof(double);
  const hipMemcpyKind kind = grids_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (spot_grid) HH_HIP(ctx, hipMemcpyAsync(spot_grid, ctx->lsm_grid, bytes, kind, ctx->stream));
  if (var_grid) HH_HIP(ctx, hipMemcpyAsync(var_grid, ctx->heston_var, bytes, kind, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (out) {
    std::memset(out, 0, sizeof(*out));
    if ((rc = heston_grid_counters(ctx, c->n_steps, &out->bk_newton_fail, &out->bk_bisect_fallback,
                                   &out->bk_maxguess_fallback, &out->bk_cf_terms)))
      return rc;
    out->n_paths_done = c->n_paths;
    float ms = 0.f;
    HH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    out->kernel_ms = ms;
    out->total_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return HH_OK;
}

int hh_lsm_solve(hh_ctx* ctx, const hh_model* m, const hh_config* c, int32_t degree,
                 double step_discount, hh_lsm_result* out, int32_t* stop_time, double* stop_value,
                 double* spot_grid) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!m || !c || !out) return fail(ctx, HH_ERR_INVALID, "hh_lsm_solve: NULL argument");
  const auto t0 = std::chrono::steady_clock::now();
  // The reference's LSM regresses on the first state component of simulate_paths' solution
  // (least_squares_montecarlo.jl:53,76).  Path sources: the GBM noise process of (LognormalDynamics,
  // BlackScholesExact) (montecarlo.jl:140-159), whose state IS the spot, and the per-date exact
  // Heston transitions of (HestonDynamics, HestonBroadieKaya) (montecarlo.jl:209-231), whose spot
  // rows exp(log S) are used here (the reference hands its regression the log-state).
  const bool gbm = c->dynamics == HH_LOGNORMAL && c->strategy == HH_EXACT_LAW;
  const bool heston = c->dynamics == HH_HESTON && c->strategy == HH_BROADIE_KAYA;
  if (!gbm && !heston)
    return fail(ctx, HH_ERR_UNSUPPORTED,
                "LSM needs LognormalDynamics + BlackScholesExact or HestonDynamics + "
                "HestonBroadieKaya paths");
  if (c->noise_mode != HH_NOISE_GENERATE || c->n_partials != 0)
    return fail(ctx, HH_ERR_UNSUPPORTED, "LSM: GENERATE noise, no dual partials");
  int rc = lsm_check_scalars(ctx, m, c, degree, step_discount);
  if (rc) return rc;
  HH_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t ntot = c->n_paths * (c->antithetic ? 2 : 1);
  const size_t grid_elems = hh_lsm_grid_elems(c->n_paths, c->n_steps, c->antithetic);
  HH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if (gbm) {
    if ((rc = ensure(ctx, ctx->lsm_grid, ctx->lsm_grid_cap, grid_elems))) return rc;
    const uint64_t* seeds_dev = nullptr;
    if ((rc = stage_path_seeds(ctx, c, &seeds_dev))) return rc;
    HH_HIP(ctx, hh::launch_gbm_grid(seeds_dev, c->n_paths, c->n_steps, m->S0, m->r_drift, m->sigma,
                                    m->T, c->antithetic, ctx->lsm_grid, ctx->stream));
  } else {
    if ((rc = run_heston_grid(ctx, m, c))) return rc;
  }
  rc = lsm_on_grid(ctx, ctx->lsm_grid, ntot, c->n_steps, m, degree, step_discount, out, stop_time,
                   stop_value, t0);
  if (rc) return rc;
  if (spot_grid) {
    HH_HIP(ctx, hipMemcpyAsync(spot_grid, ctx->lsm_grid, grid_elems * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
    HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return HH_OK;
}

// ---- LSM on an ensemble sharded over several devices ------------------------------------------------

size_t hh_lsm_shard_xchg_elems(uint32_t n_steps, int32_t degree) {
  return (size_t)(n_steps + 1) * (size_t)(2 * (degree < 1 ? 1 : degree) + 1);
}

static int shard_phase(hh_ctx* ctx, int phase, uint32_t t, const double* in_dev, double* out_dev) {
  const auto& sh = ctx->shard;
  HH_HIP(ctx, hh::launch_lsm_phase(phase, t, ctx->lsm_grid, sh.ntot, sh.n_steps, sh.m.strike, sh.m.cp,
                                   sh.step_discount, sh.degree, ctx->lsm_tau, ctx->lsm_val,
                                   ctx->lsm_scratch, ctx->records, in_dev, out_dev, ctx->stream));
  return HH_OK;
}

int hh_lsm_shard_begin(hh_ctx* ctx, const hh_model* m, const hh_config* c, int32_t degree,
                       double step_discount, double* xchg_dev) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  ctx->shard.active = false;
  if (!m || !c || !xchg_dev) return fail(ctx, HH_ERR_INVALID, "hh_lsm_shard_begin: NULL argument");
  const bool gbm = c->dynamics == HH_LOGNORMAL && c->strategy == HH_EXACT_LAW;
  const bool heston = c->dynamics == HH_HESTON && c->strategy == HH_BROADIE_KAYA;
  if (!gbm && !heston)
    return fail(ctx, HH_ERR_UNSUPPORTED,
                "LSM needs LognormalDynamics + BlackScholesExact or HestonDynamics + "
                "HestonBroadieKaya paths");
  if (c->noise_mode != HH_NOISE_GENERATE || c->n_partials != 0)
    return fail(ctx, HH_ERR_UNSUPPORTED, "LSM: GENERATE noise, no dual partials");
  int rc = lsm_check_scalars(ctx, m, c, degree, step_discount);
  if (rc) return rc;
  HH_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t ntot = c->n_paths * (c->antithetic ? 2 : 1);
  if (gbm) {
    const size_t grid_elems = hh_lsm_grid_elems(c->n_paths, c->n_steps, c->antithetic);
    if ((rc = ensure(ctx, ctx->lsm_grid, ctx->lsm_grid_cap, grid_elems))) return rc;
    const uint64_t* seeds_dev = nullptr;
    if ((rc = stage_path_seeds(ctx, c, &seeds_dev))) return rc;
    HH_HIP(ctx, hh::launch_gbm_grid(seeds_dev, c->n_paths, c->n_steps, m->S0, m->r_drift, m->sigma,
                                    m->T, c->antithetic, ctx->lsm_grid, ctx->stream));
  } else {
    if ((rc = run_heston_grid(ctx, m, c))) return rc;
  }
  if ((rc = ensure(ctx, ctx->lsm_val, ctx->lsm_val_cap, (size_t)ntot))) return rc;
  if ((rc = ensure(ctx, ctx->lsm_tau, ctx->lsm_tau_cap, (size_t)ntot))) return rc;
  if ((rc = ensure(ctx, ctx->lsm_scratch, ctx->lsm_scratch_cap,
                   hh::lsm_scratch_doubles(ntot, c->n_steps, degree))))
    return rc;
  if ((rc = ensure(ctx, ctx->records, ctx->records_cap, (size_t)hh::lsm_chunks(ntot) * hh::kRecStride)))
    return rc;
  ctx->shard.m = *m;
  ctx->shard.ntot = ntot;
  ctx->shard.n_steps = c->n_steps;
  ctx->shard.degree = degree;
  ctx->shard.step_discount = step_discount;
  if ((rc = shard_phase(ctx, hh::kLsmPhaseStats, 0, nullptr, xchg_dev))) return rc;
  ctx->shard.active = true;
  return release_host_operands(ctx);
}

int hh_lsm_shard_phase(hh_ctx* ctx, int32_t phase, uint32_t t, const double* in_dev, double* out_dev) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!ctx->shard.active) return fail(ctx, HH_ERR_INVALID, "no sharded LSM in progress");
  if (phase != HH_LSM_PHASE_POW && phase != HH_LSM_PHASE_INIT && phase != HH_LSM_PHASE_STEP)
    return fail(ctx, HH_ERR_INVALID, "unknown LSM phase %d", phase);
  if (!in_dev) return fail(ctx, HH_ERR_INVALID, "hh_lsm_shard_phase: in_dev is NULL");
  const bool emits = phase != HH_LSM_PHASE_STEP ? (phase == HH_LSM_PHASE_POW || ctx->shard.n_steps >= 2)
                                                : t >= 2;
  if (emits && !out_dev) return fail(ctx, HH_ERR_INVALID, "hh_lsm_shard_phase: out_dev is NULL");
  if (phase == HH_LSM_PHASE_STEP && (t < 1 || t >= ctx->shard.n_steps))
    return fail(ctx, HH_ERR_INVALID, "LSM step index %u outside 1..n_steps-1", t);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  return shard_phase(ctx, phase, t, in_dev, out_dev);
}

int hh_lsm_shard_finish(hh_ctx* ctx, double* accum_dev, int32_t* stop_time, double* stop_value,
                        double* spot_grid, uint32_t* rows_regressed, uint32_t* rows_skipped) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!ctx->shard.active) return fail(ctx, HH_ERR_INVALID, "no sharded LSM in progress");
  if (!accum_dev) return fail(ctx, HH_ERR_INVALID, "accum_dev is NULL");
  HH_HIP(ctx, hipSetDevice(ctx->device));
  const auto& sh = ctx->shard;
  int rc = shard_phase(ctx, hh::kLsmPhaseFinal, 0, nullptr, nullptr);
  if (rc) return rc;
  HH_HIP(ctx, hh::launch_reduce_records(ctx->records, hh::lsm_chunks(sh.ntot), (double)sh.ntot,
                                        accum_dev, ctx->stream));
  const size_t nscr = hh::lsm_scratch_doubles(sh.ntot, sh.n_steps, sh.degree);
  double counters[2] = {0, 0};
  HH_HIP(ctx, hipMemcpyAsync(counters, ctx->lsm_scratch + nscr - 2 - hh::kLsmStampSlotsApi, 2 * sizeof(double),
                             hipMemcpyDeviceToHost, ctx->stream));
  if (stop_time)
    HH_HIP(ctx, hipMemcpyAsync(stop_time, ctx->lsm_tau, sh.ntot * sizeof(int32_t),
                               hipMemcpyDeviceToHost, ctx->stream));
  if (stop_value)
    HH_HIP(ctx, hipMemcpyAsync(stop_value, ctx->lsm_val, sh.ntot * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
  if (spot_grid)
    HH_HIP(ctx, hipMemcpyAsync(spot_grid, ctx->lsm_grid,
                               (size_t)(sh.n_steps + 1) * sh.ntot * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (rows_regressed) *rows_regressed = (uint32_t)counters[0];
  if (rows_skipped) *rows_skipped = (uint32_t)counters[1];
  ctx->shard.active = false;
  return HH_OK;
}

int hh_lsm_debug_read(hh_ctx* ctx, uint64_t n_paths_total, uint32_t n_steps, int32_t degree,
                      double* out8) {
  if (!ctx || !out8) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nscr = hh::lsm_scratch_doubles(n_paths_total, n_steps, degree);
  if (!ctx->lsm_scratch || ctx->lsm_scratch_cap < nscr)
    return fail(ctx, HH_ERR_INVALID, "hh_lsm_debug_read: no LSM solve of that shape has run");
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  HH_HIP(ctx, hipMemcpy(out8, ctx->lsm_scratch + nscr - hh::kLsmStampSlotsApi,
                        hh::kLsmStampSlotsApi * sizeof(double), hipMemcpyDeviceToHost));
  return HH_OK;
}

int hh_lsm_finalize(const double* acc, hh_lsm_result* out) {
  if (!acc || !out) return HH_ERR_INVALID;
  const double n = acc[HH_ACC_NPATHS];
  if (!(n >= 1.0)) return HH_ERR_INVALID;
  const double mean = acc[HH_ACC_SUM] / n;
  double var = n > 1.0 ? (acc[HH_ACC_SUMSQ] - n * mean * mean) / (n - 1.0) : 0.0;
  if (!(var > 0.0)) var = 0.0;
  std::memset(out, 0, sizeof(*out));
  out->price = mean;  // price = mean(discount^t * val) (least_squares_montecarlo.jl:133-134)
  out->std_error = std::sqrt(var / n);
  out->n_paths_total = (uint64_t)n;
  return HH_OK;
}

int hh_lsm_solve_grid(hh_ctx* ctx, const hh_model* m, const double* spot_grid_dev, uint64_t n_paths,
                      uint32_t n_steps, int32_t degree, double step_discount, hh_lsm_result* out,
                      int32_t* stop_time, double* stop_value) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!m || !spot_grid_dev || !out)
    return fail(ctx, HH_ERR_INVALID, "hh_lsm_solve_grid: NULL argument");
  const auto t0 = std::chrono::steady_clock::now();
  hh_config c{};
  c.n_paths = n_paths;
  c.n_steps = n_steps;
  int rc = lsm_check_scalars(ctx, m, &c, degree, step_discount);
  if (rc) return rc;
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return lsm_on_grid(ctx, spot_grid_dev, n_paths, n_steps, m, degree, step_discount, out, stop_time,
                     stop_value, t0);
}

int hh_ctx_enable_timing(hh_ctx* ctx, int32_t on) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  if (on && !ctx->tev[0][0]) {
    for (auto& pr : ctx->tev)
      for (auto& e : pr) HH_HIP(ctx, hipEventCreate(&e));
  }
  ctx->timing = on != 0;
  ctx->t_count = 0;
  return HH_OK;
}

int hh_ctx_read_timings(hh_ctx* ctx, double* ms, int32_t cap, int32_t* n_out) {
  if (!ctx || !ms || !n_out || cap < 0) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int n = ctx->t_count < hh_ctx::kTimingSlots ? ctx->t_count : hh_ctx::kTimingSlots;
  if (n > cap) n = cap;
  for (int i = 0; i < n; ++i) {
    float t = 0.f;
    HH_HIP(ctx, hipEventElapsedTime(&t, ctx->tev[i][0], ctx->tev[i][1]));
    ms[i] = t;
  }
  *n_out = n;
  ctx->t_count = 0;
  return HH_OK;
}

uint64_t hh_seeds_fingerprint(const uint64_t* seeds, uint64_t n) {
  // four independent multiply-xorshift lanes over the whole vector (about a millisecond per 10^6 seeds on one
  // core), folded with the length: every element and its position enter
  uint64_t h[4] = {0x9E3779B97F4A7C15ull ^ n, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
  uint64_t i = 0;
  for (; i + 4 <= n; i += 4)
    for (int l = 0; l < 4; ++l) {
      uint64_t x = (seeds[i + l] + h[l]) * 0xFF51AFD7ED558CCDull;
      h[l] = (x ^ (x >> 29)) + 0x9E3779B97F4A7C15ull;
    }
  for (; i < n; ++i) {
    uint64_t x = (seeds[i] + h[i & 3]) * 0xFF51AFD7ED558CCDull;
    h[i & 3] = (x ^ (x >> 29)) + 0x9E3779B97F4A7C15ull;
  }
  uint64_t f = n;
  for (int l = 0; l < 4; ++l) {
    f = (f ^ h[l]) * 0xC4CEB9FE1A85EC53ull;
    f ^= f >> 32;
  }
  return f ? f : 1;  // 0 is "not given" in hh_seeds_cache
}

int hh_seeds_cache(hh_ctx* ctx, const uint64_t* seeds, uint64_t n, uint64_t fingerprint, const uint64_t** dev_out) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (!seeds || n == 0 || !dev_out) return fail(ctx, HH_ERR_INVALID, "hh_seeds_cache: seeds, n >= 1 and dev_out");
  *dev_out = nullptr;
  if (fingerprint == 0) fingerprint = hh_seeds_fingerprint(seeds, n);
  const uint64_t head = seeds[0], tail = seeds[n - 1];
  hh_ctx::SeedEntry* victim = &ctx->seed_cache[0];
  for (auto& e : ctx->seed_cache) {
    if (e.dev && e.n == n && e.fingerprint == fingerprint && e.head == head && e.tail == tail) {
      e.stamp = ++ctx->seed_clock;
      ++ctx->seed_hits;
      *dev_out = e.dev;
      return HH_OK;
    }
    // the least recently used one goes — never the entry the call before this one returned (its stamp is the clock's):
    // a second thread of this context may be between ITS lookup and the solve that reads the pointer
    const bool newest = e.dev && e.stamp == ctx->seed_clock, victim_newest = victim->dev && victim->stamp == ctx->seed_clock;
    if (!e.dev ? victim->dev != nullptr : (victim->dev && !newest && (victim_newest || e.stamp < victim->stamp))) victim = &e;
  }
  HH_HIP(ctx, hipSetDevice(ctx->device));
  if (victim->dev) {  // hipFree waits for the device: nothing in flight reads the copy any more
    HH_HIP(ctx, hipFree(victim->dev));
    *victim = hh_ctx::SeedEntry{};
    ++ctx->seed_evictions;
  }
  uint64_t* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, n * sizeof(uint64_t));
  if (e != hipSuccess) return fail(ctx, HH_ERR_NOMEM, "hipMalloc(%llu bytes) failed: %s", (unsigned long long)(n * 8), hipGetErrorString(e));
  // synchronous: the caller's vector may change, or go, as soon as this returns
  if ((e = hipMemcpy(d, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice)) != hipSuccess) {
    (void)hipFree(d);
    return fail(ctx, HH_ERR_HIP, "hipMemcpy of the seeds failed: %s", hipGetErrorString(e));
  }
  victim->dev = d;
  victim->n = n;
  victim->fingerprint = fingerprint;
  victim->head = head;
  victim->tail = tail;
  victim->stamp = ++ctx->seed_clock;
  ++ctx->seed_uploads;
  *dev_out = d;
  return HH_OK;
}

int hh_seeds_cache_stats(hh_ctx* ctx, uint64_t* hits, uint64_t* uploads, uint64_t* evictions) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  if (hits) *hits = ctx->seed_hits;
  if (uploads) *uploads = ctx->seed_uploads;
  if (evictions) *evictions = ctx->seed_evictions;
  return HH_OK;
}

int hh_device_malloc(hh_ctx* ctx, size_t bytes, void** out_dev) {
  if (!ctx || !out_dev) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(out_dev, bytes);
  if (e != hipSuccess)
    return fail(ctx, HH_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  return HH_OK;
}

int hh_device_free(hh_ctx* ctx, void* dev) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hipFree(dev));
  return HH_OK;
}

int hh_memcpy_h2d(hh_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return HH_OK;
}

int hh_memcpy_d2h(hh_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  if (!ctx) return HH_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lock__(ctx->mu);
  HH_HIP(ctx, hipSetDevice(ctx->device));
  HH_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return HH_OK;
}

}  // extern "C"
