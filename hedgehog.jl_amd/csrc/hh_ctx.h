// The context object behind the C-ABI's opaque hh_ctx, and the helpers every host translation unit
// (hh_api.hip, hh_mgpu.hip) shares.  Internal; the public surface is include/hedgehog_mc.h.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <mutex>

#include "hh_kernels.h"

struct hh_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_switch = nullptr;
  // recorded behind the last copy FROM a caller's host buffer; staged_host: such a copy is queued and the entry
  // point has to wait for it (release_host_operands) before it returns — the caller owns its buffers again then
  hipEvent_t ev_stage = nullptr;
  bool staged_host = false;
  double* records = nullptr;
  size_t records_cap = 0;  // in records
  uint64_t* seeds = nullptr;
  size_t seeds_cap = 0;  // in elements
  double* replay = nullptr;  // tile-major staging
  size_t replay_cap = 0;
  double* replay_src = nullptr;  // path-major staging
  size_t replay_src_cap = 0;
  double* terminal = nullptr;
  size_t terminal_cap = 0;
  double* terminal_d = nullptr;  // [P][n_total] terminal partials (basket Greeks)
  size_t terminal_d_cap = 0;
  double* payoffs = nullptr;  // [2][n_payoffs]: strikes, cps
  size_t payoffs_cap = 0;
  double* basket_records = nullptr;
  size_t basket_records_cap = 0;
  double* basket_accum = nullptr;
  size_t basket_accum_cap = 0;
  unsigned char* bk_scratch = nullptr;
  size_t bk_scratch_cap = 0;
  hh::BkTableKey bk_table_key{};  // Bessel tables resident in bk_scratch (dropped by ensure_bk_scratch)
  double* lsm_grid = nullptr;  // [n_steps+1][ntot]
  size_t lsm_grid_cap = 0;
  double* heston_var = nullptr;  // [n_steps+1][n_paths] variance rows of the exact Heston grid
  size_t heston_var_cap = 0;
  double* lsm_val = nullptr;
  size_t lsm_val_cap = 0;
  int32_t* lsm_tau = nullptr;
  size_t lsm_tau_cap = 0;
  double* lsm_scratch = nullptr;
  size_t lsm_scratch_cap = 0;
  // sharded LSM in progress (hh_lsm_shard_begin .. hh_lsm_shard_finish)
  struct {
    bool active = false;
    hh_model m{};
    uint64_t ntot = 0;
    uint32_t n_steps = 0;
    int32_t degree = 0;
    double step_discount = 1.0;
  } shard;
  int lsm_form = hh::kLsmFormAuto;  // hh_ctx_set_option(HH_OPT_LSM_FORM)
  int bk_term_cache = 0;            // hh_ctx_set_option(HH_OPT_BK_TERM_CACHE); 0 = the default
  uint64_t bk_last_n = 0;           // trajectories of the last Broadie–Kaya solve (hh_bk_decisions)
  int bk_last_cache = 0;            // … and the term cache it ran with
  int grid_form = HH_GRID_FORM_BATCHED;  // hh_ctx_set_option(HH_OPT_GRID_FORM)
  int grid_order = 1;                    // hh_ctx_set_option(HH_OPT_GRID_ORDER): 1 = a chain's pairs sorted by their Bessel arguments
  unsigned char* bk_sort = nullptr;      // scratch of the ordered form (hh::bk_grid_sort_bytes)
  size_t bk_sort_cap = 0;
  uint64_t lsm_persistent_fallbacks = 0;  // persistent launches that gave up and were redone per date
  long long lsm_spin_ticks = -1;          // hh_ctx_set_option(HH_OPT_LSM_SPIN_TICKS); < 0 = the default (1 s)
  double* frecords = nullptr;      // records of the launches that reduce them themselves: kPoison between launches (hh_sim.h)
  size_t frecords_cap = 0;         // in doubles
  int fuse_reduce = 2;             // hh_ctx_set_option(HH_OPT_FUSE_REDUCE): 0 a second kernel, 1 in the simulation kernel, 2 by size
  unsigned int* finish_state = nullptr;  // device word: a reducer inside a simulation kernel gave up (hh_sim.h); cleared by recover_finish
  long long finish_spin_ticks = -1;      // hh_ctx_set_option(HH_OPT_FINISH_SPIN_TICKS); < 0 = the default (5 s)
  int finish_tile_first = 0;             // hh_ctx_set_option(HH_OPT_FINISH_TILE_FIRST)
  // seed vectors kept in device memory (hh_seeds_cache): content-addressed, least recently used one out
  struct SeedEntry {
    uint64_t* dev = nullptr;
    uint64_t n = 0, fingerprint = 0, head = 0, tail = 0, stamp = 0;
  };
  static constexpr int kSeedEntries = HH_SEED_CACHE_ENTRIES;
  SeedEntry seed_cache[kSeedEntries];
  uint64_t seed_clock = 0, seed_hits = 0, seed_uploads = 0, seed_evictions = 0;
  double* accum = nullptr;       // device, HH_ACC_LEN
  double* accum_host = nullptr;  // pinned, HH_ACC_LEN + 8 (LSM: row counters and the give-up word behind the accumulator)
  // optional per-launch timing of the simulation kernel (hh_ctx_enable_timing)
  static constexpr int kTimingSlots = 256;
  bool timing = false;
  hipEvent_t tev[kTimingSlots][2] = {};
  int t_count = 0;  // pairs recorded since the last read (capped at kTimingSlots)
  char err[512] = {0};
  std::recursive_mutex mu;  // entry points serialise on it: a ctx may be shared between threads
};

namespace {

[[maybe_unused]] int fail(hh_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define HH_HIP(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t e__ = (hipError_t)(expr);                                                    \
    if (e__ != hipSuccess)                                                                  \
      return fail(ctx, HH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                  __FILE__, __LINE__);                                                      \
  } while (0)

template <class T>
int ensure(hh_ctx* ctx, T*& buf, size_t& cap, size_t need) {
  if (need <= cap) return HH_OK;
  if (buf) HH_HIP(ctx, hipFree(buf));
  buf = nullptr;
  cap = 0;
  hipError_t e = hipMalloc((void**)&buf, need * sizeof(T));
  if (e != hipSuccess)
    return fail(ctx, HH_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", need * sizeof(T),
                hipGetErrorString(e));
  cap = need;
  return HH_OK;
}

// a copy from a caller's host buffer has just been queued on the ctx stream
[[maybe_unused]] inline int note_host_copy(hh_ctx* ctx) {
  HH_HIP(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
  ctx->staged_host = true;
  return HH_OK;
}
// Before an ASYNCHRONOUS entry point returns: every copy it queued from the caller's host memory has read its
// source (pageable memory is staged by the runtime before hipMemcpyAsync returns, PINNED memory is read by the
// DMA engine whenever the stream gets there) — only the copies are waited for, the kernels behind them run on.
[[maybe_unused]] inline int release_host_operands(hh_ctx* ctx) {
  if (!ctx->staged_host) return HH_OK;
  ctx->staged_host = false;
  HH_HIP(ctx, hipEventSynchronize(ctx->ev_stage));
  return HH_OK;
}

// The Broadie–Kaya scratch also holds the Bessel tables bk_table_key describes: a re-allocation may
// come back at the SAME address (hipFree + hipMalloc of a larger block), so the key is dropped with
// the old block and the tables are uploaded again.
[[maybe_unused]] inline int ensure_bk_scratch(hh_ctx* ctx, size_t need) {
  const size_t before = ctx->bk_scratch_cap;
  const int rc = ensure(ctx, ctx->bk_scratch, ctx->bk_scratch_cap, need);
  if (ctx->bk_scratch_cap != before) ctx->bk_table_key = hh::BkTableKey{};
  return rc;
}

}  // namespace
