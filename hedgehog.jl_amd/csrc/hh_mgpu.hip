// hh_mgpu_*: the path-sharded multi-GPU solve behind ONE call of the caller's host thread
// (include/hedgehog_mc.h).  What it stands in for is still ONE call of
// solve(prob, method) (montecarlo.jl:478-493) on one EnsembleProblem (:329-333,351): the trajectories
// are cut into contiguous ranges, every device runs the single-GPU kernel sequence of hh_mc_accumulate
// on its range and on its own stream, and the HH_ACC_LEN-double accumulator vectors are combined by
// one RCCL all-reduce over xGMI — or, when RCCL is not there or refuses, by an ordered sum on the
// host.  No arithmetic of the pricing path happens here except that ordered sum.
//
// Enqueueing a shard costs the host tens of microseconds (argument checks, two event records, two or
// three launches: profiles/r04_a_mgpu_enqueue.txt), which over eight devices in a row is a fifth of the
// 0.57 ms the kernels take.  So the shards are enqueued CONCURRENTLY: device 0 by the calling thread,
// device g > 0 by a library-owned worker thread bound to that device, which spins for a short window
// after each job (back-to-back solves find it awake) and parks on a condition variable otherwise.
// The exchange itself is issued by the calling thread once every shard is enqueued.
//
// A collective that fails for rank g after ranks < g were enqueued leaves kernels on those ranks'
// streams that wait for peers which never come.  Such a stream is never synchronised: the
// communicators are aborted, every shard stream is retired (replaced by a fresh one that waits for the
// event recorded BEFORE the collective), and the sums are finished on the host (rccl_give_up).
//
// RCCL is bound with dlopen at run time: the product library has no link-time dependency on it (a
// Julia host that never shards does not need librccl at all), and inside a PyTorch process the
// already-loaded librccl.so.1 — built against the HIP runtime the process uses — is what the soname
// resolves to.
#include <dlfcn.h>

#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include "hh_ctx.h"

namespace {

// the few RCCL declarations used (rccl.h: ncclComm_t, ncclResult_t, ncclSum = 0, ncclDouble = 8)
typedef struct ncclComm* rccl_comm_t;
constexpr int kNcclSuccess = 0, kNcclSum = 0, kNcclDouble = 8;

struct RcclApi {
  void* lib = nullptr;
  int (*CommInitAll)(rccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*CommAbort)(rccl_comm_t) = nullptr;  // required: what lets a collective that lost its peers leave its stream
  int (*GetVersion)(int*) = nullptr;        // optional (reported by hh_mgpu_rccl_info)
  char path[512] = {0};                     // the file ncclAllReduce was bound from (dladdr)
  int from_env = 0;                         // 1: $HEDGEHOG_MC_RCCL named it
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  char why[256] = {0};  // why it is not available
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = std::getenv("HEDGEHOG_MC_RCCL");
    if (env && *env) {
      api.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
      api.from_env = api.lib != nullptr;
    }
    // a copy the process has ALREADY loaded first (PyTorch ships its own librccl.so, built against the HIP
    // runtime the process runs on): the communicators then live on the same runtime as the streams
    const char* loaded[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : loaded)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) {
      snprintf(api.why, sizeof(api.why), "librccl not found (%s)", dlerror());
      return;
    }
    api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.CommAbort = (decltype(api.CommAbort))dlsym(api.lib, "ncclCommAbort");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    api.GetVersion = (decltype(api.GetVersion))dlsym(api.lib, "ncclGetVersion");
    Dl_info info{};
    if (api.AllReduce && dladdr((void*)api.AllReduce, &info) && info.dli_fname)
      snprintf(api.path, sizeof(api.path), "%s", info.dli_fname);
    // ncclCommAbort is part of the contract: after a collective that fails in the middle it is the only way
    // the ranks already enqueued leave their streams (rccl_give_up); a library without it is not used
    api.ok = api.CommInitAll && api.CommDestroy && api.CommAbort && api.AllReduce && api.GroupStart &&
             api.GroupEnd && api.GetErrorString;
    if (!api.ok) snprintf(api.why, sizeof(api.why), "librccl lacks an expected symbol");
  });
  return api;
}

}  // namespace

// One per device g >= 1 (device 0 is driven by the calling thread).
struct hh_worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<uint64_t> posted{0}, finished{0};
  std::atomic<int> parked{0}, stop{0};
  int rc = 0;
};

struct hh_mgpu {
  int n = 0;
  int flags = HH_MGPU_AUTO;
  int mode = HH_MGPU_REDUCE_HOST;
  bool rccl_lost = false;  // the communicators were aborted after a failed collective
  bool stuck = false;      // … and a shard is still on a stream that may hold the orphaned collective
  std::vector<int> devices;
  std::vector<hh_ctx*> ctx;
  std::vector<double*> acc, red;  // per device: local sums, all-reduced sums (RCCL writes out of place)
  std::vector<size_t> acc_cap;    // doubles
  std::vector<rccl_comm_t> comms;
  double* host = nullptr;  // pinned, n x acc_cap
  size_t host_cap = 0;
  std::vector<double*> xchg;  // per device: the exchange vector of a sharded LSM solve
  std::vector<size_t> xchg_cap;
  // concurrent enqueue
  int enqueue = HH_MGPU_ENQUEUE_THREADS;
  std::vector<std::unique_ptr<hh_worker>> workers;  // [g - 1], made on first use
  const std::function<int(int)>* job = nullptr;
  std::vector<int> rcs;
  std::vector<std::array<char, 320>> derr;  // per device: what failed inside a job (jobs never write err)
  std::vector<double> enq_us;               // host time of each shard's enqueue in the last solve
  double enq_phase_us = 0.0;                // … and of the whole enqueue phase (wall)
  // retiring streams that may hold an orphaned collective
  std::vector<hipEvent_t> pre;    // per device: recorded before a collective is enqueued
  std::vector<hipEvent_t> fence;  // per device: the event a retired stream's successor waits for
  struct Retired {
    int device;
    hipStream_t s;
  };
  std::vector<Retired> retired;
  char err[512] = {0};
  std::mutex mu;
};

namespace {

using clk = std::chrono::steady_clock;
inline double us_since(clk::time_point t0) {
  return std::chrono::duration<double, std::micro>(clk::now() - t0).count();
}
inline void cpu_relax() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  __builtin_ia32_pause();
#endif
}

int mfail(hh_mgpu* mg, int code, const char* fmt, ...) {
  if (mg) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(mg->err, sizeof(mg->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

// hh_mc_finalize refused the summed accumulator: because a shard's in-kernel record reduction gave up (every shard's
// context is asked — hh_ctx_check_last resets the ones it finds so — and the named status goes to the caller), or for
// what it says
int mfinalize_failed(hh_mgpu* mg, int rc) {
  for (int g = 0; g < mg->n; ++g)
    if (hh_ctx_check_last(mg->ctx[g]) == HH_ERR_DEVICE_TIMEOUT)
      rc = mfail(mg, HH_ERR_DEVICE_TIMEOUT, "device %d: %s", g, hh_last_error(mg->ctx[g]));
  return rc == HH_ERR_DEVICE_TIMEOUT ? rc : mfail(mg, rc, "finalize failed: the accumulator holds no trajectories");
}

#define HH_MHIP(mg, expr)                                                                     \
  do {                                                                                        \
    hipError_t e__ = (hipError_t)(expr);                                                      \
    if (e__ != hipSuccess)                                                                    \
      return mfail(mg, HH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),    \
                   __FILE__, __LINE__);                                                       \
  } while (0)

// inside a per-device job (possibly on a worker thread): the text goes to the device's own slot
#define HH_DHIP(mg, g, expr)                                                                       \
  do {                                                                                             \
    hipError_t e__ = (hipError_t)(expr);                                                           \
    if (e__ != hipSuccess) {                                                                       \
      snprintf(mg->derr[g].data(), mg->derr[g].size(), "%s failed: %s (%s:%d)", #expr,             \
               hipGetErrorString(e__), __FILE__, __LINE__);                                        \
      return HH_ERR_HIP;                                                                           \
    }                                                                                              \
  } while (0)

void shard_range(uint64_t n_paths, int n_dev, int g, bool tile_aligned, uint64_t* start, uint64_t* stop) {
  uint64_t per = (n_paths + (uint64_t)n_dev - 1) / (uint64_t)n_dev;
  if (tile_aligned) per = (per + hh::kTile - 1) / hh::kTile * hh::kTile;
  const uint64_t a = std::min(n_paths, (uint64_t)g * per);
  *start = a;
  *stop = std::min(n_paths, a + per);
}

// ---- the workers ------------------------------------------------------------------------------------

constexpr double kSpinWindowUs = 150.0;  // how long an idle worker polls before it parks

void worker_main(hh_mgpu* mg, int g) {
  hh_worker& w = *mg->workers[g - 1];
  (void)hipSetDevice(mg->devices[g]);
  uint64_t seen = 0;
  for (;;) {
    const auto idle0 = clk::now();
    while (w.posted.load() == seen && !w.stop.load()) {
      if (us_since(idle0) < kSpinWindowUs) {
        cpu_relax();
        continue;
      }
      std::unique_lock<std::mutex> lk(w.m);
      w.parked.store(1);
      w.cv.wait(lk, [&] { return w.posted.load() != seen || w.stop.load(); });
      w.parked.store(0);
    }
    if (w.stop.load()) return;
    ++seen;
    w.rc = (*mg->job)(g);
    w.finished.store(seen);
  }
}

void stop_workers(hh_mgpu* mg) {
  for (auto& w : mg->workers) {
    if (!w) continue;
    w->stop.store(1);
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->cv.notify_one();
    }
    if (w->th.joinable()) w->th.join();
  }
  mg->workers.clear();
}

// fn(g) for every device — concurrently when the context enqueues with threads — and the first failure
// in device order (every job has returned by then).  A job reports through mg->derr[g] only.
int for_each_device(hh_mgpu* mg, const std::function<int(int)>& fn) {
  for (int g = 0; g < mg->n; ++g) mg->derr[g][0] = 0;
  bool threads = mg->enqueue == HH_MGPU_ENQUEUE_THREADS && mg->n > 1;
  if (threads && mg->workers.empty()) {
    mg->workers.resize(mg->n - 1);
    for (int g = 1; g < mg->n && threads; ++g) {
      mg->workers[g - 1].reset(new (std::nothrow) hh_worker());
      if (!mg->workers[g - 1]) {
        threads = false;
        break;
      }
      try {
        mg->workers[g - 1]->th = std::thread(worker_main, mg, g);
      } catch (...) {
        threads = false;
      }
    }
    if (!threads) {  // no threads to be had: the shards are enqueued in a row from here on
      stop_workers(mg);
      mg->enqueue = HH_MGPU_ENQUEUE_SERIAL;
    }
  }
  if (!threads) {
    for (int g = 0; g < mg->n; ++g) mg->rcs[g] = fn(g);
  } else {
    mg->job = &fn;
    for (int g = 1; g < mg->n; ++g) {
      hh_worker& w = *mg->workers[g - 1];
      w.posted.fetch_add(1);
      if (w.parked.load()) {
        std::lock_guard<std::mutex> lk(w.m);
        w.cv.notify_one();
      }
    }
    mg->rcs[0] = fn(0);
    for (int g = 1; g < mg->n; ++g) {
      hh_worker& w = *mg->workers[g - 1];
      const uint64_t want = w.posted.load();
      const auto t0 = clk::now();
      while (w.finished.load() != want) {
        if (us_since(t0) < 2000.0) cpu_relax(); else std::this_thread::yield();
      }
      mg->rcs[g] = w.rc;
    }
    mg->job = nullptr;
  }
  for (int g = 0; g < mg->n; ++g)
    if (mg->rcs[g]) return mfail(mg, mg->rcs[g], "shard %d (device %d): %s", g, mg->devices[g], mg->derr[g].data());
  return HH_OK;
}

// ---- buffers ----------------------------------------------------------------------------------------

// the staging / accumulator buffers of an accumulator vector of n_acc doubles per device
int ensure_acc(hh_mgpu* mg, size_t n_acc) {
  for (int g = 0; g < mg->n; ++g) {
    if (mg->acc_cap[g] >= n_acc) continue;
    HH_MHIP(mg, hipSetDevice(mg->devices[g]));
    if (mg->acc[g]) HH_MHIP(mg, hipFree(mg->acc[g]));
    if (mg->red[g]) HH_MHIP(mg, hipFree(mg->red[g]));
    mg->acc[g] = mg->red[g] = nullptr;
    mg->acc_cap[g] = 0;
    if (hipMalloc((void**)&mg->acc[g], n_acc * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&mg->red[g], n_acc * sizeof(double)) != hipSuccess)
      return mfail(mg, HH_ERR_NOMEM, "hipMalloc of the accumulator vectors failed");
    mg->acc_cap[g] = n_acc;
  }
  if (mg->host_cap < n_acc * (size_t)mg->n) {
    if (mg->host) HH_MHIP(mg, hipHostFree(mg->host));
    mg->host = nullptr;
    mg->host_cap = 0;
    HH_MHIP(mg, hipHostMalloc((void**)&mg->host, n_acc * (size_t)mg->n * sizeof(double), hipHostMallocDefault));
    mg->host_cap = n_acc * (size_t)mg->n;
  }
  return HH_OK;
}

// Every failure AFTER something was enqueued leaves through here: earlier shards' kernels may still
// read the caller's buffers, and a sharded LSM must not stay half open.  Only ever called on streams
// that hold no collective (after rccl_give_up these are the fresh ones).
int drain(hh_mgpu* mg, int rc) {
  for (int g = 0; g < mg->n; ++g) {
    (void)hipSetDevice(mg->devices[g]);
    (void)hipStreamSynchronize(mg->ctx[g]->stream);
    mg->ctx[g]->shard.active = false;
  }
  return rc;
}
#define HH_MHIP_DRAIN(mg, expr)                                                                       \
  do {                                                                                                \
    hipError_t e__ = (hipError_t)(expr);                                                              \
    if (e__ != hipSuccess)                                                                            \
      return drain(mg, mfail(mg, HH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                             __FILE__, __LINE__));                                                    \
  } while (0)

// ---- the exchange -----------------------------------------------------------------------------------

// ONE grouped all-reduce over the shards' streams.  fence[g] must already name an event recorded on
// stream g behind the kernels that produced send[g] (and before anything of the collective).
int all_reduce_group(hh_mgpu* mg, double* const* send, double* const* recv, size_t n) {
  RcclApi& api = rccl();
  int e = api.GroupStart();
  for (int g = 0; g < mg->n && e == kNcclSuccess; ++g)
    e = api.AllReduce(send[g], recv[g], n, kNcclDouble, kNcclSum, mg->comms[g], mg->ctx[g]->stream);
  const int e2 = api.GroupEnd();  // launches what WAS enqueued, also after a failure in the middle
  return e == kNcclSuccess ? e2 : e;
}

// After a failed collective.  Ranks enqueued before the failure may sit in their streams waiting for
// peers that never arrive, so no shard stream is waited for again: the communicators are aborted (which
// lets such kernels leave), every stream is retired in favour of a fresh one that continues behind
// fence[g], and from here on this context sums on the host.  Returns with nothing synchronised; false when a
// shard could NOT be moved (a stream lent by the caller, or no fresh stream to be had): the context is then
// `stuck` — nothing may be finished on it, every call returns HH_ERR_RCCL once the caller's buffers are free.
bool rccl_give_up(hh_mgpu* mg) {
  RcclApi& api = rccl();
  for (rccl_comm_t c : mg->comms)
    if (c) (void)api.CommAbort(c);
  mg->comms.clear();
  mg->mode = HH_MGPU_REDUCE_HOST;
  mg->rccl_lost = true;
  bool all_moved = true;
  for (int g = 0; g < mg->n; ++g) {
    hh_ctx* c = mg->ctx[g];
    std::lock_guard<std::recursive_mutex> lock__(c->mu);
    c->shard.active = false;
    // a stream lent by the caller is the caller's to retire, and a fresh stream may be refused: such a shard
    // stays on the stream that may hold the orphan — the caller of this function must not finish on it
    hipStream_t fresh = nullptr;
    (void)hipSetDevice(mg->devices[g]);
    if (c->stream != c->own_stream || hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) {
      all_moved = false;
      continue;
    }
    if (mg->fence[g]) (void)hipStreamWaitEvent(fresh, mg->fence[g], 0);
    mg->retired.push_back({mg->devices[g], c->own_stream});
    c->own_stream = c->stream = fresh;
  }
  mg->stuck = mg->stuck || !all_moved;
  return all_moved;
}

// the caller's buffers are free once the kernels in front of the fences have run
void wait_fences(hh_mgpu* mg) {
  for (int g = 0; g < mg->n; ++g)
    if (mg->fence[g]) {
      (void)hipSetDevice(mg->devices[g]);
      (void)hipEventSynchronize(mg->fence[g]);
    }
}

struct Basket {
  const double *strikes, *cps;
  uint32_t n_payoffs;
  uint32_t n_models = 0;  // > 0: not a basket but n_models models on the same draws (hh_mc_accumulate_multi);
};                        //      `m` of run_shards then points to the first of them

// Enqueue every shard, combine the accumulator vectors, leave the combined vector in mg->host[0 .. n_acc).
// cfgs[g].n_paths == 0 leaves device g idle (it contributes zeros).  terminals[g] (nullable): device
// buffer for the shard's terminal samples, already part of cfgs[g] — nothing to do with it here.
int run_shards(hh_mgpu* mg, const hh_model* m, const hh_config* cfgs, const Basket* basket,
               double* const* terminals, double* kernel_ms) {
  const bool multi = basket && basket->n_models > 0;
  const size_t n_acc = (size_t)HH_ACC_LEN * (multi ? basket->n_models : basket ? basket->n_payoffs : 1u);
  if (mg->stuck)  // no drain: a shard's stream may hold a collective that never ends
    return mfail(mg, HH_ERR_RCCL, "a collective failed and a shard could not be moved off the stream it was enqueued on "
                                  "(a stream lent with hh_ctx_set_stream, or no new stream): create a new context");
  if (mg->flags == HH_MGPU_RCCL && mg->rccl_lost)
    return drain(mg, mfail(mg, HH_ERR_RCCL, "the RCCL communicators of this context were aborted after a failed "
                                            "collective (HH_MGPU_RCCL does not fall back): create a new one"));
  int rc = ensure_acc(mg, n_acc);
  if (rc) return drain(mg, rc);  // cut_config may have queued staging copies
  // 1. every shard's kernel sequence: nobody waits, the devices are served concurrently
  const auto t_enq = clk::now();
  rc = for_each_device(mg, [&](int g) -> int {
    const auto t0 = clk::now();
    hh_ctx* c = mg->ctx[g];
    HH_DHIP(mg, g, hipSetDevice(mg->devices[g]));
    HH_DHIP(mg, g, hipEventRecord(c->ev0, c->stream));
    if (cfgs[g].n_paths == 0) {
      HH_DHIP(mg, g, hipMemsetAsync(mg->acc[g], 0, n_acc * sizeof(double), c->stream));
    } else {
      double* term = terminals ? terminals[g] : nullptr;
      const int r = multi    ? hh_mc_accumulate_multi(c, m, basket->n_models, &cfgs[g], mg->acc[g], nullptr)
                    : basket ? hh_mc_accumulate_basket(c, m, &cfgs[g], basket->strikes, basket->cps,
                                                       basket->n_payoffs, mg->acc[g], term)
                             : hh_mc_accumulate(c, m, &cfgs[g], mg->acc[g], term);
      if (r) {
        snprintf(mg->derr[g].data(), mg->derr[g].size(), "%s", hh_last_error(c));
        return r;
      }
    }
    HH_DHIP(mg, g, hipEventRecord(c->ev1, c->stream));
    mg->fence[g] = c->ev1;
    mg->enq_us[g] = us_since(t0);
    return HH_OK;
  });
  mg->enq_phase_us = us_since(t_enq);
  if (rc) return drain(mg, rc);  // earlier shards still read the caller's buffers
  // 2. the path's one exchange
  bool reduced = false;
  if (mg->mode == HH_MGPU_REDUCE_RCCL) {
    const int e = all_reduce_group(mg, mg->acc.data(), mg->red.data(), n_acc);
    if (e == kNcclSuccess) {
      HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[0]));
      HH_MHIP_DRAIN(mg, hipMemcpyAsync(mg->host, mg->red[0], n_acc * sizeof(double), hipMemcpyDeviceToHost,
                                       mg->ctx[0]->stream));
      reduced = true;
    } else {
      mfail(mg, HH_ERR_RCCL, "ncclAllReduce failed: %s — communicators aborted, accumulators are summed on the host",
            rccl().GetErrorString(e));
      const bool moved = rccl_give_up(mg);
      if (mg->flags == HH_MGPU_RCCL || !moved) {
        wait_fences(mg);
        return HH_ERR_RCCL;
      }  // else: the local sums are untouched (out-of-place reduce) — finish on the host, on the fresh streams
    }
  }
  if (!reduced) {
    for (int g = 0; g < mg->n; ++g) {
      HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
      HH_MHIP_DRAIN(mg, hipMemcpyAsync(mg->host + (size_t)g * n_acc, mg->acc[g], n_acc * sizeof(double),
                                       hipMemcpyDeviceToHost, mg->ctx[g]->stream));
    }
  }
  // 3. wait for every device (the caller's buffers are free again after this), then the ordered sum
  for (int g = 0; g < mg->n; ++g) {
    HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
    HH_MHIP_DRAIN(mg, hipStreamSynchronize(mg->ctx[g]->stream));
  }
  if (!reduced) {
    for (int g = 1; g < mg->n; ++g)  // fixed order g = 0 … G-1: deterministic for a given sharding
      for (size_t i = 0; i < n_acc; ++i) mg->host[i] += mg->host[(size_t)g * n_acc + i];
  }
  double worst = 0.0;
  for (int g = 0; g < mg->n; ++g) {
    float ms = 0.f;
    HH_MHIP(mg, hipSetDevice(mg->devices[g]));
    HH_MHIP(mg, hipEventElapsedTime(&ms, mg->ctx[g]->ev0, mg->ctx[g]->ev1));
    if (ms > worst) worst = ms;
  }
  *kernel_ms = worst;
  return HH_OK;
}

int ncomp(int dynamics) { return dynamics == HH_HESTON ? 2 : 1; }

}  // namespace

extern "C" {

void hh_mgpu_shard_range(uint64_t n_paths, int n_devices, int g, int tile_aligned, uint64_t* start,
                         uint64_t* stop) {
  uint64_t a = 0, b = 0;
  if (n_devices > 0 && g >= 0 && g < n_devices) shard_range(n_paths, n_devices, g, tile_aligned != 0, &a, &b);
  if (start) *start = a;
  if (stop) *stop = b;
}

int hh_mgpu_create(hh_mgpu** out, const int* device_ids, int n_devices, int flags) {
  if (!out) return HH_ERR_INVALID;
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 64) return HH_ERR_INVALID;
  if (flags != HH_MGPU_AUTO && flags != HH_MGPU_HOST_SUM && flags != HH_MGPU_RCCL) return HH_ERR_INVALID;
  int n_hip = 0;
  if (hipGetDeviceCount(&n_hip) != hipSuccess || n_hip <= 0) return HH_ERR_HIP;  // no CPU fallback
  for (int g = 0; g < n_devices; ++g)
    if (device_ids[g] < 0 || device_ids[g] >= n_hip) return HH_ERR_INVALID;
  hh_mgpu* mg = new (std::nothrow) hh_mgpu();
  if (!mg) return HH_ERR_NOMEM;
  mg->n = n_devices;
  mg->flags = flags;
  mg->devices.assign(device_ids, device_ids + n_devices);
  mg->ctx.assign(n_devices, nullptr);
  mg->acc.assign(n_devices, nullptr);
  mg->red.assign(n_devices, nullptr);
  mg->acc_cap.assign(n_devices, 0);
  mg->xchg.assign(n_devices, nullptr);
  mg->xchg_cap.assign(n_devices, 0);
  mg->rcs.assign(n_devices, 0);
  mg->derr.resize(n_devices);
  mg->enq_us.assign(n_devices, 0.0);
  mg->pre.assign(n_devices, nullptr);
  mg->fence.assign(n_devices, nullptr);
  for (int g = 0; g < n_devices; ++g) {
    int rc = hh_ctx_create(&mg->ctx[g], device_ids[g]);
    if (!rc && hipEventCreateWithFlags(&mg->pre[g], hipEventDisableTiming) != hipSuccess) rc = HH_ERR_HIP;
    if (rc) {
      hh_mgpu_destroy(mg);
      return rc;
    }
  }
  const bool want = flags == HH_MGPU_RCCL || (flags == HH_MGPU_AUTO && n_devices > 1);
  if (want) {
    RcclApi& api = rccl();
    int e = -1;
    if (api.ok) {
      mg->comms.assign(n_devices, nullptr);
      e = api.CommInitAll(mg->comms.data(), n_devices, device_ids);  // refuses a device listed twice
      if (e != kNcclSuccess) {
        mg->comms.clear();
        mfail(mg, HH_ERR_RCCL, "ncclCommInitAll failed: %s — accumulators are summed on the host",
              api.GetErrorString(e));
      }
    } else {
      mfail(mg, HH_ERR_RCCL, "RCCL unavailable: %s — accumulators are summed on the host", api.why);
    }
    if (e == kNcclSuccess) {
      mg->mode = HH_MGPU_REDUCE_RCCL;
    } else if (flags == HH_MGPU_RCCL) {
      hh_mgpu_destroy(mg);
      return HH_ERR_RCCL;
    }
  }
  *out = mg;
  return HH_OK;
}

void hh_mgpu_destroy(hh_mgpu* mg) {
  if (!mg) return;
  stop_workers(mg);
  // A retired stream that has not drained may hold a collective whose peers never came (ncclCommAbort is
  // supposed to release it; until it has, it has not).  hipDeviceSynchronize and hipFree wait for EVERY stream
  // of the device, that one included: a device with such a stream — and a stuck context altogether — is left
  // alone (its streams, buffers and hh_ctx are leaked) rather than waited for.
  std::vector<char> busy(mg->n, mg->stuck ? 1 : 0);
  for (const auto& r : mg->retired) {
    (void)hipSetDevice(r.device);
    if (hipStreamQuery(r.s) == hipSuccess) {
      (void)hipStreamDestroy(r.s);
    } else {
      for (int g = 0; g < mg->n; ++g)
        if (mg->devices[g] == r.device) busy[g] = 1;
    }
  }
  for (int g = 0; g < mg->n; ++g) {
    if (!mg->ctx[g] || busy[g]) continue;
    (void)hipSetDevice(mg->devices[g]);
    (void)hipStreamSynchronize(mg->ctx[g]->stream);
  }
  for (rccl_comm_t c : mg->comms)
    if (c) (void)rccl().CommDestroy(c);
  for (int g = 0; g < mg->n; ++g) {
    if (busy[g]) continue;
    if (mg->pre[g]) (void)hipEventDestroy(mg->pre[g]);
    if (!mg->ctx[g]) continue;  // creation stopped before this device: nothing of it exists
    (void)hipSetDevice(mg->devices[g]);
    if (mg->acc[g]) (void)hipFree(mg->acc[g]);
    if (mg->red[g]) (void)hipFree(mg->red[g]);
    if (mg->xchg[g]) (void)hipFree(mg->xchg[g]);
    if (mg->ctx[g]) hh_ctx_destroy(mg->ctx[g]);
  }
  bool any_busy = false;
  for (char b : busy) any_busy = any_busy || b;
  if (mg->host && !any_busy) (void)hipHostFree(mg->host);
  delete mg;
}

const char* hh_mgpu_last_error(const hh_mgpu* mg) { return mg ? mg->err : "hedgehog_mc: no multi-GPU context"; }
int hh_mgpu_n_devices(const hh_mgpu* mg) { return mg ? mg->n : 0; }
int hh_mgpu_reduce_mode(const hh_mgpu* mg) { return mg ? mg->mode : HH_MGPU_REDUCE_HOST; }
hh_ctx* hh_mgpu_ctx(hh_mgpu* mg, int i) { return (mg && i >= 0 && i < mg->n) ? mg->ctx[i] : nullptr; }

int hh_mgpu_selftest(hh_mgpu* mg, int32_t* ranks_out, int32_t* reduce_mode_out) {
  if (!mg) return HH_ERR_INVALID;
  std::lock_guard<std::mutex> lock__(mg->mu);
  if (!ranks_out) return mfail(mg, HH_ERR_INVALID, "hh_mgpu_selftest: ranks_out is NULL");
  if (mg->stuck) return mfail(mg, HH_ERR_RCCL, "stuck context (see hh_mgpu_last_error of the failed call)");
  int rc = ensure_acc(mg, HH_ACC_LEN);
  if (rc) return rc;
  // every device contributes a vector of ones through the exchange a solve uses: what comes back counts the
  // ranks that took part — the number a caller may print next to "reduce: rccl"
  std::vector<double> ones(HH_ACC_LEN, 1.0);
  for (int g = 0; g < mg->n; ++g) {
    HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
    HH_MHIP_DRAIN(mg, hipMemcpyAsync(mg->acc[g], ones.data(), HH_ACC_LEN * sizeof(double), hipMemcpyHostToDevice,
                                     mg->ctx[g]->stream));
    HH_MHIP_DRAIN(mg, hipMemsetAsync(mg->red[g], 0, HH_ACC_LEN * sizeof(double), mg->ctx[g]->stream));
    HH_MHIP_DRAIN(mg, hipEventRecord(mg->pre[g], mg->ctx[g]->stream));
    mg->fence[g] = mg->pre[g];
  }
  for (int g = 0; g < mg->n; ++g) {  // `ones` is pageable: the copies have read it once the streams are idle
    HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
    HH_MHIP_DRAIN(mg, hipStreamSynchronize(mg->ctx[g]->stream));
  }
  double total = 0.0;
  if (mg->mode == HH_MGPU_REDUCE_RCCL) {
    const int e = all_reduce_group(mg, mg->acc.data(), mg->red.data(), HH_ACC_LEN);
    if (e != kNcclSuccess) {
      mfail(mg, HH_ERR_RCCL, "ncclAllReduce failed in the self-test: %s — communicators aborted", rccl().GetErrorString(e));
      (void)rccl_give_up(mg);
      wait_fences(mg);
      return HH_ERR_RCCL;
    }
    for (int g = 0; g < mg->n; ++g) {  // every rank must hold the same count in every slot
      HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
      HH_MHIP_DRAIN(mg, hipMemcpyAsync(mg->host + (size_t)g * HH_ACC_LEN, mg->red[g], HH_ACC_LEN * sizeof(double),
                                       hipMemcpyDeviceToHost, mg->ctx[g]->stream));
    }
    for (int g = 0; g < mg->n; ++g) {
      HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
      HH_MHIP_DRAIN(mg, hipStreamSynchronize(mg->ctx[g]->stream));
    }
    total = mg->host[0];
    for (size_t i = 0; i < (size_t)mg->n * HH_ACC_LEN; ++i)
      if (mg->host[i] != total) {
        // a collective that returns different sums to different ranks must not carry a solve: as after a failed one,
        // the communicators are aborted and later solves add on the host (or fail, where RCCL was required)
        mfail(mg, HH_ERR_RCCL, "self-test: the ranks disagree on the all-reduced count (%g vs %g) — communicators aborted",
              mg->host[i], total);
        (void)rccl_give_up(mg);
        wait_fences(mg);
        return HH_ERR_RCCL;
      }
  } else {
    for (int g = 0; g < mg->n; ++g) {
      HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
      HH_MHIP_DRAIN(mg, hipMemcpyAsync(mg->host + (size_t)g * HH_ACC_LEN, mg->acc[g], HH_ACC_LEN * sizeof(double),
                                       hipMemcpyDeviceToHost, mg->ctx[g]->stream));
    }
    for (int g = 0; g < mg->n; ++g) {
      HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
      HH_MHIP_DRAIN(mg, hipStreamSynchronize(mg->ctx[g]->stream));
    }
    for (int g = 0; g < mg->n; ++g) total += mg->host[(size_t)g * HH_ACC_LEN];
  }
  *ranks_out = (int32_t)(total + 0.5);
  if (reduce_mode_out) *reduce_mode_out = mg->mode;
  return HH_OK;
}

int hh_mgpu_rccl_info(const hh_mgpu* mg, char* path_out, size_t cap, int32_t* version_out, int32_t* from_env_out) {
  (void)mg;
  RcclApi& api = rccl();
  if (path_out && cap) snprintf(path_out, cap, "%s", api.lib ? api.path : "");
  int v = 0;
  if (api.lib && api.GetVersion && api.GetVersion(&v) != kNcclSuccess) v = 0;
  if (version_out) *version_out = v;
  if (from_env_out) *from_env_out = api.from_env;
  return api.ok ? HH_OK : HH_ERR_RCCL;
}

int hh_mgpu_set_option(hh_mgpu* mg, int32_t option, int64_t value) {
  if (!mg) return HH_ERR_INVALID;
  std::lock_guard<std::mutex> lock__(mg->mu);
  switch (option) {
    case HH_MGPU_OPT_ENQUEUE:
      if (value != HH_MGPU_ENQUEUE_SERIAL && value != HH_MGPU_ENQUEUE_THREADS)
        return mfail(mg, HH_ERR_INVALID, "HH_MGPU_OPT_ENQUEUE: 0 (one shard after the other) or 1 (a thread per device)");
      if (value == HH_MGPU_ENQUEUE_SERIAL) stop_workers(mg);
      mg->enqueue = (int)value;
      return HH_OK;
    default:
      return mfail(mg, HH_ERR_INVALID, "unknown option %d", option);
  }
}

int hh_mgpu_enqueue_stats(hh_mgpu* mg, double* shard_us, double* phase_us) {
  if (!mg) return HH_ERR_INVALID;
  std::lock_guard<std::mutex> lock__(mg->mu);
  if (shard_us)
    for (int g = 0; g < mg->n; ++g) shard_us[g] = mg->enq_us[g];
  if (phase_us) *phase_us = mg->enq_phase_us;
  return HH_OK;
}

int hh_mgpu_solve_shards(hh_mgpu* mg, const hh_model* m, const hh_config* cfgs, hh_result* out,
                         double* const* terminals) {
  if (!mg) return HH_ERR_INVALID;
  std::lock_guard<std::mutex> lock__(mg->mu);
  if (!m || !cfgs || !out) return mfail(mg, HH_ERR_INVALID, "hh_mgpu_solve_shards: NULL argument");
  const auto t0 = std::chrono::steady_clock::now();
  int first = -1;
  for (int g = 0; g < mg->n && first < 0; ++g)
    if (cfgs[g].n_paths) first = g;
  if (first < 0) return mfail(mg, HH_ERR_INVALID, "every shard is empty");
  std::memset(out, 0, sizeof(*out));
  double kernel_ms = 0.0;
  int rc = run_shards(mg, m, cfgs, nullptr, terminals, &kernel_ms);
  if (rc) return rc;
  rc = hh_mc_finalize(m, &cfgs[first], mg->host, out);  // reads n_partials and the discount seeds only
  if (rc) return mfinalize_failed(mg, rc);
  out->kernel_ms = kernel_ms;
  out->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return HH_OK;
}

// Shards of a whole-ensemble config with host buffers.  Host slices go to hh_mc_accumulate as they are
// (it stages them on the shard's stream, from the shard's thread) unless the kernels' operand rules say
// otherwise: BK REPLAY draws [V_T | u | Z] are three slices per shard, and a slice that does not start on
// a 16-byte boundary (one double per trajectory — the exact law, lognormal Euler with one step — or an
// odd row length, cut at an odd trajectory) is staged here into the shard ctx's own buffer.
static int cut_config(hh_mgpu* mg, const hh_config* cfg, bool want_terminal, std::vector<hh_config>& cs,
                      std::vector<double*>& term_dev, std::vector<uint64_t>& starts) {
  if (cfg->n_paths == 0) return mfail(mg, HH_ERR_INVALID, "n_paths must be >= 1");
  if (cfg->seeds_on_device || cfg->replay_on_device || cfg->terminal_on_device)
    return mfail(mg, HH_ERR_INVALID,
                 "hh_mgpu_solve takes host buffers; device-resident shards go through hh_mgpu_solve_shards");
  const bool euler = cfg->strategy == HH_EULER_MARUYAMA;
  const bool replay = cfg->noise_mode == HH_NOISE_REPLAY;
  const bool bk = cfg->strategy == HH_BROADIE_KAYA;
  const bool tile = replay && euler && cfg->replay_layout == HH_REPLAY_TILE_MAJOR;
  if (replay && !cfg->replay) return mfail(mg, HH_ERR_INVALID, "REPLAY needs a replay buffer");
  if (replay && ((uintptr_t)cfg->replay & 15u) != 0)
    return mfail(mg, HH_ERR_INVALID, "replay buffer must be 16-byte aligned");
  if (!replay && !cfg->seeds) return mfail(mg, HH_ERR_INVALID, "GENERATE needs seeds");
  const uint64_t N = cfg->n_paths;
  const size_t per_path = euler ? (size_t)cfg->n_steps * ncomp(cfg->dynamics) : 1;
  if (replay && cfg->replay_len) {  // whole-ensemble operand shape, as hh_mc_solve would check it
    const uint64_t need = bk ? 3 * N : tile ? (uint64_t)hh::tiles_for(N) * hh::kTile * per_path : N * per_path;
    if (cfg->replay_len < need)
      return mfail(mg, HH_ERR_INVALID, "replay buffer holds %llu elements, %llu needed",
                   (unsigned long long)cfg->replay_len, (unsigned long long)need);
  }
  if (!replay && cfg->seeds_len && cfg->seeds_len < (euler ? N : 1))
    return mfail(mg, HH_ERR_INVALID, "Number of seeds (%llu) must be >= number of trajectories (%llu)",
                 (unsigned long long)cfg->seeds_len, (unsigned long long)(euler ? N : 1));
  cs.assign(mg->n, *cfg);
  term_dev.assign(mg->n, nullptr);
  starts.assign(mg->n, 0);
  for (int g = 0; g < mg->n; ++g) {
    uint64_t a, b;
    shard_range(N, mg->n, g, tile, &a, &b);
    hh_config& c = cs[g];
    hh_ctx* x = mg->ctx[g];
    starts[g] = a;
    c.n_paths = b - a;
    c.path_offset = cfg->path_offset + a;
    c.seeds_len = c.replay_len = 0;
    if (c.n_paths == 0) continue;
    if (!replay && euler) c.seeds = cfg->seeds + a;
    if (replay && !bk) c.replay = cfg->replay + a * per_path;  // a is a multiple of 256 for tile-major data
    HH_MHIP(mg, hipSetDevice(mg->devices[g]));
    if (replay && bk) {
      int rc = ensure(x, x->replay, x->replay_cap, (size_t)3 * c.n_paths);
      if (rc) return mfail(mg, rc, "%s", x->err);
      for (int k = 0; k < 3; ++k)
        HH_MHIP(mg, hipMemcpyAsync(x->replay + (size_t)k * c.n_paths, cfg->replay + (size_t)k * N + a,
                                   c.n_paths * sizeof(double), hipMemcpyHostToDevice, x->stream));
      c.replay = x->replay;
      c.replay_on_device = 1;
    } else if (replay && ((uintptr_t)c.replay & 15u) != 0) {
      // path-major rows go where hh_mc_accumulate would stage them itself (replay_src: both the direct
      // kernel and the repack read from there); one normal per trajectory of the exact law likewise
      const bool pm = cfg->replay_layout == HH_REPLAY_PATH_MAJOR;
      const size_t n_el = (size_t)c.n_paths * per_path;
      int rc = pm ? ensure(x, x->replay_src, x->replay_src_cap, n_el)
                  : ensure(x, x->replay, x->replay_cap, (size_t)hh::tiles_for(c.n_paths) * hh::kTile);
      if (rc) return mfail(mg, rc, "%s", x->err);
      double* dst = pm ? x->replay_src : x->replay;
      HH_MHIP(mg, hipMemcpyAsync(dst, c.replay, n_el * sizeof(double), hipMemcpyHostToDevice, x->stream));
      c.replay = dst;
      c.replay_on_device = 1;
    }
    if (want_terminal) {
      int rc = ensure(x, x->terminal, x->terminal_cap, (size_t)c.n_paths * (c.antithetic ? 2 : 1));
      if (rc) return mfail(mg, rc, "%s", x->err);
      term_dev[g] = x->terminal;
      c.terminal_on_device = 1;
    }
  }
  return HH_OK;
}

int hh_mgpu_solve(hh_mgpu* mg, const hh_model* m, const hh_config* cfg, hh_result* out, double* terminal) {
  if (!mg) return HH_ERR_INVALID;
  std::lock_guard<std::mutex> lock__(mg->mu);
  if (!m || !cfg || !out) return mfail(mg, HH_ERR_INVALID, "hh_mgpu_solve: NULL argument");
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<hh_config> cs;
  std::vector<double*> term_dev;
  std::vector<uint64_t> starts;
  int rc = cut_config(mg, cfg, terminal != nullptr, cs, term_dev, starts);
  if (rc) return drain(mg, rc);
  std::memset(out, 0, sizeof(*out));
  double kernel_ms = 0.0;
  rc = run_shards(mg, m, cs.data(), nullptr, terminal ? term_dev.data() : nullptr, &kernel_ms);
  if (rc) return rc;
  if (terminal) {  // the shards' samples into the caller's whole-ensemble layout [N] (+ [N] mirrored)
    for (int g = 0; g < mg->n; ++g) {
      const uint64_t n = cs[g].n_paths;
      if (!n) continue;
      HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
      HH_MHIP_DRAIN(mg, hipMemcpyAsync(terminal + starts[g], term_dev[g], n * sizeof(double), hipMemcpyDeviceToHost,
                                       mg->ctx[g]->stream));
      if (cfg->antithetic)
        HH_MHIP_DRAIN(mg, hipMemcpyAsync(terminal + cfg->n_paths + starts[g], term_dev[g] + n, n * sizeof(double),
                                         hipMemcpyDeviceToHost, mg->ctx[g]->stream));
    }
    for (int g = 0; g < mg->n; ++g) {
      HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
      HH_MHIP_DRAIN(mg, hipStreamSynchronize(mg->ctx[g]->stream));
    }
  }
  rc = hh_mc_finalize(m, cfg, mg->host, out);
  if (rc) return mfinalize_failed(mg, rc);
  out->kernel_ms = kernel_ms;
  out->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return HH_OK;
}

int hh_mgpu_solve_multi(hh_mgpu* mg, const hh_model* models, uint32_t n_models, const hh_config* cfg,
                        hh_result* out) {
  if (!mg) return HH_ERR_INVALID;
  std::lock_guard<std::mutex> lock__(mg->mu);
  if (!models || !cfg || !out || n_models == 0 || n_models > HH_MAX_MODELS)
    return mfail(mg, HH_ERR_INVALID, "hh_mgpu_solve_multi: 1 .. %d models and their results", HH_MAX_MODELS);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<hh_config> cs;
  std::vector<double*> term_dev;
  std::vector<uint64_t> starts;
  int rc = cut_config(mg, cfg, false, cs, term_dev, starts);
  if (rc) return drain(mg, rc);
  Basket b{nullptr, nullptr, 0, n_models};
  double kernel_ms = 0.0;
  rc = run_shards(mg, models, cs.data(), &b, nullptr, &kernel_ms);
  if (rc) return rc;
  const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (uint32_t k = 0; k < n_models; ++k) {
    std::memset(&out[k], 0, sizeof(hh_result));
    rc = hh_mc_finalize(&models[k], cfg, mg->host + (size_t)k * HH_ACC_LEN, &out[k]);
    if (rc) return mfinalize_failed(mg, rc);
    out[k].kernel_ms = kernel_ms;
    out[k].total_ms = total;
  }
  return HH_OK;
}

int hh_mgpu_solve_basket(hh_mgpu* mg, const hh_model* m, const hh_config* cfg, const double* strikes,
                         const double* cps, uint32_t n_payoffs, hh_result* out) {
  if (!mg) return HH_ERR_INVALID;
  std::lock_guard<std::mutex> lock__(mg->mu);
  if (!m || !cfg || !out || !strikes || !cps || n_payoffs == 0 || n_payoffs > 65535)
    return mfail(mg, HH_ERR_INVALID, "hh_mgpu_solve_basket: bad arguments");
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<hh_config> cs;
  std::vector<double*> term_dev;
  std::vector<uint64_t> starts;
  int rc = cut_config(mg, cfg, false, cs, term_dev, starts);
  if (rc) return drain(mg, rc);
  const Basket b{strikes, cps, n_payoffs};
  double kernel_ms = 0.0;
  rc = run_shards(mg, m, cs.data(), &b, nullptr, &kernel_ms);
  if (rc) return rc;
  const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (uint32_t k = 0; k < n_payoffs; ++k) {
    std::memset(&out[k], 0, sizeof(hh_result));
    rc = hh_mc_finalize(m, cfg, mg->host + (size_t)k * HH_ACC_LEN, &out[k]);
    if (rc) return mfinalize_failed(mg, rc);
    out[k].kernel_ms = kernel_ms;
    out[k].total_ms = total;
  }
  return HH_OK;
}

// SUM all-reduce of the first n doubles of every device's exchange vector, in place.  HH_ERR_RCCL: the
// collective failed, rccl_give_up has run (nothing may be synchronised; the vectors are lost).
static int lsm_exchange(hh_mgpu* mg, size_t n) {
  if (mg->n == 1) return HH_OK;
  if (mg->mode == HH_MGPU_REDUCE_RCCL) {
    for (int g = 0; g < mg->n; ++g) {  // what a retired stream's successor would wait for
      HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
      HH_MHIP_DRAIN(mg, hipEventRecord(mg->pre[g], mg->ctx[g]->stream));
      mg->fence[g] = mg->pre[g];
    }
    const int e = all_reduce_group(mg, mg->xchg.data(), mg->xchg.data(), n);
    if (e == kNcclSuccess) return HH_OK;
    mfail(mg, HH_ERR_RCCL, "ncclAllReduce failed inside the LSM induction: %s — communicators aborted",
          rccl().GetErrorString(e));
    rccl_give_up(mg);
    return HH_ERR_RCCL;
  }
  // host ordered sum: local vectors back, added in the order g = 0 … G-1, the total out again
  int rc = ensure_acc(mg, n > (size_t)HH_ACC_LEN ? n : (size_t)HH_ACC_LEN);
  if (rc) return drain(mg, rc);
  for (int g = 0; g < mg->n; ++g) {
    HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
    HH_MHIP_DRAIN(mg, hipMemcpyAsync(mg->host + (size_t)g * n, mg->xchg[g], n * sizeof(double), hipMemcpyDeviceToHost,
                                     mg->ctx[g]->stream));
  }
  for (int g = 0; g < mg->n; ++g) {
    HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
    HH_MHIP_DRAIN(mg, hipStreamSynchronize(mg->ctx[g]->stream));
  }
  for (int g = 1; g < mg->n; ++g)
    for (size_t i = 0; i < n; ++i) mg->host[i] += mg->host[(size_t)g * n + i];
  for (int g = 0; g < mg->n; ++g) {
    HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
    HH_MHIP_DRAIN(mg, hipMemcpyAsync(mg->xchg[g], mg->host, n * sizeof(double), hipMemcpyHostToDevice, mg->ctx[g]->stream));
  }
  for (int g = 0; g < mg->n; ++g) {  // mg->host is reused by the next exchange
    HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
    HH_MHIP_DRAIN(mg, hipStreamSynchronize(mg->ctx[g]->stream));
  }
  return HH_OK;
}

// One pass of the sharded induction in the context's current reduce mode.  HH_ERR_RCCL: see lsm_exchange.
static int lsm_attempt(hh_mgpu* mg, const hh_model* m, const hh_config* cfg, int32_t degree, double step_discount,
                       hh_lsm_result* out, int32_t* stop_time, double* stop_value) {
  const auto t0 = std::chrono::steady_clock::now();
  const uint64_t N = cfg->n_paths;
  const uint32_t steps = cfg->n_steps;
  const size_t rows = (size_t)steps + 1, nv = 2 * (size_t)degree + 1, nb = (size_t)degree + 1;
  size_t n_x = hh_lsm_shard_xchg_elems(steps, degree);
  if (n_x < (size_t)HH_ACC_LEN) n_x = HH_ACC_LEN;  // the same vector carries the final accumulator
  std::vector<hh_config> cs(mg->n, *cfg);
  std::vector<uint64_t> starts(mg->n, 0);
  for (int g = 0; g < mg->n; ++g) {
    uint64_t a, b;
    shard_range(N, mg->n, g, false, &a, &b);
    if (b <= a) return mfail(mg, HH_ERR_INVALID, "every device needs at least one trajectory");
    starts[g] = a;
    cs[g].n_paths = b - a;
    cs[g].seeds = cfg->seeds + a;  // trajectory i keyed by seeds[i] (montecarlo.jl:331), both path sources
    cs[g].seeds_len = 0;
    if (mg->xchg_cap[g] < n_x) {
      HH_MHIP(mg, hipSetDevice(mg->devices[g]));
      if (mg->xchg[g]) HH_MHIP(mg, hipFree(mg->xchg[g]));
      mg->xchg[g] = nullptr;
      mg->xchg_cap[g] = 0;
      if (hipMalloc((void**)&mg->xchg[g], n_x * sizeof(double)) != hipSuccess)
        return mfail(mg, HH_ERR_NOMEM, "hipMalloc of the LSM exchange vector failed");
      mg->xchg_cap[g] = n_x;
    }
  }
  int rc = ensure_acc(mg, (size_t)HH_ACC_LEN);
  if (rc) return rc;
  auto shard_call = [&](int g, int r) -> int {  // a shard entry point's status, with its text
    if (r) snprintf(mg->derr[g].data(), mg->derr[g].size(), "%s", hh_last_error(mg->ctx[g]));
    return r;
  };
  rc = for_each_device(mg, [&](int g) -> int {
    HH_DHIP(mg, g, hipSetDevice(mg->devices[g]));
    HH_DHIP(mg, g, hipEventRecord(mg->ctx[g]->ev0, mg->ctx[g]->stream));
    return shard_call(g, hh_lsm_shard_begin(mg->ctx[g], m, &cs[g], degree, step_discount, mg->xchg[g]));
  });
  if (rc) return drain(mg, rc);
  auto phase = [&](int32_t ph, uint32_t t, size_t n_in) -> int {
    int e = lsm_exchange(mg, n_in);
    if (e) return e;  // already drained, or (HH_ERR_RCCL) given up without a wait
    // one launch per device: cheaper from this thread in a row (5 µs each) than a hand-off to the workers
    // and back (profiles/r04_a_mgpu_enqueue.txt: 13.8 against 15.4 ms for 100 dates over four shards)
    for (int g = 0; g < mg->n; ++g)
      if ((e = hh_lsm_shard_phase(mg->ctx[g], ph, t, mg->xchg[g], mg->xchg[g])))
        return drain(mg, mfail(mg, e, "shard %d (device %d): %s", g, mg->devices[g], hh_last_error(mg->ctx[g])));
    return HH_OK;
  };
  if ((rc = phase(HH_LSM_PHASE_POW, 0, rows * 3))) return rc;
  if ((rc = phase(HH_LSM_PHASE_INIT, 0, rows * nv))) return rc;
  for (uint32_t t = steps - 1; t >= 1; --t)  // for i = nsteps:-1:2, t = i-1 (:112-113)
    if ((rc = phase(HH_LSM_PHASE_STEP, t, nb))) return rc;
  // the shards' Σ, Σ² of the discounted stopped values; their stopping_info into the caller's order
  const int halves = cfg->antithetic ? 2 : 1;
  std::vector<std::vector<int32_t>> tau(mg->n);
  std::vector<std::vector<double>> val(mg->n);
  std::vector<uint32_t> rg(mg->n, 0), sk(mg->n, 0);
  rc = for_each_device(mg, [&](int g) -> int {
    const uint64_t ntot = cs[g].n_paths * (uint64_t)halves;
    if (stop_time) tau[g].resize(ntot);
    if (stop_value) val[g].resize(ntot);
    HH_DHIP(mg, g, hipSetDevice(mg->devices[g]));
    HH_DHIP(mg, g, hipEventRecord(mg->ctx[g]->ev1, mg->ctx[g]->stream));
    return shard_call(g, hh_lsm_shard_finish(mg->ctx[g], mg->xchg[g], stop_time ? tau[g].data() : nullptr,
                                             stop_value ? val[g].data() : nullptr, nullptr, &rg[g], &sk[g]));
  });
  if (rc) return drain(mg, rc);
  for (int g = 0; g < mg->n; ++g) {
    const uint64_t n = cs[g].n_paths;
    for (int h = 0; h < halves; ++h) {
      if (stop_time) std::memcpy(stop_time + (size_t)h * N + starts[g], tau[g].data() + (size_t)h * n, n * sizeof(int32_t));
      if (stop_value) std::memcpy(stop_value + (size_t)h * N + starts[g], val[g].data() + (size_t)h * n, n * sizeof(double));
    }
  }
  if ((rc = lsm_exchange(mg, (size_t)HH_ACC_LEN))) return rc;
  HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[0]));
  HH_MHIP_DRAIN(mg, hipMemcpyAsync(mg->host, mg->xchg[0], HH_ACC_LEN * sizeof(double), hipMemcpyDeviceToHost,
                                   mg->ctx[0]->stream));
  double worst = 0.0;
  for (int g = 0; g < mg->n; ++g) {
    HH_MHIP_DRAIN(mg, hipSetDevice(mg->devices[g]));
    HH_MHIP_DRAIN(mg, hipStreamSynchronize(mg->ctx[g]->stream));
    float ms = 0.f;
    HH_MHIP(mg, hipEventElapsedTime(&ms, mg->ctx[g]->ev0, mg->ctx[g]->ev1));
    if (ms > worst) worst = ms;
  }
  if ((rc = hh_lsm_finalize(mg->host, out))) return mfinalize_failed(mg, rc);
  out->rows_regressed = rg[0];
  out->rows_skipped = sk[0];
  out->form = HH_LSM_FORM_PER_DATE;
  out->kernel_ms = worst;
  out->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return HH_OK;
}

int hh_mgpu_lsm_solve(hh_mgpu* mg, const hh_model* m, const hh_config* cfg, int32_t degree, double step_discount,
                      hh_lsm_result* out, int32_t* stop_time, double* stop_value) {
  if (!mg) return HH_ERR_INVALID;
  std::lock_guard<std::mutex> lock__(mg->mu);
  if (!m || !cfg || !out) return mfail(mg, HH_ERR_INVALID, "hh_mgpu_lsm_solve: NULL argument");
  if (cfg->seeds_on_device || !cfg->seeds || cfg->noise_mode != HH_NOISE_GENERATE)
    return mfail(mg, HH_ERR_INVALID, "hh_mgpu_lsm_solve: GENERATE noise with host seeds");
  if (cfg->n_paths < (uint64_t)mg->n) return mfail(mg, HH_ERR_INVALID, "every device needs at least one trajectory");
  if (cfg->seeds_len && cfg->seeds_len < cfg->n_paths)
    return mfail(mg, HH_ERR_INVALID, "Number of seeds (%llu) must be >= number of trajectories (%llu)",
                 (unsigned long long)cfg->seeds_len, (unsigned long long)cfg->n_paths);
  if (degree < 1 || degree > 8 || cfg->n_steps == 0) return mfail(mg, HH_ERR_INVALID, "LSM: 1 <= degree <= 8, n_steps >= 1");
  if (mg->n == 1) {  // nothing to exchange: the fused induction (one persistent launch where it fits); same bits
    const int rc1 = hh_lsm_solve(mg->ctx[0], m, cfg, degree, step_discount, out, stop_time, stop_value, nullptr);
    return rc1 ? mfail(mg, rc1, "device %d: %s", mg->devices[0], hh_last_error(mg->ctx[0])) : HH_OK;
  }
  if (mg->stuck || (mg->flags == HH_MGPU_RCCL && mg->rccl_lost))
    return mfail(mg, HH_ERR_RCCL, "the RCCL communicators of this context were aborted after a failed collective "
                                  "(HH_MGPU_RCCL does not fall back; a stuck context cannot): create a new one");
  const bool was_rccl = mg->mode == HH_MGPU_REDUCE_RCCL;
  int rc = lsm_attempt(mg, m, cfg, degree, step_discount, out, stop_time, stop_value);
  if (rc == HH_ERR_RCCL && was_rccl) {
    // the exchange is in place, so the local sums went with the failed collective: nothing to finish
    wait_fences(mg);  // the kernels in front of the collective have read the caller's seeds
    if (mg->flags == HH_MGPU_AUTO && !mg->stuck)  // … the whole induction again, on the fresh streams, summed on the host
      rc = lsm_attempt(mg, m, cfg, degree, step_discount, out, stop_time, stop_value);
  }
  return rc;
}

}  // extern "C"
