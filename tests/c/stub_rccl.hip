// TEST-ONLY stand-in for librccl (tests/test_gpu_mgpu_rccl_stub.py; never part of the product).
//
// A one-GPU test box cannot run hh_mgpu's RCCL branch with more than one rank: the real
// ncclCommInitAll refuses a device that is listed twice.  This library exports the seven RCCL entry
// points hh_mgpu.hip binds with dlsym ($HEDGEHOG_MC_RCCL points at it) and carries them out with plain
// HIP on the caller's streams, so that G "ranks" may share a device:
//
//   * a complete group (every rank of the communicator enqueued): an event per rank stream, a fold
//     kernel on rank 0's stream that adds the send vectors IN RANK ORDER ((v0 + v1) + v2 …: the same
//     doubles as the library's host ordered sum, so results can be compared bit for bit), copies to
//     every receive vector, and an event the other rank streams wait for — asynchronous, like RCCL;
//   * a group that stays INCOMPLETE because ncclAllReduce was told to fail for one rank
//     (stub_rccl_fail_at): every rank that did enqueue gets what real RCCL leaves behind — a kernel on
//     its stream that waits for peers that never come.  It leaves when the communicator is aborted
//     (ncclCommAbort sets a flag in host memory the kernel polls) or, so that a bug in the library
//     under test cannot wedge the GPU box, after three seconds.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <mutex>
#include <vector>

namespace {

constexpr int kSuccess = 0, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5;
constexpr int kSum = 0, kDouble = 8;
constexpr int kMaxRanks = 64;
constexpr size_t kTmpDoubles = 1u << 20;
constexpr unsigned long long kOrphanTicks = 300000000ull;  // 3 s of the 100 MHz constant clock

struct Shared {
  int n = 0;
  int live = 0;
  unsigned* abort_flag = nullptr;  // pinned, coherent; never freed (an orphan may still poll it)
  double* tmp = nullptr;           // on rank 0's device
  hipEvent_t done = nullptr;
};

}  // namespace

struct ncclComm {
  int rank = 0, n = 0, device = 0;
  Shared* sh = nullptr;
  hipEvent_t local = nullptr;
};

namespace {

struct Op {
  const double* send;
  double* recv;
  size_t count;
  ncclComm* comm;
  hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

std::mutex g_mu;
int g_fail_rank = -1, g_fail_countdown = 0;
std::atomic<int> g_calls{0}, g_complete{0}, g_orphans{0}, g_aborts{0};

struct Ptrs {
  const double* p[kMaxRanks];
};

__global__ void fold_kernel(Ptrs src, int n, size_t count, double* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = src.p[0][i];
  for (int r = 1; r < n; ++r) s = s + src.p[r][i];  // rank order, as the host ordered sum
  out[i] = s;
}

__global__ void orphan_kernel(const unsigned* abort_flag, unsigned long long max_ticks) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u &&
         wall_clock64() - t0 < max_ticks)
    __builtin_amdgcn_s_sleep(64);
}

int run_complete(std::vector<Op>& ops) {  // ops sorted by rank, one per rank of the communicator
  Shared* sh = ops[0].comm->sh;
  const int n = sh->n;
  const size_t count = ops[0].count;
  if (count > kTmpDoubles) return kInvalidArgument;
  for (const Op& o : ops)
    if (o.count != count) return kInvalidArgument;
  Ptrs src{};
  for (int r = 0; r < n; ++r) {
    src.p[r] = ops[r].send;
    if (hipSetDevice(ops[r].comm->device) != hipSuccess) return kInternalError;
    if (hipEventRecord(ops[r].comm->local, ops[r].stream) != hipSuccess) return kInternalError;
  }
  hipStream_t s0 = ops[0].stream;
  if (hipSetDevice(ops[0].comm->device) != hipSuccess) return kInternalError;
  for (int r = 1; r < n; ++r)
    if (hipStreamWaitEvent(s0, ops[r].comm->local, 0) != hipSuccess) return kInternalError;
  if (count) {
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s0, src, n, count, sh->tmp);
    if (hipGetLastError() != hipSuccess) return kInternalError;
    for (int r = 0; r < n; ++r)
      if (hipMemcpyAsync(ops[r].recv, sh->tmp, count * sizeof(double), hipMemcpyDeviceToDevice, s0) != hipSuccess)
        return kInternalError;
  }
  if (hipEventRecord(sh->done, s0) != hipSuccess) return kInternalError;
  for (int r = 1; r < n; ++r) {
    if (hipSetDevice(ops[r].comm->device) != hipSuccess) return kInternalError;
    if (hipStreamWaitEvent(ops[r].stream, sh->done, 0) != hipSuccess) return kInternalError;
  }
  ++g_complete;
  return kSuccess;
}

int flush_group() {
  std::vector<Op> ops;
  ops.swap(g_ops);
  int rc = kSuccess;
  while (!ops.empty()) {
    Shared* sh = ops[0].comm->sh;
    std::vector<Op> mine(sh->n, Op{nullptr, nullptr, 0, nullptr, nullptr});
    std::vector<Op> rest;
    int have = 0;
    for (const Op& o : ops) {
      if (o.comm->sh != sh) {
        rest.push_back(o);
      } else if (!mine[o.comm->rank].comm) {
        mine[o.comm->rank] = o;
        ++have;
      } else {
        return kInvalidUsage;  // one collective per rank and group is all this stand-in does
      }
    }
    if (have == sh->n) {
      const int e = run_complete(mine);
      if (e != kSuccess) rc = e;
    } else {  // what RCCL leaves behind: the enqueued ranks wait for the missing ones
      for (const Op& o : mine) {
        if (!o.comm) continue;
        (void)hipSetDevice(o.comm->device);
        hipLaunchKernelGGL(orphan_kernel, dim3(1), dim3(1), 0, o.stream, sh->abort_flag, kOrphanTicks);
        ++g_orphans;
      }
    }
    ops.swap(rest);
  }
  return rc;
}

void release(ncclComm* c) {
  Shared* sh = c->sh;
  (void)hipSetDevice(c->device);
  if (c->local) (void)hipEventDestroy(c->local);
  std::lock_guard<std::mutex> lk(g_mu);
  if (--sh->live == 0) {
    // the owner synchronises its streams before ncclCommDestroy (hh_mgpu_destroy does), so nothing queued
    // still uses these; after an ABORT they are leaked instead (ncclCommAbort below)
    (void)hipFree(sh->tmp);
    (void)hipEventDestroy(sh->done);
    (void)hipHostFree(sh->abort_flag);
    delete sh;
  }
  delete c;
}

}  // namespace

extern "C" {

int ncclCommInitAll(ncclComm** comms, int n, const int* devs) {
  if (!comms || n < 1 || n > kMaxRanks) return kInvalidArgument;
  Shared* sh = new Shared();
  sh->n = sh->live = n;
  const int dev0 = devs ? devs[0] : 0;
  if (hipSetDevice(dev0) != hipSuccess ||
      hipHostMalloc((void**)&sh->abort_flag, sizeof(unsigned), hipHostMallocCoherent) != hipSuccess ||
      hipMalloc((void**)&sh->tmp, kTmpDoubles * sizeof(double)) != hipSuccess ||
      hipEventCreateWithFlags(&sh->done, hipEventDisableTiming) != hipSuccess) {
    delete sh;
    return kInternalError;
  }
  *sh->abort_flag = 0u;
  for (int r = 0; r < n; ++r) {
    ncclComm* c = new ncclComm();
    c->rank = r;
    c->n = n;
    c->device = devs ? devs[r] : r;  // a device may be listed twice: that is the point
    c->sh = sh;
    if (hipSetDevice(c->device) != hipSuccess ||
        hipEventCreateWithFlags(&c->local, hipEventDisableTiming) != hipSuccess)
      return kInternalError;
    comms[r] = c;
  }
  return kSuccess;
}

int ncclCommDestroy(ncclComm* c) {
  if (!c) return kInvalidArgument;
  release(c);
  return kSuccess;
}

int ncclCommAbort(ncclComm* c) {
  if (!c) return kInvalidArgument;
  __atomic_store_n(c->sh->abort_flag, 1u, __ATOMIC_SEQ_CST);  // the orphans leave
  ++g_aborts;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    --c->sh->live;  // Shared (tmp, flag, event) is leaked on purpose: queued kernels may still read it
  }
  delete c;
  return kSuccess;
}

// a version no real RCCL reports: a line that names this library cannot be taken for one from the real thing
int ncclGetVersion(int* v) {
  if (!v) return kInvalidArgument;
  *v = 9990000;
  return kSuccess;
}

int ncclGroupStart() {
  ++g_depth;
  return kSuccess;
}

int ncclGroupEnd() {
  if (g_depth <= 0) return kInvalidUsage;
  if (--g_depth > 0) return kSuccess;
  return flush_group();
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, ncclComm* comm, hipStream_t stream) {
  ++g_calls;
  if (!comm || (count && (!send || !recv))) return kInvalidArgument;
  if (dtype != kDouble || op != kSum) return kInvalidArgument;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_fail_rank == comm->rank && --g_fail_countdown == 0) {
      g_fail_rank = -1;
      return kInternalError;
    }
  }
  g_ops.push_back(Op{(const double*)send, (double*)recv, count, comm, stream});
  if (g_depth == 0) return comm->n == 1 ? flush_group() : kInvalidUsage;
  return kSuccess;
}

const char* ncclGetErrorString(int e) {
  switch (e) {
    case kSuccess: return "no error";
    case kInternalError: return "internal error (stub_rccl: injected or HIP failure)";
    case kInvalidArgument: return "invalid argument";
    case kInvalidUsage: return "invalid usage";
    default: return "unknown result code";
  }
}

// ---- controls of the stand-in (bound by the test through ctypes on the same file) ----
// the nth ncclAllReduce call made for `rank` from now on returns ncclInternalError, once
void stub_rccl_fail_at(int rank, int nth) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_fail_rank = nth > 0 ? rank : -1;
  g_fail_countdown = nth;
}
// ncclAllReduce calls, complete groups carried out, orphan kernels left behind, communicators aborted
void stub_rccl_counters(int out[4]) {
  out[0] = g_calls.load();
  out[1] = g_complete.load();
  out[2] = g_orphans.load();
  out[3] = g_aborts.load();
}

}  // extern "C"
