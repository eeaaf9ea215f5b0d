// Device code shared by the translation units that hold simulation kernels (hh_kernels.hip: one model per
// launch; hh_multi.hip: several models stepped on the same draws): the model policies of one explicit
// Euler–Maruyama step, the payoff, the workgroup reduction that leaves one record per workgroup, and the
// record reduction folded into the kernel that wrote the records.  Internal.
#pragma once
#include <type_traits>

#include "hh_kernels.h"
#include "hh_reduce.h"
#include "hh_rng.h"

namespace hh {

// ------------------------------------------------------------------------------------------
// model policies: one explicit Euler–Maruyama step on the log-state
// ------------------------------------------------------------------------------------------

// heston.jl:7-31.  u = [log S, v];  f = [mu - v+/2, kappa(theta - v+)],  g = [sqrt(v+), sigma sqrt(v+)]
// with v+ = max(v, 0).  K = u + dt f(u);  u' = K + g(.) dW, g taken at K (SPLIT, the integrator's
// split-step form) or at u.
#ifndef HH_LEAN_SQRT
#define HH_LEAN_SQRT 1
#endif
// sqrt of the clipped variance w >= 0.  The library routine is v_rsq_f64 + one coupled Newton step
// + two residual corrections, wrapped in a 2^±256 range scaling for arguments below 2^-767 and a
// class test for 0/inf: 18 instructions, more than half of a Heston path-step.  This is the same
// core sequence (so the same, correctly rounded, result for every w >= 2^-767) with the zero
// handled by ONE v_min_f64 on the seed: rsq(0) = +inf would make w·y = NaN; capped at 2^1000 (every
// rsq of a positive double is below 2^540) the whole sequence is exact zeros for w = 0 — g = 0·2^1000 = 0,
// r = 1/2, both corrections add 0 — and untouched for w > 0.  (Before round 4: compare + two v_cndmask on
// the result; the two instructions are 4 of the antithetic kernel's 47 per pair-step.)  A clipped
// variance between 0 and 2^-767 cannot change any later state.
__device__ __forceinline__ double sqrt_clipped(double w) {
#if HH_LEAN_SQRT
  const double y = __builtin_fmin(__builtin_amdgcn_rsq(w), 0x1p1000);
  double g = w * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, w), h, g);
  return fma(fma(-g, g, w), h, g);
#else
  return sqrt(w);
#endif
}

template <int P, bool SPLIT>
struct HestonModel {
  static constexpr int NCOMP = 2;
  struct State {
    DualT<P> x, v;
  };
  __device__ static __forceinline__ void init(State& s, const SimArgs<P>& a) {
    s.x = a.x0;
    s.v = a.v0;
  }
  __device__ static __forceinline__ void step(State& s, const SimArgs<P>& a, double dW1,
                                              double dW2) {
    const bool pos = s.v.v > 0.0;
    // v+ = max(v, 0) as ONE v_max_f64: written `pos ? v : 0` (or fmax) the compiler first canonicalises the
    // loop-carried v with v_max_f64 v, v, v — an instruction per path-step where the kernel issues one per cycle
    double vp;
    asm("v_max_f64 %0, %1, 0" : "=v"(vp) : "v"(s.v.v));
    const double th_m_v = a.theta.v - vp;
    const double Kx = fma(a.dt, fma(-0.5, vp, a.r.v), s.x.v);  // r - vp/2: the product is exact
    const double Kv = fma(a.dt, a.kappa.v * th_m_v, s.v.v);
    const bool wpos = SPLIT ? (Kv > 0.0) : pos;
    const double w = SPLIT ? (wpos ? Kv : 0.0) : vp;
    const double sq = sqrt_clipped(w);
    if constexpr (P > 0) {
      // d sqrt(w+) = dw / (2 sqrt(w)) for w > 0, and 0 at the clip ("dual rules")
      // 1/(2 sqrt(w)): hardware reciprocal + one Newton step (relative error ~1e-16; an IEEE
      // division would cost ~14 instructions per path-step and the partials do not need it)
      double inv2s = 0.0;
      if (wpos) {
        const double r0 = __builtin_amdgcn_rcp(sq);
        inv2s = 0.5 * fma(r0, fma(-sq, r0, 1.0), r0);
      }
      // The propagation of (dx, dv) is the same linear map for every direction, plus a forcing that
      // differs by the direction's seeds: the map's coefficients are formed ONCE per step
      //   Kvd = m·dv + f_k,        m = 1 - dt κ [v>0],   f_k = dκ_k·dt(θ - v+) + dt κ·dθ_k
      //   dx' = dx + dt·dr_k - h·dv + e1·Kvd,            h = dt/2 [v>0],  e1 = dW1 / (2 sqrt w) [w>0]
      //   dv' = e2·Kvd + dσ_k·(sqrt(w) dW2),             e2 = 1 + σ dW2 / (2 sqrt w) [w>0]
      // (diffusion taken at u instead of K: the sqrt's tangent acts on dv, not on Kvd) — 6-7 fused
      // operations per direction instead of 12; algebraically the step-by-step dual rules above.
      const double A = a.dt * th_m_v, B = sq * dW2;
      const double m = pos ? 1.0 - a.dt * a.kappa.v : 1.0;
      const double h = pos ? 0.5 * a.dt : 0.0;
      const double e1 = inv2s * dW1;
      const double se2 = (a.sigma.v * inv2s) * dW2;
#pragma unroll
      for (int k = 0; k < P; ++k) {
        const double dv = s.v.d[k];
        const double f = fma(a.kappa.d[k], A, (a.dt * a.kappa.v) * a.theta.d[k]);
        const double Kvd = fma(m, dv, f);
        const double x0 = s.x.d[k] + a.dt * a.r.d[k];
        if constexpr (SPLIT) {
          s.x.d[k] = fma(e1, Kvd, fma(-h, dv, x0));
          s.v.d[k] = fma(se2, Kvd, fma(a.sigma.d[k], B, Kvd));
        } else {
          s.x.d[k] = fma(e1 - h, dv, x0);
          s.v.d[k] = fma(se2, dv, fma(a.sigma.d[k], B, Kvd));
        }
      }
    }
    s.x.v = fma(sq, dW1, Kx);
    s.v.v = fma(a.sigma.v * sq, dW2, Kv);
  }
};

// heston.jl:33-52.  x' = x + dt (mu - sigma^2/2) + sigma dW  (g constant, so split is irrelevant)
template <int P>
struct GbmModel {
  static constexpr int NCOMP = 1;
  struct State {
    DualT<P> x;
  };
  __device__ static __forceinline__ void init(State& s, const SimArgs<P>& a) { s.x = a.x0; }
  __device__ static __forceinline__ void step(State& s, const SimArgs<P>& a, double dW, double) {
    if constexpr (P > 0) {
#pragma unroll
      for (int k = 0; k < P; ++k)
        s.x.d[k] = fma(a.sigma.d[k], dW, fma(a.dt, a.gdrift.d[k], s.x.d[k]));
    }
    s.x.v = fma(a.sigma.v, dW, fma(a.dt, a.gdrift.v, s.x.v));
  }
};

// ------------------------------------------------------------------------------------------
// payoff + reduction
// ------------------------------------------------------------------------------------------

// S = exp(x) (montecarlo.jl:398); payoff max(cp (S-K), 0) (payoffs.jl:154-156)
// Partials: only the directions that reach the variance/diffusion ("active", see PartialMap) are
// carried per path; pd[k] = 1[itm]·cp·S·∂x_k.  wS = 1[itm]·cp·S and wN = 1[itm]·cp feed the two sums
// from which every PASSIVE direction (spot, drift rate, strike: ∂x_T is the same constant on every
// path) is finished in closed form by the record reduction.
template <int P>
__device__ __forceinline__ void payoff_of(const DualT<P>& x, const SimArgs<P>& a, double& S,
                                          double& p, double (&pd)[P > 0 ? P : 1], double& wS,
                                          double& wN) {
  S = exp(x.v);
  const double m = a.cp * (S - a.strike.v);
  const bool itm = m > 0.0;
  p = itm ? m : 0.0;
  wS = itm ? a.cp * S : 0.0;
  wN = itm ? a.cp : 0.0;
  if constexpr (P > 0) {
#pragma unroll
    for (int k = 0; k < P; ++k) pd[k] = wS * x.d[k];
  }
}

// wave64 shuffle tree, then across the workgroup's waves through LDS; lane 0 writes the record.
// The last TAIL entries of acc go to record slots kRecItmS, kRecItmS+1 (the in-the-money sums).
template <int N, int NWAVES, int TAIL = 0>
__device__ __forceinline__ void block_reduce_store(double (&acc)[N], double* __restrict__ rec) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_down(acc[i], off, 64);
  }
  __shared__ double sm[NWAVES][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) sm[wave][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kRecStride; ++i) rec[i] = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double t = sm[0][i];
#pragma unroll
      for (int w = 1; w < NWAVES; ++w) t += sm[w][i];
      rec[i < N - TAIL ? i : kRecItmS + (i - (N - TAIL))] = t;
    }
  }
}

template <int P, bool ANTI, class State>
__device__ __forceinline__ void finish_path(const State& st, const State& sa, const SimArgs<P>& a,
                                            uint64_t path, double (&acc)[4 + P]) {
  if (path >= a.n_paths) return;
  double S, p, pd[P > 0 ? P : 1], wS, wN;
  payoff_of<P>(st.x, a, S, p, pd, wS, wN);
  if (a.terminal) a.terminal[path] = S;
  const uint64_t n_total = ANTI ? 2 * a.n_paths : a.n_paths;
  if constexpr (P > 0) {
    if (a.terminal_d) {
#pragma unroll
      for (int k = 0; k < P; ++k) a.terminal_d[(uint64_t)k * n_total + path] = S * st.x.d[k];
    }
  }
  if constexpr (ANTI) {
    double Sa, pa, pda[P > 0 ? P : 1], wSa, wNa;
    payoff_of<P>(sa.x, a, Sa, pa, pda, wSa, wNa);
    if (a.terminal) a.terminal[a.n_paths + path] = Sa;
    if constexpr (P > 0) {
      if (a.terminal_d) {
#pragma unroll
        for (int k = 0; k < P; ++k)
          a.terminal_d[(uint64_t)k * n_total + a.n_paths + path] = Sa * sa.x.d[k];
      }
    }
    p = (p + pa) / 2;  // montecarlo.jl:431
    wS = (wS + wSa) / 2;
    wN = (wN + wNa) / 2;
    if constexpr (P > 0) {
#pragma unroll
      for (int k = 0; k < P; ++k) pd[k] = (pd[k] + pda[k]) / 2;
    }
  }
  acc[0] += p;
  acc[1] = fma(p, p, acc[1]);
  if constexpr (P > 0) {
#pragma unroll
    for (int k = 0; k < P; ++k) acc[2 + k] += pd[k];
  }
  acc[2 + P] += wS;
  acc[3 + P] += wN;
}

// ------------------------------------------------------------------------------------------
// exact lognormal law: the two standard normals of the trajectories path0, path0 + 1
// ------------------------------------------------------------------------------------------
// GENERATE: ONE key for the whole sample (montecarlo.jl:456); trajectory G (global index) takes component G&1
// of Philox block G>>1, so a lane's pair is one block when path_offset is even.  REPLAY: the caller's normals.
template <bool REPLAY, int P>
__device__ __forceinline__ void exact_pair_normals(const SimArgs<P>& a, uint64_t path0, double (&z)[2]) {
  if constexpr (REPLAY) {
    z[0] = path0 < a.n_paths ? a.replay[path0] : 0.0;
    z[1] = path0 + 1 < a.n_paths ? a.replay[path0 + 1] : 0.0;
  } else {
    const uint64_t g0 = a.path_offset + path0;
    const uint64_t key = a.seeds[0];
    double z1, z2;
    normal_pair(key, (uint32_t)(g0 >> 1), (uint32_t)(g0 >> 33), 0u, kDomExactGbm, z1, z2);
    if ((g0 & 1ull) == 0) {
      z[0] = z1;
      z[1] = z2;
    } else {
      z[0] = z2;
      const uint64_t g1 = g0 + 1;
      normal_pair(key, (uint32_t)(g1 >> 1), (uint32_t)(g1 >> 33), 0u, kDomExactGbm, z1, z2);
      z[1] = z1;
    }
  }
}

// ------------------------------------------------------------------------------------------
// the record reduction, folded into the kernel that wrote the records  (mean(payoffs), montecarlo.jl:490)
// ------------------------------------------------------------------------------------------
//
// Until round 4 a second kernel (reduce_records_kernel) added the workgroups' records: 6.5 µs plus a kernel
// boundary behind every solve — 1.8 % of a 10^6 x 252 REPLAY step, 45 % of an exact-law solve.  Now the
// workgroup of the LAST tile, once its own tile is done, adds all records in the fixed order of that kernel —
// virtual thread t the records t, t + 256, … in that order, then a binary tree over the 256 partial sums — so
// every sum has the same bits as before.
//
// How it knows a record is there: the records of this path live in a buffer of their own whose every word
// holds kPoison between launches — a quiet NaN that no arithmetic produces (the hardware's own NaN is
// 0x7FF8000000000000; a payload carried in from a caller's increments is moved off it by publish_value()).
// A workgroup stores its record write-through (sc1: 16 lanes of wave 0, one 128-byte line in ONE store
// instruction) and LEAVES: no counter, no ticket, no fence, nothing to wait for.  The reducer reads with sc1
// loads (the per-XCD L2s are not coherent; MI355X_MICROARCH.md, "inter-workgroup visibility") and simply reads
// again while a word it needs still holds the poison: every 8-byte word validates itself, so no ordering
// between words or workgroups is assumed.  When it has its sums it poisons every record again for the next
// launch.  The last tile's workgroup is the last one dispatched, so as a rule every record is there when it
// looks; if not it waits — it holds one workgroup slot, every other workgroup needs only a slot of its own, so
// the grid drains whatever the dispatch order — and after SimArgs::finish_spin_ticks of the 100 MHz clock (5 s:
// a record that never comes — a workgroup died, or the queue was preempted for that long) it gives up: NaN in
// the accumulator, which hh_mc_finalize refuses, and a 1 in the context's give-up word (SimArgs::finish_state).
// That word is what keeps a give-up from poisoning LATER solves: the record of the straggler, should it still
// come, lands on a buffer the reducer has already poisoned again and would pass for a record of the next launch.
// So every reducer looks at the word first and, while it is set, waits for nothing and leaves NaN itself —
// launches queued behind a give-up fail too, visibly — until the host, which learns of it from the NaN (or from
// hh_ctx_check_last), has filled the buffer with the poison again and cleared the word (recover_finish, hh_api.hip).
//
// Measured alternatives (profiles/r05_a_fuse_ab.txt): a ticket per workgroup drawn with a returning atomic
// ("last one in reduces") holds every workgroup for its store drain and the atomic's round trip — +9 % on the
// REPLAY kernel at two workgroups per CU, and 3907 adds to one address are 45 µs on an 8 µs exact-law kernel;
// a resident grid of workgroups looping over tiles (one ticket each) loses 3.5-4 % to the fixed assignment.

#ifndef HH_FINISH_STAMPS
#define HH_FINISH_STAMPS 0
#endif
// which workgroup reduces: SimArgs::reducer_tile — the last tile's; the FIRST one's under HH_OPT_FINISH_TILE_FIRST
// (a diagnostic: it then has to wait for nearly every record)
template <class Args>
__device__ __forceinline__ bool reduces_records(uint32_t tile, const Args& a) {
  return tile == a.reducer_tile;
}
// The context's give-up word (finish_records says what it is for) by a SCALAR load: issued when the reducer starts,
// waited for where the sums are written.  The word lies on a line nobody else touches — a trip to HBM — and vector
// loads return in order: asked for in front of the record loads it held them all up (+1.7 µs per solve), asked for
// at the kernel's start it cost the REPLAY kernel 1.2 % (profiles/r06_e_headline_ab_vs_round5.txt, r06_g_*).  The
// scalar unit has its own queue and counter.  The scalar cache is invalidated when a kernel starts, and a give-up this
// launch must know of happened in a launch before it on the stream: the value cannot be stale.
__device__ __forceinline__ unsigned int scalar_load_begin(const unsigned int* p) {
  unsigned int v;
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
  return v;  // NOT valid before scalar_load_end()
}
__device__ __forceinline__ unsigned int scalar_load_end(unsigned int v) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) : : "memory");
  return v;
}

__device__ __forceinline__ void store_through(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 … sc1
}
__device__ __forceinline__ double load_through(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dwordx2 … sc1
}
__device__ __forceinline__ bool is_poison(double v) {
  return (unsigned long long)__double_as_longlong(v) == kPoison;
}
// a sum that came out as exactly the poison pattern (a NaN payload carried in by the caller) -> another NaN
__device__ __forceinline__ double publish_value(double v) {
  return is_poison(v) ? __longlong_as_double((long long)(kPoison ^ 1ull)) : v;
}

// block_reduce_store, with the record written by lanes 0..15 of wave 0 (slot = lane) — write-through and
// poison-free when `publish`.  Same adds in the same order per slot: the same record, bit for bit.
// A published record holds only the slots the reducer will read (and poison again): Σp, Σp², the carried
// derivative sums and — `itm_live`: some direction has a passive part — the two in-the-money sums.
template <int N, int NWAVES, int TAIL>
__device__ __forceinline__ void block_reduce_publish(double (&acc)[N], double* __restrict__ rec, bool publish,
                                                     bool itm_live) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_down(acc[i], off, 64);
  }
  __shared__ double sm[NWAVES][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) sm[wave][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < (unsigned)kRecStride) {
    const int slot = threadIdx.x;
    static_assert(N - TAIL <= kRecItmS, "payoff sums and in-the-money sums overlap");
    const int i = slot < N - TAIL ? slot
                  : (slot >= kRecItmS && slot < kRecItmS + TAIL) ? N - TAIL + (slot - kRecItmS) : -1;
    double t = 0.0;
    if (i >= 0) {
      t = sm[0][i];
#pragma unroll
      for (int w = 1; w < NWAVES; ++w) t += sm[w][i];
    }
    if (!publish) rec[slot] = t;
    else if (i >= 0 && (slot < kRecItmS || itm_live)) store_through(rec + slot, publish_value(t));
  }
}

// ---- the reducer ---------------------------------------------------------------------------------------
// Partial sums of the 256 virtual threads of sum_slot() (reduce_records_kernel) — virtual thread vt adds the
// records vt, vt + 256, … in that order — left in LDS, waiting for records that are not there yet.  The tail
// of a solve is this workgroup alone, so everything it needs of a thread's records is requested at once:
// sixteen records per round trip, Σp and Σp² of a record in ONE 16-byte load.
constexpr int kFinishBatch = 16;
using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int;

// true while the caller should read again; after the spin bound: *gave_up is set and the wait is over
__device__ __forceinline__ bool keep_waiting(unsigned long long& t0, bool& give, unsigned int* gave_up,
                                             unsigned long long spin_ticks, const unsigned int* state) {
  const unsigned long long now = wall_clock64();
  if (t0 == 0) t0 = now;
  if (now - t0 >= spin_ticks) {  // a record that never comes: NaN sums, and say so
    *gave_up = 1u;
    give = true;
    return false;
  }
  // a wait that has lasted 20 µs (the last record of a healthy launch is 7 µs away at most): is this a buffer to wait
  // on at all?  (Not asked at the first miss: the word is a trip to HBM, and the next look at the records would queue
  // behind it — +1.5 µs on every REPLAY solve, profiles/r06_h_headline_ab_vs_round5.txt)
  if (now - t0 > 2000ull && __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    give = true;
    return false;
  }
  __builtin_amdgcn_s_sleep(8);
  return true;
}

// slots 0 and 1 of every record -> sm[vt], sm[256 + vt]
template <int NT>
__device__ __forceinline__ void partial_pairs(const double* rec, uint32_t n, double* __restrict__ sm,
                                              unsigned int* gave_up, unsigned long long spin_ticks, const unsigned int* state) {
  static_assert(256 % NT == 0, "the 256 virtual threads are dealt evenly");
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(rec), 0, (int)(n * (uint32_t)(kRecStride * 8)), 0x00020000);
  bool give = *gave_up != 0u;  // set in an earlier pass: do not wait again
  for (int vt = threadIdx.x; vt < 256; vt += NT) {
    double t0 = 0.0, t1 = 0.0;
    for (uint32_t b = vt; b < n; b += 256 * kFinishBatch) {
      u32x4 v[kFinishBatch];
      unsigned long long since = 0;
      for (;;) {
        bool there = true;
#pragma unroll
        for (int u = 0; u < kFinishBatch; ++u) {
          const uint32_t i = b + 256u * u;
          v[u] = u32x4{0u, 0u, 0u, 0u};
          if (i < n) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(i * (uint32_t)(kRecStride * 8)), 0, 16);  // sc1
          const unsigned long long lo = ((unsigned long long)v[u][1] << 32) | v[u][0];
          const unsigned long long hi = ((unsigned long long)v[u][3] << 32) | v[u][2];
          there = there && lo != kPoison && hi != kPoison;
        }
        if (there || give || !keep_waiting(since, give, gave_up, spin_ticks, state)) break;
      }
#pragma unroll
      for (int u = 0; u < kFinishBatch; ++u) {
        t0 += __longlong_as_double((long long)(((unsigned long long)v[u][1] << 32) | v[u][0]));
        t1 += __longlong_as_double((long long)(((unsigned long long)v[u][3] << 32) | v[u][2]));
      }
    }
    sm[vt] = t0;
    sm[256 + vt] = t1;
  }
}

// NS other slots of every record -> sm[q·256 + vt]
template <int NT, int NS, class SlotOf>
__device__ __forceinline__ void partial_slots(const double* __restrict__ rec, uint32_t n, SlotOf slot_of,
                                              double* __restrict__ sm, unsigned int* gave_up, unsigned long long spin_ticks,
                                              const unsigned int* state) {
  constexpr int kB = NS <= 2 ? kFinishBatch : 8;
  bool give = *gave_up != 0u;
  for (int vt = threadIdx.x; vt < 256; vt += NT) {
    double t[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) t[q] = 0.0;
    for (uint32_t b = vt; b < n; b += 256 * kB) {
      double v[NS][kB];
      unsigned long long since = 0;
      for (;;) {
        bool there = true;
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const uint32_t i = b + 256u * u;
#pragma unroll
          for (int q = 0; q < NS; ++q) {
            v[q][u] = i < n ? load_through(rec + (size_t)i * kRecStride + slot_of(q)) : 0.0;
            there = there && !is_poison(v[q][u]);
          }
        }
        if (there || give || !keep_waiting(since, give, gave_up, spin_ticks, state)) break;
      }
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int u = 0; u < kB; ++u) t[q] += v[q][u];
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) sm[q * 256 + vt] = t[q];
  }
}

// Called by every thread of the reducing workgroup (reduces_records) after block_reduce_publish(…, true).
// NT = threads of the workgroup.
template <int NT, int P, class Args>
__device__ __forceinline__ void finish_records(const Args& a) {
  double* __restrict__ records = a.records;
  double* __restrict__ accum = a.accum;
  const uint32_t n_rec = a.n_tiles;
  const double n_paths = a.acc_n_paths;
  const PartialMap* map = &a.map;
  const unsigned long long spin_ticks = a.finish_spin_ticks;
  __shared__ double sm[4 * 256];
  __shared__ unsigned int gave_up;
#if HH_FINISH_STAMPS  // a diagnostic build (tools/finish_stamps.py): where the tail's time goes, in accum[11..15]
  const unsigned long long st0 = wall_clock64();
#endif
  // (a give-up of an earlier launch the host has not dealt with yet: this buffer cannot be trusted — the sums are NaN
  // whatever it holds, and a wait for a record that looks missing ends at once: keep_waiting asks too)
  const unsigned int state_in_flight = scalar_load_begin(a.finish_state);
  if (threadIdx.x == 0) gave_up = 0u;
  __syncthreads();
  const bool itm = map->n > 0;  // some direction has a passive part: the two in-the-money sums are live
  // Every published word goes back to the poison as soon as the thread that read it has it in its partial sum
  // (records i = tid mod NT are the ones of its virtual threads): the write-through stores then travel while
  // the sums are finished, instead of holding the end of the kernel up.  A word that never came is poison already.
  const double poison = __longlong_as_double((long long)kPoison);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc(records, 0, (int)(n_rec * (uint32_t)(kRecStride * 8)), 0x00020000);
  const unsigned int ph = (unsigned int)(kPoison >> 32), pl = (unsigned int)kPoison;
  partial_pairs<NT>(records, n_rec, sm, &gave_up, spin_ticks, a.finish_state);
  if (itm) partial_slots<NT, 2>(records, n_rec, [](int q) { return kRecItmS + q; }, sm + 512, &gave_up, spin_ticks, a.finish_state);
  for (uint32_t i = threadIdx.x; i < n_rec; i += NT) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{pl, ph, pl, ph}, wsrc, (int)(i * (uint32_t)(kRecStride * 8)), 0, 16);  // Σp, Σp²: sc1
    if (itm) {
      store_through(records + (size_t)i * kRecStride + kRecItmS, poison);
      store_through(records + (size_t)i * kRecStride + kRecItmS + 1, poison);
    }
  }
#if HH_FINISH_STAMPS
  const unsigned long long st1 = wall_clock64();
#endif
  __syncthreads();
  double s[4] = {0.0, 0.0, 0.0, 0.0}, d[P > 0 ? P : 1];
  if (threadIdx.x < 64) {
    s[0] = tree256(sm);
    s[1] = tree256(sm + 256);
    if (itm) {
      s[2] = tree256(sm + 512);
      s[3] = tree256(sm + 768);
    }
  }
  if constexpr (P > 0) {  // the carried derivative sums
    __syncthreads();
    partial_slots<NT, P>(records, n_rec, [](int q) { return HH_ACC_DSUM + q; }, sm, &gave_up, spin_ticks, a.finish_state);
    for (uint32_t i = threadIdx.x; i < n_rec; i += NT) {
#pragma unroll
      for (int j = 0; j < P; ++j) store_through(records + (size_t)i * kRecStride + HH_ACC_DSUM + j, poison);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
#pragma unroll
      for (int j = 0; j < P; ++j) d[j] = tree256(sm + j * 256);
    }
  }
  if (threadIdx.x < (unsigned)kRecStride) {  // lane `slot` of wave 0 writes accum[slot]
    const int slot = threadIdx.x;
    double out = slot == HH_ACC_SUM ? s[0] : slot == HH_ACC_SUMSQ ? s[1] : 0.0;
    // requested direction k from the carried ones and the passive sums: reduce_records_kernel's arithmetic
    const int k = slot - HH_ACC_DSUM;
    if (k >= 0 && k < map->n) {
      out = 0.0;
      if constexpr (P > 0) {
#pragma unroll
        for (int j = 0; j < P; ++j)
          if (j < map->n_active && map->w[k][j] != 0.0) out = fma(map->w[k][j], d[j], out);
      }
      if (map->xdT[k] != 0.0) out = fma(map->xdT[k], s[2], out);
      if (map->dK[k] != 0.0) out -= map->dK[k] * s[3];
    }
    if (slot == HH_ACC_NPATHS) out = n_paths;
#if HH_FINISH_STAMPS
    if (slot == 11) out = (double)st0;               // this workgroup's own record is out
    if (slot == 12) out = (double)st1;               // thread `slot` has all its records
    if (slot == 13) out = (double)wall_clock64();    // sums done
#endif
    const bool dirty = scalar_load_end(state_in_flight) != 0u;
    accum[slot] = (gave_up || dirty) ? __longlong_as_double(0x7FF8000000000000ll) : out;
    if (slot == 0 && gave_up) __hip_atomic_store(a.finish_state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------
// REPLAY stream loads
// ------------------------------------------------------------------------------------------

template <int PPT>
struct VecOf;
template <>
struct VecOf<1> {
  using type = double;
  __device__ static __forceinline__ double get(const type& v, int) { return v; }
};
template <>
struct VecOf<2> {
  using type = double __attribute__((ext_vector_type(2)));
  __device__ static __forceinline__ double get(const type& v, int j) { return j ? v.y : v.x; }
};

#ifndef HH_REPLAY_NT
#define HH_REPLAY_NT 1            // the increments are a read-once stream: nontemporal loads
#endif
template <class Vec>
__device__ __forceinline__ Vec stream_load(const double* p) {
#if HH_REPLAY_NT
  return __builtin_nontemporal_load(reinterpret_cast<const Vec*>(p));  // read-once stream
#else
  return *reinterpret_cast<const Vec*>(p);
#endif
}

}  // namespace hh
