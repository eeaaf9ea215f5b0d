// Longstaff–Schwartz American pricing on the full path grid — the consumer of simulate_paths that
// needs every time step.  Reference: solve(::PricingProblem{VanillaOption{…,
// American,…}}, ::LSM), src/pricing_methods/least_squares_montecarlo.jl:99-165, on the paths of
// sde_problem(::LognormalDynamics, ::BlackScholesExact) (montecarlo.jl:140-159, antithetic :270-284).
//
// Layout: spot grid S[step][path] (step-major: every backward step streams contiguous rows),
// per-path stopping state tau[path] (int32), val[path] (fp64).
//
// The backward induction is serial in time and data-parallel over paths.  What couples the paths
// is, per exercise date t, a handful of sums over ALL of them: the in-the-money statistics
// (n, Σx, Σx²) that standardise the regressor z = (x - mean)/std, the power sums Σ z^k (the Gram
// matrix of Polynomials.fit(x, y, degree), :126, in the z basis — same polynomial space, well
// conditioned in fp64) and the moment sums Σ z^k y.  Two forms compute them, with the SAME
// summation tree (so their coefficients, stopping decisions and prices are bit-identical):
//
//  * ONE PERSISTENT (cooperative) LAUNCH (lsm_persistent_kernel; every ensemble of up to 256 chunks =
//    2^21 trajectories): every workgroup keeps the stopping state and the current row of its
//    trajectories in registers for the whole induction (512 threads x 2 … 16 trajectories, up to 256
//    registers per lane; rows t-1 and t-2 wait in LDS) and reads every row of the grid exactly once.
//    Per date the workgroups publish one record in two halves — A: Σ z^k y of row t-1 and the
//    statistics of row t-3, B: Σ z^k of row t-2 (the pipeline runs statistics three rows, power sums
//    two, moment sums one row ahead of the decisions) — and gather everybody's: every value travels as
//    a self-validating 16-byte granule {v, v ^ (launch nonce : epoch)} written through (sc1), read
//    first through the caches and again with sc0 sc1 loads only if it does not validate (records
//    value-major so that a gathering wave reads contiguous lines).  Each workgroup then reduces the
//    records in the fixed order, solves the normal equations redundantly (one wave, rows spread over
//    its lanes) and takes the exercise decisions of its own trajectories.  No launch, no re-load of
//    tau / val / spots, no separate passes for the statistics and the power sums.  Every wait is
//    bounded (s_memrealtime): should the grid not be co-resident after all the kernel gives up, the
//    host sees a status word and runs the other form.
//  * ONE LAUNCH PER DATE (lsm_step_kernel & co.): statistics and power sums of every row in one-off
//    launches, then a launch per exercise date.  Used beyond 2^21 trajectories, when the runtime
//    refuses the cooperative grid, and — cut at the global sums — for ensembles sharded over several
//    GPUs (launch_lsm_phase), where the sums are all-reduced between launches.
//  Measured, 2·10^6 trajectories x 100 dates, degree 5 (profiles/r03_g_lsm_*, r03_m_*): 1.44-1.60 ms
//  in one launch by box (10.3 µs per date: arithmetic of both waves of a SIMD 7.4, solve window 1.35,
//  gather 1.1) against 2.9 ms with a launch per date; round 2 2.05 ms, round 1 4.3 ms.
//
// Summation tree (independent of the form and of the GPU): chunk = 512 lanes x Q trajectories
// (trajectory = chunk·512·Q + j·512 + lane; Q = the smallest of 2, 4, 8, 16 that fits 256 chunks: lsm_q); a lane adds
// its Q terms in order j (out-of-the-money terms are exact zeros); a wave adds its 64 lanes by the
// butterfly (l, l^32), (l, l^16), …; the chunk adds its 8 waves in order; the records of the chunks
// are dealt to 256 lanes (r, r+256, …, added in order) which are summed by the same butterfly and,
// over their 4 waves, in order.
#include <cmath>
#include <type_traits>

#include "hh_kernels.h"
#include "hh_math.h"
#include "hh_rng.h"
#include <atomic>

namespace hh {

namespace {

constexpr int kLsmFinalChunk = 1024;  // paths per workgroup of the final Σ, Σ² kernel (16-double records)
#ifndef HH_LSM_WG
#define HH_LSM_WG 512
#endif
constexpr int kLsmWg = HH_LSM_WG;     // threads per workgroup of every kernel that forms canonical sums
constexpr int kLsmQSmall = 1024 / kLsmWg, kLsmQLarge = 8192 / kLsmWg;  // trajectories per lane
constexpr int kLsmWaves = kLsmWg / 64;
constexpr int kLsmMaxDeg = 8;
constexpr uint64_t kLsmQ1Max = 1ull << 18;  // up to here chunks of 1024 trajectories, beyond it 8192
constexpr int kLsmMaxResident = 256;        // chunks the persistent form handles (one per workgroup)
constexpr int kLsmRing = 16;                // record slots of the persistent all-gather (2 suffice for
                                            // correctness; 16 dates between two uses of a slot make it all
                                            // but certain that no L2 still holds the slot's previous lines)

// Trajectories per lane: the smallest of 2, 4, 8, 16 (x 512 lanes = chunks of 1024 … 8192) with which
// the ensemble fits 256 chunks — one workgroup per CU in the persistent form, whose date is bounded by
// what ONE workgroup has to do (a 10^6-trajectory induction in chunks of 8192 would keep half the chip
// idle and every busy CU twice as long).  The chunk size is part of the summation tree: both forms of
// the induction (and every phase of the sharded one) use lsm_q() of the same ensemble.
inline int lsm_q(uint64_t ntot) {
  int q = kLsmQSmall;
  for (uint64_t cap = kLsmQ1Max; q < kLsmQLarge && ntot > cap; cap <<= 1) q <<= 1;
  return q;
}
// f(std::integral_constant<int, Q>) for the Q that lsm_q() returned
template <class F>
inline auto with_q(int q, F&& f) {
  static_assert(kLsmQLarge == 8 * kLsmQSmall, "four chunk sizes");
  if (q == kLsmQSmall) return f(std::integral_constant<int, kLsmQSmall>{});
  if (q == 2 * kLsmQSmall) return f(std::integral_constant<int, 2 * kLsmQSmall>{});
  if (q == 4 * kLsmQSmall) return f(std::integral_constant<int, 4 * kLsmQSmall>{});
  return f(std::integral_constant<int, kLsmQLarge>{});
}
inline uint32_t lsm_nch(uint64_t ntot) {
  const uint64_t per = (uint64_t)kLsmWg * lsm_q(ntot);
  return (uint32_t)((ntot + per - 1) / per);
}

// ---- full path grid -----------------------------------------------------------------------

#ifndef HH_LSM_GRID_NT
#define HH_LSM_GRID_NT 1  // the grid is written once and read once, 1.6 GB: nontemporal stores (-2 % of the LSM chain)
#endif

#if HH_LSM_GRID_NT
#define HH_GRID_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define HH_GRID_STORE(p, v) (*(p) = (v))
#endif
template <bool ANTI>
__global__ __launch_bounds__(256) void gbm_grid_kernel(const uint64_t* __restrict__ seeds,
                                                       uint64_t n_paths, uint32_t n_steps,
                                                       double S0, double a, double b, double e2a,
                                                       double* __restrict__ grid) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_paths) return;
  const uint64_t ntot = ANTI ? 2 * n_paths : n_paths;
  const uint64_t key = seeds[i];  // montecarlo.jl:331
  double S = S0, Sa = S0;
  grid[i] = S0;
  if (ANTI) grid[n_paths + i] = S0;
  for (uint32_t s = 0; s < n_steps; s += 2) {
    double z[2];
    normal_pair(key, s >> 1, 0u, 0u, kDomEuler, z[0], z[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (s + h < n_steps) {
        // GBM process increment dW = W (exp((μ-σ²/2) dt + σ √dt z) - 1)
        const double e = fm::exp(fma(b, z[h], a));  // hh_math.h: <= 1.5 ulp, a third of the library's instructions
        S = S + S * (e - 1.0);
        HH_GRID_STORE(&grid[(size_t)(s + h + 1) * ntot + i], S);
        if (ANTI) {  // flipped σ, same draws (montecarlo.jl:276): exp(a - b z) = exp(2a) / exp(a + b z),
                     // a reciprocal (6 instructions, <= 3.5 ulp) instead of a second exponential (25)
          Sa = Sa + Sa * (e2a * fm::rcp(e) - 1.0);
          HH_GRID_STORE(&grid[(size_t)(s + h + 1) * ntot + n_paths + i], Sa);
        }
      }
    }
  }
}

// ---- the canonical reductions ----------------------------------------------------------------

constexpr int pow2_ge(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : n <= 16 ? 16 : n <= 32 ? 32 : 64; }
constexpr int log2_of(int p) { return p <= 1 ? 0 : 1 + log2_of(p / 2); }

// 64 lanes x P2 values -> P2 totals, each the butterfly tree ((l, l^32), (l, l^16), …, (l, l^1)) over
// the lanes.  While more than one value is left per lane the exchange also TRANSPOSES: the lower
// lane of a pair keeps the first half of the values, the upper lane the second half, so a step
// moves half as many values as the one before (P2 + log2(64/P2) exchanges instead of 6·P2).  The
// tree — and so every bit of the totals — is the same for every P2.  On return the total of value i
// is a[0] of the lanes with (lane >> (6 - log2 P2)) == i.
// x of lane (l ^ OFF), without the LDS round trip of __shfl_xor (ds_bpermute_b32, ~100 cycles each, and
// three 16-value butterflies per date sit on the induction's critical path): DPP moves inside a row of
// 16 lanes (xor 1, 2: quad_perm; xor 4: row_shl / row_shr by 4 into alternate banks; xor 8: row_ror:8),
// gfx950's v_permlane16_swap / v_permlane32_swap across rows.  Same partner, same value: the sums do
// not change by a bit.
#ifndef HH_LSM_DPP
#define HH_LSM_DPP 1
#endif
#ifndef HH_LSM_SWAP_PAIRS  // A/B switches of round 3's later cuts (tools/lsm_breakdown.py; both bit-identical)
#define HH_LSM_SWAP_PAIRS 1
#endif
#ifndef HH_LSM_EARLY_STATS
#define HH_LSM_EARLY_STATS 1
#endif
#ifndef HH_LSM_GATHER_USED
#define HH_LSM_GATHER_USED 1
#endif

template <int OFF>
__device__ __forceinline__ unsigned xor_lane_u32(unsigned x, int lane) {
  if constexpr (OFF == 1) return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false);
  else if constexpr (OFF == 2) return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false);
  else if constexpr (OFF == 4) {
    int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);  // lanes 0-3, 8-11 of a row: l + 4
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false);  // lanes 4-7, 12-15: l - 4
  } else if constexpr (OFF == 8) return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xF, 0xF, false);
  else if constexpr (OFF == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // odd rows of [0] <-> even rows of [1]
    return (lane & 16) ? r[0] : r[1];
  } else {
    static_assert(OFF == 32, "xor offsets of a wave64 butterfly");
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // upper half of [0] <-> lower half of [1]
    return (lane & 32) ? r[0] : r[1];
  }
}
template <int OFF>
__device__ __forceinline__ double xor_lane(double x, int lane) {
#if HH_LSM_DPP
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = xor_lane_u32<OFF>((unsigned)b, lane), hi = xor_lane_u32<OFF>((unsigned)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
#else
  return __shfl_xor(x, OFF, 64);
#endif
}

template <int P2>
__device__ __forceinline__ void wave_reduce_multi(double (&a)[P2]) {
  const int lane = threadIdx.x & 63;
  int cnt = P2;
  auto step = [&](auto off_c) {
    constexpr int off = decltype(off_c)::value;
    if (cnt > 1) {
      const int h = cnt / 2;
      const bool upper = (lane & off) != 0;
#pragma unroll
      for (int i = 0; i < P2 / 2; ++i) {
        if (i < h) {
#if HH_LSM_DPP && HH_LSM_SWAP_PAIRS
          if constexpr (off == 32 || off == 16) {
            // v_permlane{32,16}_swap IS the exchange of this step: it swaps a[i] of the upper lanes with
            // a[i + h] of their partners, after which every lane holds {what it keeps, what it was sent}
            // in the two registers — no select on either side.  (Upper lanes add recv + keep instead of
            // keep + recv: the same sum.)
            const unsigned long long x = (unsigned long long)__double_as_longlong(a[i]);
            const unsigned long long y = (unsigned long long)__double_as_longlong(a[i + h]);
            unsigned xl, yl, xh, yh;
            if constexpr (off == 32) {
              const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)y, false, false);
              const auto rh = __builtin_amdgcn_permlane32_swap((unsigned)(x >> 32), (unsigned)(y >> 32), false, false);
              xl = rl[0], yl = rl[1], xh = rh[0], yh = rh[1];
            } else {
              const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)y, false, false);
              const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)(x >> 32), (unsigned)(y >> 32), false, false);
              xl = rl[0], yl = rl[1], xh = rh[0], yh = rh[1];
            }
            a[i] = __longlong_as_double((long long)(((unsigned long long)xh << 32) | xl)) +
                   __longlong_as_double((long long)(((unsigned long long)yh << 32) | yl));
            continue;
          }
#endif
          // both elements into registers first: written as `upper ? a[i] : a[i + h]` the compiler
          // selects the ADDRESS, which makes `a` a dynamically indexed array in scratch memory
          const double lo = a[i], hi = a[i + h];
          const double send = upper ? lo : hi;
          const double keep = upper ? hi : lo;
          a[i] = keep + xor_lane<off>(send, lane);
        }
      }
      cnt = h;
    } else {
      a[0] += xor_lane<off>(a[0], lane);
    }
  };
  step(std::integral_constant<int, 32>{});
  step(std::integral_constant<int, 16>{});
  step(std::integral_constant<int, 8>{});
  step(std::integral_constant<int, 4>{});
  step(std::integral_constant<int, 2>{});
  step(std::integral_constant<int, 1>{});
}

// NW waves x 64 lanes x P2 values -> tot[P2] in LDS (valid for every thread after the call):
// wave butterflies, then the waves in order 0, 1, …  `scratch` holds NW·P2 doubles.  All threads of
// the workgroup must call it (waves >= NW only take part in the barriers).
template <int P2, int NW>
__device__ __forceinline__ void block_reduce_multi(double (&a)[P2], double* scratch, double* tot) {
  constexpr int kShift = 6 - log2_of(P2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < NW) {
    wave_reduce_multi<P2>(a);
    if ((lane & ((1 << kShift) - 1)) == 0) scratch[wave * P2 + (lane >> kShift)] = a[0];
  }
  __syncthreads();
  if (threadIdx.x < P2) {
    double t = scratch[threadIdx.x];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += scratch[w * P2 + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
}

// The same reduction in two pieces, so that several groups of values share ONE pair of barriers:
// wave_part() per group (butterfly, totals of this wave into scratch[wave][off ..]), then
// finish_block<TOT>() once.  scratch holds NW·TOT doubles.  Bit-identical to block_reduce_multi.
template <int P2, int TOT>
__device__ __forceinline__ void wave_part(double (&a)[P2], double* scratch, int off) {
  constexpr int kShift = 6 - log2_of(P2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_reduce_multi<P2>(a);
  if ((lane & ((1 << kShift) - 1)) == 0) scratch[wave * TOT + off + (lane >> kShift)] = a[0];
}
template <int TOT, int NW>
__device__ __forceinline__ void finish_block(const double* scratch, double* tot) {
  __syncthreads();
  if (threadIdx.x < TOT) {
    double t = scratch[threadIdx.x];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += scratch[w * TOT + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
}

// records of the chunks, rec[r·stride + i] (i < NV), dealt to lanes 0..255 and summed: tot[NV]
template <int NV, int P2>
__device__ __forceinline__ void reduce_chunk_records(const double* __restrict__ rec, uint32_t n_rec,
                                                     int stride, double* scratch, double* tot) {
  double a[P2];
#pragma unroll
  for (int i = 0; i < P2; ++i) a[i] = 0.0;
  if (threadIdx.x < 256) {
    for (uint32_t r = threadIdx.x; r < n_rec; r += 256) {
#pragma unroll
      for (int i = 0; i < NV; ++i) a[i] += rec[(size_t)r * stride + i];
    }
  }
  block_reduce_multi<P2, 4>(a, scratch, tot);
}

struct RowStat {
  double n, mu, isd;  // count, mean, 1/std of the in-the-money spots: z = (x - mu)·isd
};

// from (n, Σx, Σx²); one division per ROW here instead of one per trajectory and date
__device__ __forceinline__ RowStat rowstat_of(double n, double sx, double sxx) {
  RowStat r{n, 0.0, 1.0};
  if (n > 0.0) {
    r.mu = sx / n;
    const double var = sxx / n - r.mu * r.mu;
    r.isd = var > 0.0 ? 1.0 / sqrt(var) : 1.0;
  }
  return r;
}

// Per-lane terms; every form calls exactly these, in trajectory order j = 0..Q-1.  Branch-free: a
// trajectory that is out of the money (or beyond the ensemble: `live` false) adds exact zeros, which
// leave every partial sum bit for bit as it was — so the compiler can interleave the dependent
// chains of a lane's trajectories instead of serialising them behind one branch each.
__device__ __forceinline__ bool in_the_money(double x, double cp, double strike, bool live) {
  return live && cp * (x - strike) > 0.0;  // payoff_t .> 0 (:118-119)
}
__device__ __forceinline__ void add_stats(double x, double cp, double strike, bool live, double* v) {
  const bool itm = in_the_money(x, cp, strike, live);
  const double xm = itm ? x : 0.0;
  v[0] += itm ? 1.0 : 0.0;
  v[1] += xm;
  v[2] = fma(xm, xm, v[2]);
}
template <int D>
__device__ __forceinline__ void add_powers(double x, double cp, double strike, bool live,
                                           const RowStat& r, double* v) {  // v[2D+1] += z^i
  const double z = (x - r.mu) * r.isd;
  double pw = in_the_money(x, cp, strike, live) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i <= 2 * D; ++i) {
    v[i] += pw;
    pw *= z;
  }
}
// the same without Σ z^0 (the count n): v[k-1] += z^k, k = 1..2D, the products formed exactly as above
template <int D>
__device__ __forceinline__ void add_powers_from1(double x, double cp, double strike, bool live,
                                                 const RowStat& r, double* v) {
  const double z = (x - r.mu) * r.isd;
  double pw = in_the_money(x, cp, strike, live) ? z : 0.0;  // = 1.0 * z
#pragma unroll
  for (int i = 0; i < 2 * D; ++i) {
    v[i] += pw;
    pw *= z;
  }
}
// Σ z^i y, y = D^(tau - row) val (least_squares_montecarlo.jl:115-116); tau >= row + 1
template <int D>
__device__ __forceinline__ void add_moments(double x, double cp, double strike, bool live,
                                            const RowStat& r, double y, double* v) {  // v[D+1]
  const double z = (x - r.mu) * r.isd;
  double pw = in_the_money(x, cp, strike, live) ? y : 0.0;
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    v[i] += pw;
    pw *= z;
  }
}

// 1/x to <= 1 ulp: hardware reciprocal + two Newton steps (5 dependent instructions; the IEEE
// division sequence is ~12).  Pivots are far from the overflow / underflow thresholds.
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
}

// Normal equations G c = B, G_jk = P[j+k] = Σ z^(j+k) (P[0] = the in-the-money count, handed over
// separately as p0; Pm1 points at P[1]).  G is a Gram matrix — symmetric positive (semi-)definite —
// so elimination needs no pivoting (it is the LDLᵀ factorisation, stable as it stands); a column
// whose pivot has vanished against ITS OWN diagonal entry Σ z^2c (fewer distinct in-the-money spots than
// coefficients) is dropped, its coefficient 0.  (Until late round 3 the pivot was held against the
// largest diagonal entry: with heavy-tailed z and degree 7-8 that one, Σ z^16, is 10^13 times the
// count Σ z^0, and the test dropped the constant and the linear column of a perfectly determined fit.)  Every workgroup of either form runs this on the same
// sums and so obtains the same coefficients.  One-thread form:
template <int D>
__device__ void solve_normal_equations(const double* B, double p0, const double* Pm1, double* coef) {
  constexpr int N = D + 1;
  auto Pv = [&](int i) { return i == 0 ? p0 : Pm1[i - 1]; };
  double M[N][N + 1];
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int k = 0; k < N; ++k) M[j][k] = Pv(j + k);
    M[j][N] = B[j];
  }
  double inv[N];
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const bool dead = !(M[c][c] > 1e-13 * Pv(2 * c));
    inv[c] = dead ? 0.0 : rcp_nr(M[c][c]);
    if (!dead) {
#pragma unroll
      for (int j = 0; j < N; ++j)
        if (j > c) {
          const double f = M[j][c] * inv[c];
#pragma unroll
          for (int k = 0; k <= N; ++k)
            if (k >= c) M[j][k] = fma(-f, M[c][k], M[j][k]);
        }
    }
  }
  double cf[N];
#pragma unroll
  for (int c = N - 1; c >= 0; --c) {
    double s = M[c][N];
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (k > c) s = fma(-M[c][k], cf[k], s);
    cf[c] = s * inv[c];  // 0 for a dropped column
  }
#pragma unroll
  for (int c = 0; c < N; ++c) coef[c] = cf[c];
}

// The same elimination with the rows spread over the lanes of one wave (lane j holds row j in
// registers, the pivot row travels by v_readlane): the same operations on the same operands in the
// same order per element — bit-identical coefficients — at ~300 instructions, most of them
// independent across the lanes.  Called by all 64 lanes of a wave; coef written by lane 0.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int D>
__device__ void solve_normal_equations_wave(const double* B, double p0, const double* Pm1,
                                            double* coef) {
  constexpr int N = D + 1;
  const int lane = threadIdx.x & 63;
  const int row = lane < N ? lane : 0;  // lanes >= N mirror row 0 and are never read
  auto Pv = [&](int i) { return i == 0 ? p0 : Pm1[i - 1]; };
  double r[N + 1];
#pragma unroll
  for (int k = 0; k < N; ++k) r[k] = Pv(row + k);
  r[N] = B[row];
  double inv[N];
#pragma unroll
  for (int c = 0; c < N; ++c) {
    double prow[N + 1];  // row c (wave-uniform); final, since only the rows below it change from here
#pragma unroll
    for (int k = c; k <= N; ++k) prow[k] = readlane_f64(r[k], c);
    const bool dead = !(prow[c] > 1e-13 * Pv(2 * c));
    inv[c] = dead ? 0.0 : rcp_nr(prow[c]);
    if (!dead && lane > c) {
      const double f = r[c] * inv[c];
#pragma unroll
      for (int k = c; k <= N; ++k) r[k] = fma(-f, prow[k], r[k]);
    }
  }
  // back-substitution: lane c holds the final row c; every lane runs the same instructions on its
  // own row and lane c's product is broadcast
  double cf[N];
#pragma unroll
  for (int c = N - 1; c >= 0; --c) {
    double s = r[N];
#pragma unroll
    for (int k = c + 1; k < N; ++k) s = fma(-r[k], cf[k], s);
    cf[c] = readlane_f64(s * inv[c], c);  // 0 for a dropped column
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < N; ++c) coef[c] = cf[c];
  }
}

#ifndef HH_LSM_WAVE_SOLVE
#define HH_LSM_WAVE_SOLVE 1
#endif
// timing diagnostics only (results are WRONG with any bit set; tools/lsm_breakdown.py builds variants):
// 1 = the gather does not wait for granules that fail their check, 2 = no solve, 4 = no partial sums / reductions / publish
#ifndef HH_LSM_DEBUG
#define HH_LSM_DEBUG 0
#endif
// 1: diagnostic build that stamps the phases of a date (s_memrealtime, 100 MHz) in thread 0 of one
// workgroup and leaves the totals behind the row counters (tools/lsm_breakdown.py reads them through
// hh_lsm_debug_read); in the shipped build no stamp executes
#ifndef HH_LSM_STAMPS
#define HH_LSM_STAMPS 0
#endif
constexpr int kLsmStampSlots = 8;

// coefficients of row t into LDS (coef[D+1], *have_fit) from the global sums B[0..D], P[0] = n_itm,
// P[1..2D] = Pm1[0..]; all threads call it
// Workgroup barrier for LDS traffic only.  __syncthreads() is a release/acquire fence on ALL memory:
// with a global store or load outstanding the compiler puts s_waitcnt vmcnt(0) in front of s_barrier,
// which drains the published record's write-through stores (~1.5 µs) and the NEXT row's prefetch
// (~2 µs of HBM latency) on every barrier of the persistent induction.  What its barriers order is
// LDS (partial sums, coefficients); the record protocol needs no fence at all (self-validating granules).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the solve alone, by the threads of wave 0 (no barrier): coef[D+1], *have_fit
template <int D>
__device__ __forceinline__ void fit_row_wave0(double n_itm, const double* B, const double* Pm1, double* coef,
                                              int* have_fit) {
#if HH_LSM_WAVE_SOLVE
  if (n_itm > 0.0 && !(HH_LSM_DEBUG & 2))
    solve_normal_equations_wave<D>(B, n_itm, Pm1, coef);  // isempty(in_the_money) && continue (:120)
  if (threadIdx.x == 0) *have_fit = n_itm > 0.0 ? 1 : 0;
#else
  if (threadIdx.x == 0) {
    if (n_itm > 0.0) solve_normal_equations<D>(B, n_itm, Pm1, coef);
    *have_fit = n_itm > 0.0 ? 1 : 0;
  }
#endif
}
template <int D>
__device__ __forceinline__ void fit_row(double n_itm, const double* B, const double* Pm1, double* coef,
                                        int* have_fit) {
  if (threadIdx.x < 64) fit_row_wave0<D>(n_itm, B, Pm1, coef, have_fit);
  __syncthreads();
}

// exercise decision of one trajectory at row t (update_stopping_info!, :163-164); branch-free
template <int D>
__device__ __forceinline__ bool exercise_now(double x, double cp, double strike, bool live,
                                             const RowStat& r, const double* coef, double& pay) {
  pay = cp * (x - strike);
  const double z = (x - r.mu) * r.isd;
  double cont = coef[D];  // cont_value = poly(x) (:127), Horner in z
#pragma unroll
  for (int c = D - 1; c >= 0; --c) cont = fma(cont, z, coef[c]);
  return live && pay > 0.0 && pay > cont;
}

// ---- one launch per date: per-row statistics and power sums (one-off), then the steps -----------

template <int Q>
__global__ __launch_bounds__(kLsmWg) void lsm_stats_kernel(const double* __restrict__ grid,
                                                           uint64_t ntot, double strike, double cp,
                                                           uint32_t n_chunks,
                                                           double* __restrict__ rec /*[row][chunk][3]*/) {
  __shared__ double scratch[kLsmWaves * 4], tot[4];
  const uint32_t chunk = blockIdx.x, row = blockIdx.y;
  const double* S = grid + (size_t)row * ntot;
  double v[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = (uint64_t)chunk * (kLsmWg * Q) + (uint64_t)j * kLsmWg + threadIdx.x;
    const bool lv = p < ntot;
    add_stats(lv ? S[p] : 0.0, cp, strike, lv, v);
  }
  block_reduce_multi<4, kLsmWaves>(v, scratch, tot);
  if (threadIdx.x < 3) rec[((size_t)row * n_chunks + chunk) * 3 + threadIdx.x] = tot[threadIdx.x];
}

// rs[row] from the chunk records of the row (grid = rows)
__global__ __launch_bounds__(kLsmWg) void lsm_rowstat_kernel(const double* __restrict__ rec,
                                                             uint32_t n_chunks,
                                                             RowStat* __restrict__ rs) {
  __shared__ double scratch[4 * 4], tot[4];
  const uint32_t row = blockIdx.x;
  reduce_chunk_records<3, 4>(rec + (size_t)row * n_chunks * 3, n_chunks, 3, scratch, tot);
  if (threadIdx.x == 0) rs[row] = rowstat_of(tot[0], tot[1], tot[2]);
}

// sharded form: the sums leave the device (all-reduce over the ranks) and come back
__global__ __launch_bounds__(256) void lsm_rowstat_from_sums_kernel(const double* __restrict__ sums,
                                                                    uint32_t rows,
                                                                    RowStat* __restrict__ rs) {
  const uint32_t row = blockIdx.x * 256 + threadIdx.x;
  if (row < rows) rs[row] = rowstat_of(sums[row * 3], sums[row * 3 + 1], sums[row * 3 + 2]);
}

// out[row][NV] = canonical sum over the chunk records rec[row][chunk][NV] (grid = rows)
template <int NV>
__global__ __launch_bounds__(kLsmWg) void lsm_sum_records_kernel(const double* __restrict__ rec,
                                                                 uint32_t n_chunks,
                                                                 double* __restrict__ out) {
  constexpr int P2 = pow2_ge(NV);
  __shared__ double scratch[4 * P2], tot[P2];
  const uint32_t row = blockIdx.x;
  reduce_chunk_records<NV, P2>(rec + (size_t)row * n_chunks * NV, n_chunks, NV, scratch, tot);
  if (threadIdx.x < NV) out[(size_t)row * NV + threadIdx.x] = tot[threadIdx.x];
}

template <int D, int Q>
__global__ __launch_bounds__(kLsmWg) void lsm_pow_kernel(const double* __restrict__ grid, uint64_t ntot,
                                                         double strike, double cp, uint32_t n_chunks,
                                                         const RowStat* __restrict__ rs,
                                                         double* __restrict__ rec /*[row][chunk][2D+1]*/) {
  constexpr int NV = 2 * D + 1, P2 = pow2_ge(NV);
  __shared__ double scratch[kLsmWaves * P2], tot[P2];
  const uint32_t chunk = blockIdx.x, row = blockIdx.y;
  const double* S = grid + (size_t)row * ntot;
  const RowStat r = rs[row];
  double v[P2];
#pragma unroll
  for (int i = 0; i < P2; ++i) v[i] = 0.0;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = (uint64_t)chunk * (kLsmWg * Q) + (uint64_t)j * kLsmWg + threadIdx.x;
    const bool lv = p < ntot;
    add_powers<D>(lv ? S[p] : 0.0, cp, strike, lv, r, v);
  }
  block_reduce_multi<P2, kLsmWaves>(v, scratch, tot);
  if (threadIdx.x < NV) rec[((size_t)row * n_chunks + chunk) * NV + threadIdx.x] = tot[threadIdx.x];
}

struct LsmStepArgs {
  const double* grid;
  uint64_t ntot;
  double strike, cp, ln_disc;  // ln of the per-step discount factor
  uint32_t n_steps, n_chunks;  // n_chunks: canonical chunks (512·Q trajectories each: 1024 or 8192)
  int32_t* tau;
  double* val;
  const RowStat* rs;
  const double* P;   // [row][2D+1]
  double* recB;      // [row][chunk][D+1]: partial Σ z^k y of the row
  const double* disc_pow;  // [k] = exp(ln_disc·k), k = 0..n_steps: discount over k exercise dates
  const double* B_given;   // sharded solve: the GLOBAL moment sums of the row (D+1 doubles), else NULL
  double* counters;  // [0] rows regressed, [1] rows skipped (no in-the-money path)
};

// discount^k for every k the induction can ask for: one table instead of an exp() per path and date
__global__ __launch_bounds__(256) void lsm_disc_kernel(double ln_disc, uint32_t n, double* out) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k <= n) out[k] = exp(ln_disc * (double)k);
}

// write this chunk's partial Σ z^k y of `row` (lane partials in v) to recB[row][chunk]
template <int D>
__device__ __forceinline__ void store_moments(const LsmStepArgs& a, uint32_t row,
                                              double (&v)[pow2_ge(D + 1)], double* scratch,
                                              double* tot) {
  constexpr int NV = D + 1, P2 = pow2_ge(NV);
  block_reduce_multi<P2, kLsmWaves>(v, scratch, tot);
  if (threadIdx.x < NV)
    a.recB[((size_t)row * a.n_chunks + blockIdx.x) * NV + threadIdx.x] = tot[threadIdx.x];
}

// stopping_info = [(nsteps, payoff(S_T))] (:109), and the moment sums of row nsteps-1
template <int D, int Q>
__global__ __launch_bounds__(kLsmWg) void lsm_init_kernel(const LsmStepArgs a) {
  constexpr int P2 = pow2_ge(D + 1);
  __shared__ double scratch[kLsmWaves * P2], tot[P2];
  const uint32_t M = a.n_steps;
  const double* S = a.grid + (size_t)M * a.ntot;
  const double* Sn = a.grid + (size_t)(M >= 2 ? M - 1 : M) * a.ntot;
  const RowStat r = a.rs[M >= 2 ? M - 1 : M];
  const double d1 = a.disc_pow[1];
  double v[P2];
#pragma unroll
  for (int i = 0; i < P2; ++i) v[i] = 0.0;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = (uint64_t)blockIdx.x * (kLsmWg * Q) + (uint64_t)j * kLsmWg + threadIdx.x;
    const bool lv = p < a.ntot;
    const double m = a.cp * ((lv ? S[p] : 0.0) - a.strike);
    const double val = m > 0.0 ? m : 0.0;
    if (lv) {
      a.tau[p] = (int)M;
      a.val[p] = val;
    }
    if (M >= 2) add_moments<D>(lv ? Sn[p] : 0.0, a.cp, a.strike, lv, r, d1 * val, v);
  }
  if (M >= 2) store_moments<D>(a, M - 1, v, scratch, tot);
}

// one backward step at time index t (the reference's loop body for i = t+1, :112-131): decisions at
// row t, then this chunk's moment sums of row t-1 with the stopping state as of now.  One pass over
// the chunk's trajectories, nothing kept per trajectory (the persistent form is the one that keeps
// the state in registers).
template <int D, int Q>
__global__ __launch_bounds__(kLsmWg) void lsm_step_kernel(const LsmStepArgs a, uint32_t t) {
  constexpr int N = D + 1, P2 = pow2_ge(N);
  __shared__ double scratch[kLsmWaves * P2], tot[P2], coef[N], Pt[2 * D + 1];
  __shared__ int have_fit;
  const RowStat r = a.rs[t];
  const RowStat rn = a.rs[t >= 2 ? t - 1 : t];
  const double* S = a.grid + (size_t)t * a.ntot;
  const double* Sn = a.grid + (size_t)(t >= 2 ? t - 1 : t) * a.ntot;  // row of the next step
  // this chunk's trajectories first (clamped addresses: the loads issue back to back): they are in
  // flight while the moment sums are reduced and the normal equations solved
  int tau[Q];
  double val[Q], xs[Q], xn[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = (uint64_t)blockIdx.x * (kLsmWg * Q) + (uint64_t)j * kLsmWg + threadIdx.x;
    const uint64_t pc = p < a.ntot ? p : a.ntot - 1;
    tau[j] = a.tau[pc];
    val[j] = a.val[pc];
    xs[j] = S[pc];
    xn[j] = Sn[pc];
  }
  if (a.B_given) {  // summed over the ranks by the host between two launches
    if (threadIdx.x < N) tot[threadIdx.x] = a.B_given[threadIdx.x];
  } else {
    reduce_chunk_records<N, P2>(a.recB + (size_t)t * a.n_chunks * N, a.n_chunks, N, scratch, tot);
  }
  if (threadIdx.x < 2 * D + 1) Pt[threadIdx.x] = a.P[(size_t)t * (2 * D + 1) + threadIdx.x];
  __syncthreads();
  fit_row<D>(r.n, tot, Pt + 1, coef, &have_fit);  // P[0] = Σ 1 = r.n exactly
  if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[r.n > 0.0 ? 0 : 1] += 1.0;
  const bool fit = have_fit != 0;

  double v[P2];
#pragma unroll
  for (int i = 0; i < P2; ++i) v[i] = 0.0;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = (uint64_t)blockIdx.x * (kLsmWg * Q) + (uint64_t)j * kLsmWg + threadIdx.x;
    const bool lv = p < a.ntot;
    double pay;
    if (fit && exercise_now<D>(xs[j], a.cp, a.strike, lv, r, coef, pay)) {
      tau[j] = (int)t;
      val[j] = pay;
      a.tau[p] = (int)t;
      a.val[p] = pay;
    }
    if (t >= 2)
      add_moments<D>(xn[j], a.cp, a.strike, lv, rn, a.disc_pow[tau[j] - (int)(t - 1)] * val[j], v);
  }
  __syncthreads();  // tot (the sums of row t) was read by fit_row's wave
  if (t >= 2) store_moments<D>(a, t - 1, v, scratch, tot);
}

// discounted_values = discount^t * val (:133): per-workgroup Σ and Σ² into 16-double records
__global__ __launch_bounds__(256) void lsm_final_kernel(const int32_t* __restrict__ tau,
                                                        const double* __restrict__ val,
                                                        uint64_t ntot, double ln_disc,
                                                        double* __restrict__ records) {
  __shared__ double scratch[4 * 2], tot[2];
  double v[2] = {0, 0};
  for (int j = 0; j < 4; ++j) {
    const uint64_t p = (uint64_t)blockIdx.x * kLsmFinalChunk + j * 256 + threadIdx.x;
    if (p < ntot) {
      const double d = exp(ln_disc * (double)tau[p]) * val[p];
      v[0] += d;
      v[1] = fma(d, d, v[1]);
    }
  }
  block_reduce_multi<2, 4>(v, scratch, tot);
  if (threadIdx.x == 0) {
    double* rec = records + (size_t)blockIdx.x * kRecStride;
    for (int i = 0; i < kRecStride; ++i) rec[i] = 0.0;
    rec[HH_ACC_SUM] = tot[0];
    rec[HH_ACC_SUMSQ] = tot[1];
  }
}

// ---- one persistent launch ---------------------------------------------------------------------

using gu64 = __attribute__((address_space(1))) unsigned long long;
using gu32 = __attribute__((address_space(1))) unsigned int;

struct LsmPersistArgs {
  const double* grid;
  uint64_t ntot;
  double strike, cp;
  uint32_t n_steps, n_chunks;
  int32_t* tau;
  double* val;
  const double* disc_pow;
  double* counters;
  // all-gather state, zeroed by a memset node ahead of every launch
  unsigned long long* rec;  // [kLsmRing][32 values][n_chunks] GRANULES of 16 bytes {bits, bits ^ epoch}
                            // (fp64 bit patterns), write-through: value-major, so that the 64 lanes of a
                            // gathering wave — one record each — read 1 KiB of contiguous bytes per value
  unsigned int* status;     // [0] != 0: a workgroup gave up waiting (the grid was not co-resident)
  unsigned long long spin_ticks;  // bound of every wait, in s_memrealtime ticks (100 MHz)
  double ln_disc;           // log of the per-step discount: the table discount^k is formed in the kernel
  unsigned int nonce;       // of this launch (never 0, process-wide counter): part of every granule's check
};

// Record of an epoch e (row t = M - e + 1), 32 doubles in two groups of 16 — the groups are formed,
// reduced and gathered separately, which halves the registers a lane needs for its partial sums:
//   group A  [0, D]  Σ z^k y of row t (k = 0..D)      [12, 14]  (n, Σx, Σx²) of row t-1
//   group B  [16 + k - 1]  Σ z^k of row t, k = 1..2D  (Σ z^0 is the in-the-money count n)
// Epoch 1 carries only the statistics of row M-1.
constexpr int kRecP2 = 32, kGrp = 16, kOffStats = 12;
constexpr int kDiscLds = 1024;

// Every value travels as a self-validating 16-byte granule {v, v ^ (nonce:e)} (v the fp64 bit pattern,
// e the epoch, never 0, nonce the launch's number: a granule of an earlier launch — same epochs, same
// addresses — cannot validate even if some cache still held it): no tag word, hence no "drain the stores, then raise the tag" on the publisher's side
// and no "poll the tag, then load" on the reader's — one store instruction to publish, one round trip to
// gather.  A granule that reads back as anything but {v, v ^ e} (zeros from the per-launch memset, the
// slot's previous epoch, two halves of different ages) is simply not there yet.  That also makes it
// safe to read through the caches: the FIRST sweep of a gather uses plain loads, which the XCD's L2
// serves to all but the first of its ~31 workgroups (every workgroup reads every record: through sc1
// loads that was 245² x 19 x 8 B = 9 MB of fabric traffic per date, and two dependent round trips); a
// stale or partly written line in L1 / L2 fails the validation and only then is re-read with sc1 loads,
// which see the other XCDs' write-through stores.
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = 16, kAuxVolatile = (int)0x80000000u;  // cache-policy bits of the raw buffer intrinsics (gfx94x/95x)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t record_ring(const LsmPersistArgs& a) {
  return __builtin_amdgcn_make_buffer_rsrc(a.rec, 0, (int)((size_t)kLsmRing * kRecP2 * a.n_chunks * 16), 0x00020000);
}
__device__ __forceinline__ uint32_t granule_offset(const LsmPersistArgs& a, uint32_t e, int value, uint32_t r) {
  return (((e % kLsmRing) * kRecP2 + (uint32_t)value) * a.n_chunks + r) * 16u;
}

// publish group g (0 = A, 1 = B) of this workgroup's record for epoch e: one write-through (sc1) store
// of 16 granules by the first 16 lanes of a wave.  `val` is the lane's value (lanes >= 16: unused).
__device__ __forceinline__ void publish_group(const LsmPersistArgs& a, uint32_t e, int g, double val) {
  const uint32_t lane = threadIdx.x & 63u;
  if (lane < (uint32_t)kGrp) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(val);
    const u32x4 gr = {(unsigned)b, (unsigned)(b >> 32), (unsigned)b ^ e, (unsigned)(b >> 32) ^ a.nonce};
    __builtin_amdgcn_raw_buffer_store_b128(gr, record_ring(a), granule_offset(a, e, g * kGrp + (int)lane, blockIdx.x),
                                           0, kAuxSc1);
  }
}
// both groups from the workgroup totals tot[32] (the two prologue epochs)
__device__ __forceinline__ void publish_record(const LsmPersistArgs& a, uint32_t e, const double* tot) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave == 0) publish_group(a, e, 0, lane < kGrp ? tot[lane] : 0.0);
  if (wave == 4) publish_group(a, e, 1, lane < kGrp ? tot[kGrp + lane] : 0.0);
}

// gather the records of epoch e from every workgroup and reduce them in the canonical order: lane
// r < 256 of waves 0-3 takes group A of record r, lane r of waves 4-7 group B, and loads the granules of
// the values a record of degree D carries (9 + 10 of the 32 for D = 5; the others are zero on both
// sides) until all of them validate.  tot[32] on return.  Returns false — for every thread of the
// workgroup — when a wait ran out or another workgroup gave up; the caller leaves the kernel.
template <int D>
__device__ __forceinline__ bool gather_records(const LsmPersistArgs& a, uint32_t e, double* scratch,
                                               double* tot, int* ok_flag) {
  // no barrier on entry (*ok_flag is set once, at the start of the kernel, and only ever cleared): the
  // waves start loading at once, whatever another wave of the workgroup is still doing; `scratch` is
  // the gather's own (gscratch), nobody else writes it
  double v[kGrp];
#pragma unroll
  for (int i = 0; i < kGrp; ++i) v[i] = 0.0;
  const int wave = threadIdx.x >> 6;
  static_assert(kLsmWg >= 512, "the gather uses 8 waves");
  if (threadIdx.x < 512) {
    const int g = threadIdx.x >> 8;           // group A or B (wave-uniform)
    const uint32_t r = threadIdx.x & 255u;    // record
    const bool mine = r < a.n_chunks;
    const __amdgpu_buffer_rsrc_t ring = record_ring(a);
    bool ok = a.spin_ticks != 0;  // 0 (HH_OPT_LSM_SPIN_TICKS, tests): give up at once
    // NV values of group G, value i of them at index IDX(i) of the group
    auto take = [&](auto nv_c, auto idx) {
      constexpr int NV = decltype(nv_c)::value;
      u32x4 gr[NV];
      auto sweep = [&](auto aux_c) {
        constexpr int aux = decltype(aux_c)::value;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < NV; ++i)
          gr[i] = __builtin_amdgcn_raw_buffer_load_b128(ring, granule_offset(a, e, g * kGrp + idx(i), mine ? r : 0u), 0, aux);
#pragma unroll
        for (int i = 0; i < NV; ++i) bad |= ((gr[i].x ^ gr[i].z) != e) | ((gr[i].y ^ gr[i].w) != a.nonce);
        return bad && mine;
      };
      if (ok) {
        bool bad = sweep(std::integral_constant<int, 0>{});  // through the caches (a volatile access is made sc0 sc1)
        if (__any(bad) && !(HH_LSM_DEBUG & 1)) {
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          unsigned spins = 0;
          while (true) {
            bad = sweep(std::integral_constant<int, kAuxVolatile | kAuxSc1>{});
            if (!__any(bad)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0) {  // bounded: every wave reaches an exit
              const unsigned gave_up = __hip_atomic_load((gu32*)a.status, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
              if (gave_up != 0 || __builtin_amdgcn_s_memrealtime() - t0 > a.spin_ticks) {
                ok = false;
                break;
              }
            }
          }
        }
      }
      if (ok && mine) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
          v[idx(i)] = __longlong_as_double((long long)(((unsigned long long)gr[i].y << 32) | gr[i].x));
      }
    };
    // (degrees 7 and 8 in two batches of at most 8 granules — 4 registers each — or they spill)
    constexpr int NA = D + 4, NB = 2 * D > 3 ? 2 * D : 3;  // epoch 1: three statistics in group B
    auto idx_a = [](int i) { return i <= D ? i : kOffStats + (i - D - 1); };
    if (g == 0) {
      if constexpr (NA > 10) {
        take(std::integral_constant<int, 8>{}, idx_a);
        take(std::integral_constant<int, NA - 8>{}, [idx_a](int i) { return idx_a(i + 8); });
      } else {
        take(std::integral_constant<int, NA>{}, idx_a);
      }
    } else {
      if constexpr (NB > 12) {
        take(std::integral_constant<int, 8>{}, [](int i) { return i; });
        take(std::integral_constant<int, NB - 8>{}, [](int i) { return i + 8; });
      } else {
        take(std::integral_constant<int, NB>{}, [](int i) { return i; });
      }
    }
    if (!ok && (threadIdx.x & 63) == 0) {
      __hip_atomic_store((gu32*)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ok_flag = 0;
    }
    wave_reduce_multi<kGrp>(v);
    if ((threadIdx.x & 3) == 0) scratch[wave * kGrp + ((threadIdx.x & 63) >> 2)] = v[0];
  }
  lds_barrier();
  if (threadIdx.x < kRecP2) {  // the four waves of a group, in order
    const int g = threadIdx.x >> 4, i = threadIdx.x & 15;
    double t = scratch[(4 * g) * kGrp + i];
#pragma unroll
    for (int w = 1; w < 4; ++w) t += scratch[(4 * g + w) * kGrp + i];
    tot[threadIdx.x] = t;
  }
  lds_barrier();
  return *ok_flag != 0;
}

// Register budget: 512 threads = 2 waves per SIMD = 256 registers per lane for the form with 16
// trajectories per lane (stopping state + row t of all of them live in registers); the form with 2
// per lane is held to 128 (4 waves per SIMD) — it needs ~105, and between 129 and 256 the register
// allocator starts using AGPRs, which v_readlane (the wave-parallel solve) cannot read (hipcc 7.2
// stops with "Illegal instruction detected: Operand has incorrect register class").
template <int D, int Q>
__global__ __launch_bounds__(kLsmWg, (Q * kLsmWg > 1024 ? 2 : 4)) void lsm_persistent_kernel(
    const LsmPersistArgs a) {
  constexpr int N = D + 1;
  static_assert(N <= kOffStats && 2 * D <= kGrp, "record layout");
  __shared__ double scratch[kLsmWaves * kRecP2], gscratch[kLsmWaves * kGrp], tot[kRecP2], coef[N];
  __shared__ double dtab[kDiscLds];  // discount^k, k <= n_steps, when the table fits
  // rows t-1 and t-2 of this workgroup's trajectories wait here (a lane only ever touches its own
  // slots): registers hold the stopping state and row t, the partial sums need the rest
  __shared__ double xl[2][Q * kLsmWg];
  __shared__ int have_fit, ok_flag;
  const uint32_t M = a.n_steps;
  const uint64_t p0 = (uint64_t)blockIdx.x * (kLsmWg * Q) + threadIdx.x;
  auto row_ptr = [&](uint32_t row) { return a.grid + (size_t)row * a.ntot; };
  auto live = [&](int j) { return p0 + (uint64_t)j * kLsmWg < a.ntot; };
  const bool disc_lds = M < (uint32_t)kDiscLds;
  if (disc_lds)  // lsm_disc_kernel's expression: the same table, without its launch
    for (uint32_t k = threadIdx.x; k <= M; k += kLsmWg) dtab[k] = exp(a.ln_disc * (double)k);
  auto disc = [&](int k) { return disc_lds ? dtab[k] : a.disc_pow[k]; };
  if (threadIdx.x == 0) ok_flag = 1;
  __syncthreads();

  int tau[Q];
  double val[Q], xs[Q];  // xs = row t, where the next decisions are taken
  double xin[Q];         // a row on its way from memory: issued one date before it is first used
  // A lane's 16 slots of a row: byte offsets off0 + j·(512·8), clamped to the row's last trajectory
  // (never guarded: the loads issue back to back).  The offsets are re-derived from ONE register at
  // every issue — kept opaque to the compiler, which otherwise hoists the 16 clamped 64-bit addresses
  // out of the date loop, spills them (the kernel sits at its 256-register budget) and reloads them
  // from scratch between the loads: every reload's s_waitcnt vmcnt then also waits for the row loads
  // issued before it, 4 x the HBM latency per date.
  const uint32_t off_last = (uint32_t)((a.ntot - 1) * 8);
  uint32_t off0 = (uint32_t)((p0 < a.ntot ? p0 : a.ntot - 1) * 8);
  auto issue_row = [&](uint32_t row) {
    const char* Sr = reinterpret_cast<const char*>(row_ptr(row));
    asm volatile("" : "+v"(off0));
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const uint32_t o = off0 + (uint32_t)j * (kLsmWg * 8u);
      xin[j] = *reinterpret_cast<const double*>(Sr + (o < off_last ? o : off_last));
    }
  };
  {
    const double* SM = row_ptr(M);
    const double* S1 = row_ptr(M >= 2 ? M - 1 : M);
    const double* S2 = row_ptr(M >= 3 ? M - 2 : M);
    const double* S3 = row_ptr(M >= 4 ? M - 3 : M);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const uint64_t p = p0 + (uint64_t)j * kLsmWg;
      tau[j] = (int)M;
      val[j] = xs[j] = 0.0;
      double x2 = 0.0, x3 = 0.0;
      if (p < a.ntot) {
        const double m = a.cp * (SM[p] - a.strike);  // stopping_info = [(nsteps, payoff(S_T))] (:109)
        val[j] = m > 0.0 ? m : 0.0;
        xs[j] = S1[p];
        x2 = S2[p];
        x3 = S3[p];
      }
      xl[0][j * kLsmWg + threadIdx.x] = x2;  // row M-2
      xl[1][j * kLsmWg + threadIdx.x] = x3;  // row M-3
    }
  }
  if (M >= 5) issue_row(M - 4);
  int cur = 0;  // at date t: xl[cur] = row t-1, xl[cur ^ 1] = row t-2, xin = row t-3
  double regressed = 0.0, skipped = 0.0;
#if HH_LSM_STAMPS
  unsigned long long st_acc[kLsmStampSlots] = {}, st_t0 = __builtin_amdgcn_s_memrealtime();
#ifndef HH_LSM_STAMP_THREAD
#define HH_LSM_STAMP_THREAD 0  // which thread of the middle workgroup stamps (waves 0-3 and 4-7 differ, below)
#endif
  const bool st_me = blockIdx.x == gridDim.x / 2 && threadIdx.x == HH_LSM_STAMP_THREAD;
#define HH_STAMP(k)                                                  \
  if (st_me) {                                                       \
    const unsigned long long now = __builtin_amdgcn_s_memrealtime(); \
    st_acc[k] += now - st_t0;                                        \
    st_t0 = now;                                                     \
  }
#else
#define HH_STAMP(k)
#endif
  // The pipeline of a date t (epoch e = M - t + 1).  What couples the workgroups is one record per
  // date; its two halves travel separately, because only one of them depends on the stopping state:
  //   group A, epoch e   Σ z^k y of row t (k = 0..D; needs the decisions of date t+1)  +  (n, Σx, Σx²) of row t-2
  //   group B, epoch e   Σ z^k of row t (k = 1..2D): the row and its statistics alone
  // The CRITICAL path of a date is  gather A_e → normal equations of row t → decisions at row t →
  // moment sums of row t-1 → workgroup total → publish A_{e+1}.  Everything else runs BEHIND that
  // publication, while the other workgroups' A records are on their way: the power sums of row t-2
  // (their statistics arrived with A_e) are published as B_{e+2}, two gathers before they are needed,
  // and row t-4 is issued from memory.  Statistics therefore run three rows ahead of the decisions,
  // power sums two, moment sums one.  Same per-trajectory terms and the same summation tree as the
  // launch-per-date form (bit-identical results); only WHEN a sum is formed differs.
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // group g of the workgroup's sums out of the waves' partial sums (the waves in order, as
  // finish_block adds them), straight into the record, by one wave
  auto total_and_publish = [&](uint32_t e, int g, int by_wave) {
    if (wave == by_wave) {
      double t = 0.0;
      if (lane < kGrp) {
        t = scratch[g * kGrp + lane];
#pragma unroll
        for (int w = 1; w < kLsmWaves; ++w) t += scratch[w * kRecP2 + g * kGrp + lane];
      }
      publish_group(a, e, g, t);
    }
  };
  bool alive = true;
  if (M >= 2) {
    {  // epoch 1: statistics of rows M-1 (A) and M-2 (B)
      double v[kGrp], w[kGrp];
#pragma unroll
      for (int i = 0; i < kGrp; ++i) v[i] = w[i] = 0.0;
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        add_stats(xs[j], a.cp, a.strike, live(j), v + kOffStats);
        if (M >= 3) add_stats(xl[0][j * kLsmWg + threadIdx.x], a.cp, a.strike, live(j), w);
      }
      wave_part<kGrp, kRecP2>(v, scratch, 0);
      wave_part<kGrp, kRecP2>(w, scratch, kGrp);
      finish_block<kRecP2, kLsmWaves>(scratch, tot);
      publish_record(a, 1u, tot);
      alive = gather_records<D>(a, 1u, gscratch, tot, &ok_flag);
    }
    RowStat r_cur = rowstat_of(tot[kOffStats], tot[kOffStats + 1], tot[kOffStats + 2]);  // row M-1
    RowStat r_next = rowstat_of(tot[kGrp], tot[kGrp + 1], tot[kGrp + 2]);                // row M-2
    RowStat r_nn = r_next;
    __syncthreads();  // tot is rewritten below
    if (alive) {
      // epoch 2: moment sums of row M-1 (tau = M everywhere) + statistics of row M-3 (A), power sums of
      // row M-1 (B); and the power sums of row M-2 as B of epoch 3
      const double d1 = disc(1);
      {
        double v[kGrp];
#pragma unroll
        for (int i = 0; i < kGrp; ++i) v[i] = 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
          add_moments<D>(xs[j], a.cp, a.strike, live(j), r_cur, d1 * val[j], v);
          if (M >= 4) add_stats(xl[1][j * kLsmWg + threadIdx.x], a.cp, a.strike, live(j), v + kOffStats);
        }
        wave_part<kGrp, kRecP2>(v, scratch, 0);
      }
      {
        double w[kGrp];
#pragma unroll
        for (int i = 0; i < kGrp; ++i) w[i] = 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j)
          add_powers_from1<D>(xs[j], a.cp, a.strike, live(j), r_cur, w);
        wave_part<kGrp, kRecP2>(w, scratch, kGrp);
      }
      finish_block<kRecP2, kLsmWaves>(scratch, tot);
      publish_record(a, 2u, tot);
      if (M >= 3) {
        double w[kGrp];
#pragma unroll
        for (int i = 0; i < kGrp; ++i) w[i] = 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j)
          add_powers_from1<D>(xl[0][j * kLsmWg + threadIdx.x], a.cp, a.strike, live(j), r_next, w);
        wave_part<kGrp, kRecP2>(w, scratch, kGrp);  // finish_block's last barrier is behind every read of scratch
        __syncthreads();
        total_and_publish(3u, 1, 4);
      }
    }
    // for i = nsteps:-1:2, t = i-1 (:112-113).
    // All workgroups run the same schedule, so a record published while the publisher still has work
    // of its own to do is not waited for by anybody: wave 0 stores group A, issues the next row and
    // forms its power sums like every other wave — by the time anybody gathers (it has the same power
    // sums to form first) the granules have long landed.  VMEM operations retire in order: a wave that
    // waits for the granules it loads also waits for everything it issued before them, so nothing slow
    // may be issued just ahead of a gather — the row prefetch goes out BEFORE the power sums, and the
    // stores of group B (write-through, ~1.5 µs to retire) are issued by wave 4 in the NEXT date's solve
    // window, a full date before anybody needs them.
    for (uint32_t t = M - 1; alive && t >= 1; --t) {
      const uint32_t e = M - t + 1;  // epoch whose records hold the sums of row t
      HH_STAMP(7)
      alive = gather_records<D>(a, e, gscratch, tot, &ok_flag);
      HH_STAMP(1)  // all-gather: loads, validation, reduction
      if (!alive) break;
      // The solve is one wave's; the other seven use the window for the one piece of the date's arithmetic
      // that needs neither the coefficients nor the stopping state: the statistics of the row that has
      // just landed (xin = row t-3; wave 0 forms its own beside the moment sums, as before).
      double st[3] = {0.0, 0.0, 0.0};
      if (wave == 0) {
        fit_row_wave0<D>(r_cur.n, tot, tot + kGrp, coef, &have_fit);  // Gram: P[0] = n, P[k] = group B
      } else if (HH_LSM_EARLY_STATS && t >= 4) {
#pragma unroll
        for (int j = 0; j < Q; ++j) add_stats(xin[j], a.cp, a.strike, live(j), st);
      }
      // the power sums date t+1 left in scratch (row t-1): group B of the epoch of date t-1
      if (t + 2 <= M && t >= 2 && !(HH_LSM_DEBUG & 4)) total_and_publish(e + 1, 1, 4);
      lds_barrier();
      HH_STAMP(2)  // normal equations
      // statistics of row t-2 (two divisions and a square root: a 0.3 µs chain for a wave with nothing
      // else to issue): needed by the power sums at the end of the date, so formed beside the decisions,
      // not in front of the solve
      if (t >= 3) r_nn = rowstat_of(tot[kOffStats], tot[kOffStats + 1], tot[kOffStats + 2]);
      if (r_cur.n > 0.0) regressed += 1.0; else skipped += 1.0;
      if (have_fit) {
#pragma unroll
        for (int j = 0; j < Q; ++j) {
          double pay;
          const bool ex = exercise_now<D>(xs[j], a.cp, a.strike, live(j), r_cur, coef, pay);
          tau[j] = ex ? (int)t : tau[j];
          val[j] = ex ? pay : val[j];
        }
      }
      HH_STAMP(3)  // exercise decisions
      if (t >= 2) {
        // row t-1 moves from LDS into the registers (the next decision row) and its moment sums are
        // formed with the stopping state as of now; row t-3 has landed: it takes the freed LDS slots
        // (a lane only ever touches its own) and leaves its statistics
#pragma unroll
        for (int j = 0; j < Q; ++j) xs[j] = xl[cur][j * kLsmWg + threadIdx.x];
        if (!(HH_LSM_DEBUG & 4)) {
        {
          double v[kGrp];
#pragma unroll
          for (int i = 0; i < kGrp; ++i) v[i] = 0.0;
          if (t >= 4) {
            if (HH_LSM_EARLY_STATS && wave != 0) {
#pragma unroll
              for (int j = 0; j < Q; ++j) xl[cur][j * kLsmWg + threadIdx.x] = xin[j];
#pragma unroll
              for (int i = 0; i < 3; ++i) v[kOffStats + i] = st[i];
            } else {
#pragma unroll
              for (int j = 0; j < Q; ++j) {
                xl[cur][j * kLsmWg + threadIdx.x] = xin[j];
                add_stats(xin[j], a.cp, a.strike, live(j), v + kOffStats);
              }
            }
          }
          // discount^(tau - (t-1)): from the LDS table or from memory — two copies of the loop, not a
          // select per lookup: `disc_lds ? dtab[k] : a.disc_pow[k]` compiles to FLAT loads of a selected
          // address, each followed by s_waitcnt vmcnt(0) lgkmcnt(0) (16 serialised round trips per date,
          // every one of them also waiting for the record stores and the row prefetch in flight)
          auto moments = [&](auto disc_of) {
            double y[Q];
#pragma unroll
            for (int j = 0; j < Q; ++j) y[j] = disc_of(tau[j] - (int)(t - 1)) * val[j];
#pragma unroll
            for (int j = 0; j < Q; ++j) add_moments<D>(xs[j], a.cp, a.strike, live(j), r_next, y[j], v);
          };
          if (disc_lds) moments([&](int k) { return dtab[k]; });
          else moments([&](int k) { return a.disc_pow[k]; });
          wave_part<kGrp, kRecP2>(v, scratch, 0);
        }
        HH_STAMP(4)  // moment sums + their wave butterfly
        lds_barrier();
#if HH_LSM_STAMPS > 1
        HH_STAMP(7)  // (diagnostic: the barrier's wait goes to "loop overhead")
#endif
        total_and_publish(e + 1, 0, 0);  // wave 0: group A is on its way — the date's critical path ends here
        HH_STAMP(6)  // workgroup total of A, store issued
        }
        if (t >= 5) issue_row(t - 4);  // row (t-1)-3: lands, is parked and leaves its statistics at date t-1
        HH_STAMP(0)  // next row issued
        if (t >= 3 && !(HH_LSM_DEBUG & 4)) {  // power sums of row t-2 (xl[cur ^ 1]): published from the next window
          double w[kGrp];
#pragma unroll
          for (int i = 0; i < kGrp; ++i) w[i] = 0.0;
#pragma unroll
          for (int j = 0; j < Q; ++j)
            add_powers_from1<D>(xl[cur ^ 1][j * kLsmWg + threadIdx.x], a.cp, a.strike, live(j), r_nn, w);
          wave_part<kGrp, kRecP2>(w, scratch, kGrp);
        }
        HH_STAMP(5)  // power sums + their wave butterfly
        cur ^= 1;
        r_cur = r_next;
        r_next = r_nn;
      }
    }
  }
  if (!alive) return;  // the host sees status != 0 and runs the launch-per-date form instead
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = p0 + (uint64_t)j * kLsmWg;
    if (p < a.ntot) {
      a.tau[p] = tau[j];
      a.val[p] = val[j];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.counters[0] = regressed;
    a.counters[1] = skipped;
  }
#if HH_LSM_STAMPS
  if (st_me)
    for (int k = 0; k < kLsmStampSlots; ++k) a.counters[2 + k] = (double)st_acc[k];
#endif
}
#undef HH_STAMP

// ---- launch sequences ---------------------------------------------------------------------------

// where the pieces of the caller's scratch buffer are (lsm_scratch_doubles() doubles)
struct LsmLayout {
  double* rec_stats;  // [rows][nch][3]
  RowStat* rs;        // [rows]
  double* rec_pow;    // [rows][nch][2D+1]
  double* P;          // [rows][2D+1]
  double* recB;       // [rows][nch][D+1]
  double* disc_pow;   // [rows]
  double* sync;       // persistent form: status word, then the record ring (both zeroed per launch)
  double* counters;   // [2]
  uint32_t rows, nch;
  int q;
};

constexpr size_t kSyncDoubles = 2;                                           // uint32 status word, padded to 16 bytes
constexpr size_t kRingDoubles = (size_t)kLsmRing * kLsmMaxResident * 32 * 2; // records of 32 granules of 16 bytes

LsmLayout lsm_layout(double* scratch, uint64_t ntot, uint32_t n_steps, int degree) {
  const size_t rows = (size_t)n_steps + 1, ch = lsm_nch(ntot), nv = 2 * (size_t)degree + 1;
  LsmLayout L{};
  L.sync = scratch;  // at the allocation's start: the per-launch memset covers exactly this block
  L.rec_stats = L.sync + kSyncDoubles + kRingDoubles;
  L.rs = reinterpret_cast<RowStat*>(L.rec_stats + rows * ch * 3);
  L.rec_pow = reinterpret_cast<double*>(L.rs) + rows * 3;  // RowStat = 3 doubles
  L.P = L.rec_pow + rows * ch * nv;
  L.recB = L.P + rows * nv;
  L.disc_pow = L.recB + rows * ch * (degree + 1);
  L.counters = L.disc_pow + rows;
  L.rows = (uint32_t)rows;
  L.nch = (uint32_t)ch;
  L.q = lsm_q(ntot);
  return L;
}

LsmStepArgs lsm_step_args(const LsmLayout& L, const double* grid, uint64_t ntot, uint32_t n_steps,
                          double strike, double cp, double step_discount, int32_t* tau, double* val) {
  LsmStepArgs a{};
  a.grid = grid; a.ntot = ntot; a.strike = strike; a.cp = cp; a.ln_disc = log(step_discount);
  a.n_steps = n_steps; a.tau = tau; a.val = val; a.rs = L.rs; a.P = L.P; a.recB = L.recB;
  a.disc_pow = L.disc_pow; a.counters = L.counters;
  a.n_chunks = L.nch;
  return a;
}

void launch_stats(const LsmLayout& L, const LsmStepArgs& a, hipStream_t s) {
  const dim3 g(L.nch, L.rows), b(kLsmWg);
  with_q(L.q, [&](auto qc) {
    hipLaunchKernelGGL(lsm_stats_kernel<decltype(qc)::value>, g, b, 0, s, a.grid, a.ntot, a.strike, a.cp, L.nch, L.rec_stats);
  });
}

template <int D>
void launch_pow(const LsmLayout& L, const LsmStepArgs& a, hipStream_t s) {
  const dim3 g(L.nch, L.rows), b(kLsmWg);
  with_q(L.q, [&](auto qc) {
    hipLaunchKernelGGL((lsm_pow_kernel<D, decltype(qc)::value>), g, b, 0, s, a.grid, a.ntot, a.strike, a.cp, L.nch, a.rs, L.rec_pow);
  });
}

template <int D>
void launch_init(const LsmLayout& L, const LsmStepArgs& a, hipStream_t s) {
  with_q(L.q, [&](auto qc) {
    hipLaunchKernelGGL((lsm_init_kernel<D, decltype(qc)::value>), dim3(a.n_chunks), dim3(kLsmWg), 0, s, a);
  });
}

template <int D>
void launch_step(const LsmLayout& L, const LsmStepArgs& a, uint32_t t, hipStream_t s) {
  with_q(L.q, [&](auto qc) {
    hipLaunchKernelGGL((lsm_step_kernel<D, decltype(qc)::value>), dim3(a.n_chunks), dim3(kLsmWg), 0, s, a, t);
  });
}

// the whole induction on one device, one launch per date
template <int D>
int run_lsm(const LsmLayout& L, const LsmStepArgs& a, hipStream_t s) {
  launch_stats(L, a, s);
  hipLaunchKernelGGL(lsm_rowstat_kernel, dim3(L.rows), dim3(kLsmWg), 0, s, L.rec_stats, L.nch, L.rs);
  launch_pow<D>(L, a, s);
  hipLaunchKernelGGL(lsm_sum_records_kernel<2 * D + 1>, dim3(L.rows), dim3(kLsmWg), 0, s, L.rec_pow,
                     L.nch, L.P);
  launch_init<D>(L, a, s);
  for (uint32_t t = a.n_steps - 1; t >= 1; --t)  // for i = nsteps:-1:2, t = i-1 (:112-113)
    launch_step<D>(L, a, t, s);
  return (int)hipGetLastError();
}

// The persistent form synchronises its workgroups through memory, so ALL of them must be resident
// at once.  It is therefore launched with hipLaunchCooperativeKernel: the runtime checks the grid
// against what the device can hold (registers, LDS, waves — the kernel's own occupancy) and REFUSES
// the launch (hipErrorCooperativeLaunchTooLarge) instead of starting a grid whose late workgroups
// would never meet the early ones; and a cooperative grid is not dispatched beside other kernels that
// would take its CUs.  On top of that one workgroup per CU is asked for (the induction's workgroups
// are sized for a whole CU: 512 threads x 256 registers, 130 KiB of LDS).  The bounded waits inside
// the kernel stay as the last guard (spin_ticks).
template <class K>
bool grid_fits(K kernel, uint32_t blocks) {
  int dev = 0, cus = 0, per_cu = 0, coop = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kLsmWg, 0) != hipSuccess) return false;
  return per_cu >= 1 && blocks <= (uint32_t)cus;  // counted at ONE workgroup per CU, whatever the query allows
}

template <class K>
int launch_cooperative(K kernel, uint32_t blocks, const LsmPersistArgs& a, hipStream_t s) {
  LsmPersistArgs args = a;
  void* params[] = {&args};
  const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), dim3(blocks),
                                                  dim3(kLsmWg), params, 0, s);
  if (e == hipErrorCooperativeLaunchTooLarge || e == hipErrorLaunchOutOfResources ||
      e == hipErrorInvalidConfiguration) {
    (void)hipGetLastError();  // refused, nothing started: the caller runs the launch-per-date form
    return 1;
  }
  return (int)e;
}

std::atomic<unsigned int> g_lsm_launches{0};  // numbers the persistent launches of the process (granule nonce)

// the whole induction in one launch; 1 = not applicable here (too many chunks for the chip, or the
// runtime cannot guarantee that the grid is resident)
template <int D>
int run_lsm_persistent(const LsmLayout& L, const LsmStepArgs& s_args, hipStream_t s, unsigned long long spin_ticks) {
  if (L.nch > (uint32_t)kLsmMaxResident) return 1;
  LsmPersistArgs a{};
  a.grid = s_args.grid; a.ntot = s_args.ntot; a.strike = s_args.strike; a.cp = s_args.cp;
  a.n_steps = s_args.n_steps; a.n_chunks = L.nch; a.tau = s_args.tau; a.val = s_args.val;
  a.disc_pow = L.disc_pow; a.counters = L.counters; a.ln_disc = s_args.ln_disc;
  a.status = reinterpret_cast<unsigned int*>(L.sync);
  a.rec = reinterpret_cast<unsigned long long*>(L.sync + kSyncDoubles);
  a.spin_ticks = spin_ticks;  // default 1 s of the 100 MHz constant clock
  do a.nonce = ++g_lsm_launches; while (a.nonce == 0);
  const bool fits = with_q(L.q, [&](auto qc) { return grid_fits(lsm_persistent_kernel<D, decltype(qc)::value>, L.nch); });
  if (!fits) return 1;
  // status word and the whole record ring (2 MiB): a granule of an earlier launch must not validate
  hipError_t e = hipMemsetAsync(L.sync, 0, (kSyncDoubles + kRingDoubles) * sizeof(double), s);
  if (e != hipSuccess) return (int)e;
  return with_q(L.q, [&](auto qc) { return launch_cooperative(lsm_persistent_kernel<D, decltype(qc)::value>, L.nch, a, s); });
}

// one phase of the sharded induction (see hh_kernels.h); vec_in / vec_out are device vectors
template <int D>
int run_lsm_phase(const LsmLayout& L, LsmStepArgs a, int phase, uint32_t t, const double* vec_in,
                  double* vec_out, hipStream_t s) {
  constexpr int N = D + 1, NV = 2 * D + 1;
  switch (phase) {
    case kLsmPhasePow:  // global row sums in -> row statistics; local power sums out
      hipLaunchKernelGGL(lsm_rowstat_from_sums_kernel, dim3((L.rows + 255) / 256), dim3(256), 0, s,
                         vec_in, L.rows, L.rs);
      launch_pow<D>(L, a, s);
      hipLaunchKernelGGL(lsm_sum_records_kernel<NV>, dim3(L.rows), dim3(kLsmWg), 0, s, L.rec_pow,
                         L.nch, vec_out);
      break;
    case kLsmPhaseInit: {  // global power sums in; stopping at expiry; local moment sums of row n-1 out
      hipError_t e = hipMemcpyAsync(L.P, vec_in, (size_t)L.rows * NV * sizeof(double),
                                    hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return (int)e;
      launch_init<D>(L, a, s);
      if (a.n_steps >= 2)
        hipLaunchKernelGGL(lsm_sum_records_kernel<N>, dim3(1), dim3(kLsmWg), 0, s,
                           a.recB + (size_t)(a.n_steps - 1) * a.n_chunks * N, a.n_chunks, vec_out);
      break;
    }
    case kLsmPhaseStep:  // global moment sums of row t in; decisions at t; local sums of row t-1 out
      a.B_given = vec_in;
      launch_step<D>(L, a, t, s);
      if (t >= 2)
        hipLaunchKernelGGL(lsm_sum_records_kernel<N>, dim3(1), dim3(kLsmWg), 0, s,
                           a.recB + (size_t)(t - 1) * a.n_chunks * N, a.n_chunks, vec_out);
      break;
    default: return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

}  // namespace

uint32_t lsm_chunks(uint64_t ntot) { return (uint32_t)((ntot + kLsmFinalChunk - 1) / kLsmFinalChunk); }

// scratch sizes in doubles, for the caller (hh_api.hip) to allocate
size_t lsm_scratch_doubles(uint64_t ntot, uint32_t n_steps, int degree) {
  const size_t rows = (size_t)n_steps + 1, ch = lsm_nch(ntot);
  const size_t nv = 2 * (size_t)degree + 1;
  // sync | ring | rec_stats [rows][ch][3] | rowstat [rows][3] | rec_pow [rows][ch][nv] | P [rows][nv] |
  // recB [rows][ch][degree+1] | disc_pow [rows] | counters [2] | phase stamps of a diagnostic build [8]
  return kSyncDoubles + kRingDoubles + rows * ch * 3 + rows * 3 + rows * ch * nv + rows * nv +
         rows * ch * (degree + 1) + rows + 2 + kLsmStampSlots;
}

int launch_gbm_grid(const uint64_t* seeds_dev, uint64_t n_paths, uint32_t n_steps, double S0,
                    double r, double sigma, double T, int anti, double* grid, hipStream_t s) {
  const double dt = T / (double)n_steps;
  const double a = (r - 0.5 * sigma * sigma) * dt, b = sigma * sqrt(dt);
  const dim3 g((unsigned)((n_paths + 255) / 256)), blk(256);
  if (anti)
    hipLaunchKernelGGL(gbm_grid_kernel<true>, g, blk, 0, s, seeds_dev, n_paths, n_steps, S0, a, b,
                       exp(2.0 * a), grid);
  else
    hipLaunchKernelGGL(gbm_grid_kernel<false>, g, blk, 0, s, seeds_dev, n_paths, n_steps, S0, a, b,
                       1.0, grid);
  return (int)hipGetLastError();
}

#define HH_LSM_DISPATCH(degree, CALL)            \
  switch (degree) {                                \
    case 1: rc = CALL(1); break;                   \
    case 2: rc = CALL(2); break;                   \
    case 3: rc = CALL(3); break;                   \
    case 4: rc = CALL(4); break;                   \
    case 5: rc = CALL(5); break;                   \
    case 6: rc = CALL(6); break;                   \
    case 7: rc = CALL(7); break;                   \
    default: rc = CALL(8); break;                  \
  }

// Backward induction on a device-resident grid.  `scratch` has lsm_scratch_doubles() doubles,
// `records` lsm_chunks() x kRecStride; on return `records` holds the per-workgroup Σ, Σ² of the
// discounted stopped values and scratch's last two doubles the regressed / skipped row counts.
// form: kLsmFormPersistent enqueues the one-launch form when the ensemble fits it (*form_used tells;
// the caller must then check lsm_persistent_status after synchronising) — kLsmFormPerDate never does.
int launch_lsm(const double* grid, uint64_t ntot, uint32_t n_steps, double strike, double cp,
               double step_discount, int degree, int32_t* tau, double* val, double* scratch,
               double* records, hipStream_t s, int form, int* form_used, unsigned long long spin_ticks) {
  if (degree < 1 || degree > kLsmMaxDeg) return (int)hipErrorInvalidValue;
  const LsmLayout L = lsm_layout(scratch, ntot, n_steps, degree);
  const dim3 b(256);
  const LsmStepArgs a = lsm_step_args(L, grid, ntot, n_steps, strike, cp, step_discount, tau, val);
  // the discount table in memory: the launch-per-date form reads it, the one-launch form only when it
  // does not fit its LDS copy (which it forms itself)
  bool table_out = false;
  auto table = [&]() {
    if (!table_out)
      hipLaunchKernelGGL(lsm_disc_kernel, dim3((L.rows + 255) / 256), b, 0, s, a.ln_disc, n_steps, L.disc_pow);
    table_out = true;
  };
  int rc = 1;
  if (form == kLsmFormPersistent || form == kLsmFormAuto) {  // one launch whenever the chip can hold the grid
    if (n_steps >= (uint32_t)kDiscLds) table();
    if (HH_LSM_STAMPS) {  // (the shipped kernel writes both counters itself; the stamps accumulate)
      const hipError_t e = hipMemsetAsync(L.counters, 0, (2 + kLsmStampSlots) * sizeof(double), s);
      if (e != hipSuccess) return (int)e;
    }
#define HH_CALL(D) run_lsm_persistent<D>(L, a, s, spin_ticks)
    HH_LSM_DISPATCH(degree, HH_CALL)
#undef HH_CALL
    if (rc != 0 && rc != 1) return rc;
  }
  if (form_used) *form_used = rc == 0 ? kLsmFormPersistent : kLsmFormPerDate;
  if (rc == 1) {
    const hipError_t e = hipMemsetAsync(L.counters, 0, (2 + kLsmStampSlots) * sizeof(double), s);
    if (e != hipSuccess) return (int)e;
    table();
#define HH_CALL(D) run_lsm<D>(L, a, s)
    HH_LSM_DISPATCH(degree, HH_CALL)
#undef HH_CALL
    if (rc) return rc;
  }
  hipLaunchKernelGGL(lsm_final_kernel, dim3(lsm_chunks(ntot)), b, 0, s, tau, val, ntot, a.ln_disc,
                     records);
  return (int)hipGetLastError();
}

// device address of the word the persistent form sets when a workgroup gave up waiting
const unsigned int* lsm_persistent_status(const double* scratch) {
  return reinterpret_cast<const unsigned int*>(scratch);
}

// Sharded induction, one phase per call (hh_kernels.h).  kLsmPhaseStats: local row sums
// (n, Σx, Σx² of the in-the-money spots of every row) -> vec_out[rows][3]; also the discount table and
// the counters.  kLsmPhaseFinal: per-workgroup Σ, Σ² of the discounted stopped values -> records.
int launch_lsm_phase(int phase, uint32_t t, const double* grid, uint64_t ntot, uint32_t n_steps,
                     double strike, double cp, double step_discount, int degree, int32_t* tau,
                     double* val, double* scratch, double* records, const double* vec_in,
                     double* vec_out, hipStream_t s) {
  if (degree < 1 || degree > kLsmMaxDeg) return (int)hipErrorInvalidValue;
  const LsmLayout L = lsm_layout(scratch, ntot, n_steps, degree);
  const LsmStepArgs a = lsm_step_args(L, grid, ntot, n_steps, strike, cp, step_discount, tau, val);
  const dim3 b(256);
  if (phase == kLsmPhaseStats) {
    hipError_t e = hipMemsetAsync(L.counters, 0, (2 + kLsmStampSlots) * sizeof(double), s);
    if (e != hipSuccess) return (int)e;
    launch_stats(L, a, s);
    hipLaunchKernelGGL(lsm_sum_records_kernel<3>, dim3(L.rows), dim3(kLsmWg), 0, s, L.rec_stats, L.nch,
                       vec_out);
    hipLaunchKernelGGL(lsm_disc_kernel, dim3((L.rows + 255) / 256), b, 0, s, a.ln_disc, n_steps,
                       L.disc_pow);
    return (int)hipGetLastError();
  }
  if (phase == kLsmPhaseFinal) {
    hipLaunchKernelGGL(lsm_final_kernel, dim3(lsm_chunks(ntot)), b, 0, s, tau, val, ntot, a.ln_disc,
                       records);
    return (int)hipGetLastError();
  }
  int rc = 0;
#define HH_CALL(D) run_lsm_phase<D>(L, a, phase, t, vec_in, vec_out, s)
  HH_LSM_DISPATCH(degree, HH_CALL)
#undef HH_CALL
  return rc;
}

}  // namespace hh
