// Several models on the SAME draws in one pass — the simulation behind FiniteDifference and second-order
// Greeks (greeks_problem.jl:279-303, 318-329, 360-422).
//
// The reference computes a bumped Greek as 2 (forward / backward / central), 3 (second order, one lens) or 4
// (cross) FULL solves of problems that differ in one or two numbers, all with the seeds of the same
// SimulationConfig — common random numbers.  Run one after the other, every solve draws the same normals
// again (GENERATE: 121 of the 145 instructions of a path-step make them) or streams the same
// increments from HBM again (REPLAY: 16 bytes per path-step).  Here a lane keeps K states — one per model —
// and steps all of them on each pair of increments, exactly as the antithetic kernel steps a trajectory and
// its mirror on one: the draws (or the bytes) are paid once for K solves.  Each model has its own argument
// block, record set and accumulator, and everything that touches a model's numbers is the one-model kernels'
// code (hh_sim.h) in the one-model kernels' order, so model k's accumulator vector is the one hh_mc_accumulate
// gives for it, bit for bit.
#include <type_traits>

#include "hh_sim.h"

namespace hh {

template <int K>
struct MultiArgs {
  SimArgs<0> m[K];  // everything but the model scalars and the output pointers is the same in all of them
};

#ifndef HH_MULTI_CHUNK
#define HH_MULTI_CHUNK 4          // steps per register chunk of the REPLAY pipeline (as euler_kernel)
#endif
#ifndef HH_MULTI_MAXW
#define HH_MULTI_MAXW 3           // waves per SIMD, REPLAY: K path-steps of arithmetic per 16 bytes want more
#endif                            //   waves than the price-only stream's two

// SHARED: the models agree on ρ and dt (what a bump of the spot, the variance, κ, θ, σ, the rate or the strike
// leaves alone), so the correlated increments are formed once; otherwise each model forms its own from the
// same two normals.  Same operations on the same numbers either way.
template <class M, bool REPLAY, bool ANTI, int K, bool SHARED>
__global__ __launch_bounds__(kTile)
__attribute__((amdgpu_waves_per_eu(1, REPLAY ? HH_MULTI_MAXW : 8))) void euler_multi_kernel(const MultiArgs<K> a) {
  constexpr int NC = M::NCOMP;
  using State = typename M::State;
  const SimArgs<0>& a0 = a.m[0];
  const uint32_t tile = blockIdx.x, tid = threadIdx.x;
  const uint64_t path = (uint64_t)tile * kTile + tid;
  const uint32_t n_steps = a0.n_steps;

  State st[K];
  State sa[ANTI ? K : 1];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    M::init(st[k], a.m[k]);
    if constexpr (ANTI) M::init(sa[k], a.m[k]);
  }
  auto step_all = [&](double d1, double d2) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      M::step(st[k], a.m[k], d1, d2);
      if constexpr (ANTI) M::step(sa[k], a.m[k], -d1, -d2);  // montecarlo.jl:258: -W
    }
  };

  if constexpr (REPLAY) {
    // the register pipeline of euler_kernel: two chunks, load(Y) || compute(X), steady state unguarded so
    // that the compiler counts its waits
    const double* __restrict__ base = a0.replay + (size_t)tile * n_steps * NC * kTile + tid;
    constexpr int CH = HH_MULTI_CHUNK;
    double X[CH][NC], Y[CH][NC];
    auto ld = [&](double(&buf)[CH][NC], uint32_t s0) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        if (s0 + u < n_steps) {
#pragma unroll
          for (int c = 0; c < NC; ++c) buf[u][c] = stream_load<double>(base + ((size_t)(s0 + u) * NC + c) * kTile);
        }
      }
    };
    auto go = [&](const double(&buf)[CH][NC], uint32_t s0) {
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (s0 + u < n_steps) step_all(buf[u][0], NC > 1 ? buf[u][NC - 1] : 0.0);
    };
    auto ldf = [&](double(&buf)[CH][NC], uint32_t s0) {
#pragma unroll
      for (int u = 0; u < CH; ++u)
#pragma unroll
        for (int c = 0; c < NC; ++c) buf[u][c] = stream_load<double>(base + ((size_t)(s0 + u) * NC + c) * kTile);
    };
    auto gof = [&](const double(&buf)[CH][NC]) {
#pragma unroll
      for (int u = 0; u < CH; ++u) step_all(buf[u][0], NC > 1 ? buf[u][NC - 1] : 0.0);
    };
    uint32_t s = 0;
    if (n_steps >= 2u * CH) {
      ldf(X, 0);
      while (s + 3u * CH <= n_steps) {  // chunks s, s+CH and s+2CH are full
        ldf(Y, s + CH);
        gof(X);
        ldf(X, s + 2u * CH);
        gof(Y);
        s += 2u * CH;
      }
      ld(Y, s + CH);  // fewer than 2·CH steps beyond chunk X: guarded
      gof(X);
      s += CH;
      ld(X, s + CH);
      go(Y, s);
      go(X, s + CH);
    } else {
      ld(X, 0);
      while (s < n_steps) {
        ld(Y, s + CH);
        go(X, s);
        s += CH;
        if (s >= n_steps) break;
        ld(X, s + CH);
        go(Y, s);
        s += CH;
      }
    }
  } else {
    const uint64_t key = path < a0.n_paths ? a0.seeds[path] : 0ull;
    if constexpr (NC == 2) {
      for (uint32_t s = 0; s < n_steps; ++s) {
        double z1, z2;
        normal_pair(key, s, 0u, 0u, kDomEuler, z1, z2);
        if constexpr (SHARED) {
          step_all(a0.sqrt_dt * z1, a0.sqrt_dt * fma(a0.rho, z1, a0.rho_c * z2));
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const double d1 = a.m[k].sqrt_dt * z1;
            const double d2 = a.m[k].sqrt_dt * fma(a.m[k].rho, z1, a.m[k].rho_c * z2);
            M::step(st[k], a.m[k], d1, d2);
            if constexpr (ANTI) M::step(sa[k], a.m[k], -d1, -d2);
          }
        }
      }
    } else {
      // scalar noise: one Philox block feeds two consecutive steps (as euler_kernel)
      auto one = [&](double z) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const double d = a.m[k].sqrt_dt * z;
          M::step(st[k], a.m[k], d, 0.0);
          if constexpr (ANTI) M::step(sa[k], a.m[k], -d, 0.0);
        }
      };
      for (uint32_t s = 0; s < n_steps; s += 2) {
        double z1, z2;
        normal_pair(key, s >> 1, 0u, 0u, kDomEuler, z1, z2);
        one(z1);
        if (s + 1 < n_steps) one(z2);
      }
    }
  }

#pragma unroll
  for (int k = 0; k < K; ++k) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    finish_path<0, ANTI>(st[k], sa[ANTI ? k : 0], a.m[k], path, acc);
    if (k) __syncthreads();  // the reduction's LDS staging is reused
    block_reduce_publish<4, kTile / 64, 2>(acc, a.m[k].records + (size_t)tile * kRecStride, a.m[k].accum != nullptr, false);
  }
  if (a0.accum && reduces_records(tile, a0)) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k) __syncthreads();
      finish_records<kTile, 0>(a.m[k]);
    }
  }
}

// the exact lognormal law (exact_gbm_kernel) for K models on the same normals
template <bool REPLAY, bool ANTI, int K, int PAIRS>
__global__ __launch_bounds__(kTile) void exact_multi_kernel(const MultiArgs<K> a) {
  const SimArgs<0>& a0 = a.m[0];
  const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
  struct S {
    DualT<0> x;
  };
  double acc[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[k][i] = 0.0;
#pragma unroll 2
  for (int jp = 0; jp < PAIRS; ++jp) {
    const uint64_t path0 = ((uint64_t)chunk * PAIRS + jp) * (2 * kTile) + (uint64_t)tid * 2;
    double z[2];
    exact_pair_normals<REPLAY>(a0, path0, z);
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        S st, sa;
        st.x.v = fma(a.m[k].law_sd.v, z[j], a.m[k].law_mu.v);
        if constexpr (ANTI) sa.x.v = 2 * a.m[k].law_mu.v - st.x.v;  // montecarlo.jl:387
        finish_path<0, ANTI>(st, sa, a.m[k], path0 + j, acc[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k) __syncthreads();  // the reduction's LDS staging is reused
    block_reduce_publish<4, kTile / 64, 2>(acc[k], a.m[k].records + (size_t)chunk * kRecStride, a.m[k].accum != nullptr, false);
  }
  if (a0.accum && reduces_records(chunk, a0)) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k) __syncthreads();
      finish_records<kTile, 0>(a.m[k]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

template <class M, bool REPLAY, bool ANTI, int K>
static int launch_multi_k(const MultiArgs<K>& a, bool shared, hipStream_t s) {
  const dim3 g(a.m[0].n_tiles), b(kTile);
  // the increments of a REPLAY run are the caller's: nothing to share or not to share; one noise component
  // costs one multiply per model either way
  if constexpr (!REPLAY && M::NCOMP == 2) {
    if (shared) {
      hipLaunchKernelGGL((euler_multi_kernel<M, REPLAY, ANTI, K, true>), g, b, 0, s, a);
      return (int)hipGetLastError();
    }
  }
  hipLaunchKernelGGL((euler_multi_kernel<M, REPLAY, ANTI, K, false>), g, b, 0, s, a);
  return (int)hipGetLastError();
}

template <class M, int K>
static int launch_multi_m(const MultiArgs<K>& a, bool replay, bool anti, bool shared, hipStream_t s) {
  if (replay) return anti ? launch_multi_k<M, true, true, K>(a, shared, s) : launch_multi_k<M, true, false, K>(a, shared, s);
  return anti ? launch_multi_k<M, false, true, K>(a, shared, s) : launch_multi_k<M, false, false, K>(a, shared, s);
}

template <int K>
static int launch_multi(const hh_model* models, const hh_config& c, const DevicePtrs* p, hipStream_t s) {
  MultiArgs<K> a;
  bool shared = true;
  for (int k = 0; k < K; ++k) {
    a.m[k] = make_args0(models[k], c, p[k]);
    shared = shared && a.m[k].rho == a.m[0].rho && a.m[k].dt == a.m[0].dt;
  }
  const bool anti = c.antithetic != 0, replay = c.noise_mode == HH_NOISE_REPLAY;
  if (c.strategy == HH_EXACT_LAW) {
    const dim3 g(a.m[0].n_tiles), b(kTile);
    auto pick = [&](auto replay_c, auto anti_c) {
      constexpr bool R = decltype(replay_c)::value, A = decltype(anti_c)::value;
      const int pairs = exact_pairs_per_lane(c.n_paths);
      return pairs == kExactPairsHuge ? exact_multi_kernel<R, A, K, kExactPairsHuge>
             : pairs == kExactPairs   ? exact_multi_kernel<R, A, K, kExactPairs>
             : pairs == 4             ? exact_multi_kernel<R, A, K, 4>
             : pairs == 2             ? exact_multi_kernel<R, A, K, 2>
                                      : exact_multi_kernel<R, A, K, 1>;
    };
    using T = std::true_type;
    using F = std::false_type;
    auto kernel = replay ? (anti ? pick(T{}, T{}) : pick(T{}, F{})) : (anti ? pick(F{}, T{}) : pick(F{}, F{}));
    hipLaunchKernelGGL(kernel, g, b, 0, s, a);
    return (int)hipGetLastError();
  }
  if (c.dynamics == HH_LOGNORMAL) return launch_multi_m<GbmModel<0>, K>(a, replay, anti, shared, s);
  if (c.em_split) return launch_multi_m<HestonModel<0, true>, K>(a, replay, anti, shared, s);
  return launch_multi_m<HestonModel<0, false>, K>(a, replay, anti, shared, s);
}

int launch_simulation_multi(const hh_model* models, int n_models, const hh_config& c, const DevicePtrs* p,
                            hipStream_t s) {
  switch (n_models) {
    case 2: return launch_multi<2>(models, c, p, s);
    case 3: return launch_multi<3>(models, c, p, s);
    case 4: return launch_multi<4>(models, c, p, s);
    default: return (int)hipErrorInvalidValue;
  }
}

}  // namespace hh
