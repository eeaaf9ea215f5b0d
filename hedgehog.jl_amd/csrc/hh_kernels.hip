// Path-simulation kernels for gfx950 (wave64, fp64 VALU; no MFMA — the path has no contraction).
//
// What they replace, per trajectory, is the body of StochasticDiffEq.solve(EnsembleProblem, EM();
// dt, trajectories) as driven by montecarlo.jl:342-375, with the drift/diffusion closures of
// heston.jl:7-52, followed by final_sample (montecarlo.jl:398), the payoff functor
// (payoffs.jl:154-156), the antithetic pair average (montecarlo.jl:430-432) and the Σ that feeds
// mean() at montecarlo.jl:490 — fused into one pass that keeps the whole state in registers.
//
// Work decomposition: one workgroup = one tile of 256 trajectories; the time axis is a serial
// recurrence inside a thread.  REPLAY streams the tile's increments dW[tile][step][comp][256]
// from HBM as one contiguous run per workgroup with 16-byte loads per lane and a two-chunk
// register pipeline; GENERATE draws them in registers from Philox keyed by the trajectory seed.
// Every workgroup leaves one record of partial sums; a second tiny kernel adds the records in a
// fixed order, so results are bit-reproducible for a given (n_paths, sharding).
#include <cstdlib>
#include <type_traits>

#include "hh_sim.h"

namespace hh {


// ------------------------------------------------------------------------------------------
// Euler–Maruyama kernel
// ------------------------------------------------------------------------------------------

// The shape of the REPLAY pipeline, as measured (profiles/r02_a_replay_occupancy_ab.txt,
// r02_b_replay_shape_ab.txt, r02_d_replay_counted_waits_ab.txt).  The A/B builds of tools/tune_replay.py
// (-DHH_REPLAY_VARIANTS -Itools/variants) override these with -D and add the experiment paths that did not
// ship — two trajectories per lane, the per-wave LDS-DMA ring, the occupancy pad, the antithetic pair
// split over two lanes — from tools/variants/; this translation unit holds what ships.
#ifdef HH_REPLAY_VARIANTS
#include "replay_knobs.h"
#endif
#ifndef HH_REPLAY_CHUNK_PRICE
#define HH_REPLAY_CHUNK_PRICE 4   // steps per register chunk, price-only kernel …
#endif
#ifndef HH_REPLAY_CHUNK_PPT1
#define HH_REPLAY_CHUNK_PPT1 4    // … antithetic and dual-partial kernels (one trajectory per lane: 0.608 vs 0.627 ms
#endif                            //   antithetic, 0.597 vs 0.611 ms one carried derivative, against two per lane)
#ifndef HH_REPLAY_CHUNK_TAIL
#define HH_REPLAY_CHUNK_TAIL 4    // … and of the price-only kernel's last HH_REPLAY_TAIL_TILES workgroups
#endif
#ifndef HH_REPLAY_TAIL_TILES
#define HH_REPLAY_TAIL_TILES 512
#endif
#ifndef HH_REPLAY_COUNTED
#define HH_REPLAY_COUNTED 1       // steady-state loads unguarded, so that the compiler can COUNT its waits
#endif
// Occupancy of the REPLAY kernels.  HBM delivers most when a CU runs few concurrent 1 MB streams
// (tools/ubench/hbm_read_sweep.hip), so the REPLAY variants are held BELOW what their register count
// would allow — by the compiler's own occupancy control, amdgpu_waves_per_eu(1, max): the kernel
// descriptor then reserves ⌊512 / max⌋ registers per lane and the hardware admits at most `max`
// waves per SIMD, whatever else the kernel declares.
#ifndef HH_REPLAY_MINW
#define HH_REPLAY_MINW 1
#endif
#ifndef HH_REPLAY_MAXW
#define HH_REPLAY_MAXW 2          // price-only: 2 waves per SIMD = 2 workgroups of 256 threads = 8 waves per CU
#endif
#ifndef HH_REPLAY_MAXW_DUAL
#define HH_REPLAY_MAXW_DUAL 3     // one carried derivative: 0.597 ms at 3 waves per SIMD against 0.646 at 2
#endif
#ifndef HH_REPLAY_MAXW_DUAL_WIDE
#define HH_REPLAY_MAXW_DUAL_WIDE HH_REPLAY_MAXW_DUAL  // two or more carried derivatives
#endif
#ifndef HH_REPLAY_MAXW_ANTI
#define HH_REPLAY_MAXW_ANTI 8     // antithetic: indifferent (0.629 vs 0.625 ms), left at the register limit
#endif

constexpr int replay_max_waves(bool replay, bool anti, int p) {
  return !replay ? 8 : anti ? HH_REPLAY_MAXW_ANTI : p > 1 ? HH_REPLAY_MAXW_DUAL_WIDE : p > 0 ? HH_REPLAY_MAXW_DUAL : HH_REPLAY_MAXW;
}
template <class M, int P, bool REPLAY, bool ANTI, int PPT, int RING, bool PIPE>
__global__ __launch_bounds__(kTile / PPT)
__attribute__((amdgpu_waves_per_eu(REPLAY ? HH_REPLAY_MINW : 1,
                                   replay_max_waves(REPLAY, ANTI, P)))) void euler_kernel(
    const SimArgs<P> a) {
  constexpr int NC = M::NCOMP;
  using State = typename M::State;
  using Vec = typename VecOf<PPT>::type;

  const uint32_t tile = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint64_t path0 = (uint64_t)tile * kTile + (uint64_t)tid * PPT;
  const uint32_t n_steps = a.n_steps;

  State st[PPT];
  State sa[ANTI ? PPT : 1];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    M::init(st[j], a);
    if constexpr (ANTI) M::init(sa[j], a);
  }

  if constexpr (REPLAY) {
    // the tile's increments: element (step, comp, lane) at ((step*NC + comp)*256 + lane)
    const double* __restrict__ base =
        a.replay + (size_t)tile * n_steps * NC * kTile + (size_t)tid * PPT;
    // steps per chunk: the price-only kernel (one trajectory per lane) moves 4 steps at a time
#ifdef HH_REPLAY_VARIANTS
    constexpr int kChunk = (P == 0 && !ANTI && RING == 0) ? HH_REPLAY_CHUNK_PRICE : (PPT == 1 ? HH_REPLAY_CHUNK_PPT1 : HH_REPLAY_CHUNK);
#else
    constexpr int kChunk = (P == 0 && !ANTI) ? HH_REPLAY_CHUNK_PRICE : HH_REPLAY_CHUNK_PPT1;
#endif
    // Two register chunks, load(B) || compute(A), with the occupancy capped by amdgpu_waves_per_eu (see
    // HH_REPLAY_MAXW above): 8 waves per CU for the price-only kernel, 12 for the dual-partial kernels,
    // uncapped for the antithetic one (measurements: DESIGN.md §5, tools/tune_replay.py,
    // tools/replay_sizes.py, tools/ubench/hbm_read_sweep.hip).
#ifdef HH_REPLAY_VARIANTS
#include "replay_lds_ring.inc"  // if constexpr (RING > 0 && PPT == 2) { the per-wave LDS-DMA ring } else
#endif
    {
#ifdef HH_REPLAY_VARIANTS
#include "replay_occupancy_pad.inc"
#endif
    // The last workgroups of a grid run while the chip empties: fewer streams are open, so each has
    // to keep more bytes in flight to hold the bandwidth up — they pipeline HH_REPLAY_CHUNK_TAIL steps
    // per chunk instead of kChunk (registers are there: the occupancy cap leaves a wave 256).
    auto pipeline = [&](auto depth) {
      constexpr int CH = decltype(depth)::value;
      Vec X[CH][NC], Y[CH][NC];
      auto ld = [&](Vec(&buf)[CH][NC], uint32_t s0) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (s0 + u < n_steps) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
              buf[u][c] = stream_load<Vec>(base + ((size_t)(s0 + u) * NC + c) * kTile);
          }
        }
      };
      auto go = [&](const Vec(&buf)[CH][NC], uint32_t s0) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (s0 + u < n_steps) {
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
              const double d1 = VecOf<PPT>::get(buf[u][0], j);
              const double d2 = NC > 1 ? VecOf<PPT>::get(buf[u][NC - 1], j) : 0.0;
              M::step(st[j], a, d1, d2);
              if constexpr (ANTI) M::step(sa[j], a, -d1, -d2);  // montecarlo.jl:258: -W
            }
          }
        }
      };
      // Steady state on FULL chunks, loads unguarded: with a load under `if (step < n_steps)` the
      // compiler cannot count how many younger loads are certainly in flight when chunk X is consumed,
      // and waits for ALL of them (s_waitcnt vmcnt(0)/(1) right after issuing Y) — a wave then never
      // overlaps its own loads with its own arithmetic.  Unguarded, the wait is vmcnt(CH·NC): X has
      // landed, Y stays in flight behind the arithmetic of X.
      auto ldf = [&](Vec(&buf)[CH][NC], uint32_t s0) {
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
          for (int c = 0; c < NC; ++c)
            buf[u][c] = stream_load<Vec>(base + ((size_t)(s0 + u) * NC + c) * kTile);
      };
      auto gof = [&](const Vec(&buf)[CH][NC]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
#pragma unroll
          for (int j = 0; j < PPT; ++j) {
            const double d1 = VecOf<PPT>::get(buf[u][0], j);
            const double d2 = NC > 1 ? VecOf<PPT>::get(buf[u][NC - 1], j) : 0.0;
            M::step(st[j], a, d1, d2);
            if constexpr (ANTI) M::step(sa[j], a, -d1, -d2);  // montecarlo.jl:258: -W
          }
        }
      };
      uint32_t s = 0;
      if (HH_REPLAY_COUNTED && n_steps >= 2u * CH) {
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
      } else {  // short runs (and -DHH_REPLAY_COUNTED=0, the all-guarded form, for A/B)
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
    };
    constexpr int kTail = (P == 0 && !ANTI && PPT == 1) ? HH_REPLAY_CHUNK_TAIL : kChunk;
    if (kTail != kChunk && tile >= a.tail_from)
      pipeline(std::integral_constant<int, kTail>{});
    else
      pipeline(std::integral_constant<int, kChunk>{});
    }
  } else {
    uint64_t key[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) key[j] = (path0 + j < a.n_paths) ? a.seeds[path0 + j] : 0ull;

    if constexpr (NC == 2) {
      for (uint32_t s = 0; s < n_steps; ++s) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
          double z1, z2;
          normal_pair(key[j], s, 0u, 0u, kDomEuler, z1, z2);
          const double d1 = a.sqrt_dt * z1;
          const double d2 = a.sqrt_dt * fma(a.rho, z1, a.rho_c * z2);
          M::step(st[j], a, d1, d2);
          if constexpr (ANTI) M::step(sa[j], a, -d1, -d2);
        }
      }
    } else {
      // scalar noise: one Philox block feeds two consecutive steps
      for (uint32_t s = 0; s < n_steps; s += 2) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
          double z1, z2;
          normal_pair(key[j], s >> 1, 0u, 0u, kDomEuler, z1, z2);
          const double d1 = a.sqrt_dt * z1;
          M::step(st[j], a, d1, 0.0);
          if constexpr (ANTI) M::step(sa[j], a, -d1, 0.0);
          if (s + 1 < n_steps) {
            const double d2 = a.sqrt_dt * z2;
            M::step(st[j], a, d2, 0.0);
            if constexpr (ANTI) M::step(sa[j], a, -d2, 0.0);
          }
        }
      }
    }
  }

  double acc[4 + P];
#pragma unroll
  for (int i = 0; i < 4 + P; ++i) acc[i] = 0.0;
#pragma unroll
  for (int j = 0; j < PPT; ++j) finish_path<P, ANTI>(st[j], sa[ANTI ? j : 0], a, path0 + j, acc);
  block_reduce_publish<4 + P, kTile / PPT / 64, 2>(acc, a.records + (size_t)tile * kRecStride, a.accum != nullptr, a.map.n > 0);
  if (a.accum && reduces_records(tile, a)) finish_records<kTile / PPT, P>(a);
}

#ifdef HH_REPLAY_VARIANTS
#include "anti_pair_split.inc"  // euler_pair_split_kernel (-DHH_ANTI_SPLIT=1)
#endif

// ------------------------------------------------------------------------------------------
// Euler–Maruyama on the REFERENCE's noise layout: path-major REPLAY
// ------------------------------------------------------------------------------------------
//
// The reference keeps the noise per trajectory (W.W of each solution, montecarlo.jl:258,370), i.e.
// dW[path][step][comp]: a trajectory's increments are one contiguous row of S = n_steps·NCOMP·8
// bytes.  This kernel consumes that layout directly (no repack pass): the row is cut along the
// ABSOLUTE 128-byte lines of memory, a wave owns 64 consecutive rows and moves, per chunk, one line
// of each of them — 8 rows x 128 B per LDS-DMA instruction (global_load_lds_dwordx4: per-lane source
// address, wave-linear destination), 8 instructions per chunk — into its private 8 KiB LDS image
// [row][8 pieces of 16 B], from which every lane reads its own row back with 8 ds_read_b128 and steps
// through it.  Every line of the buffer is requested exactly once, whole and aligned, by one
// wave-instruction (HBM traffic = algorithmic bytes; nothing relies on a cache keeping a half-used
// line); rows that do not start on a line boundary simply run with a per-lane phase o ∈ 0..7 pieces:
// piece j of chunk k is row piece t = 8k − o + j (Heston: step t; lognormal: steps 2t, 2t+1), outside
// 0 ≤ t < T only in the first and the last two chunks, which run a guarded copy of the loop body.
//
// The LDS image is XOR-swizzled — piece j of row r sits in slot j ^ ((r >> 1) & 7) — so that the 16
// lanes ds_read_b128 serves together (MI355X_MICROARCH.md, LDS table) cover all 64 banks; LDS-DMA
// writes lane-linearly, so the swizzle is applied to the SOURCE address (guide rule 21).  The image is
// private to its wave: no workgroup barrier anywhere, the only ordering needed is the wave's own
// vmcnt(0) before it reads (microarch guide, item 7) and lgkmcnt(0) before the next chunk overwrites.
// While the wave steps through the 32 registers of chunk k, the DMA of chunk k+1 is in flight.
//
// Needs S to be a multiple of 16 bytes (n_steps·NCOMP even: every Heston shape); the one remaining
// case (lognormal, odd n_steps) goes through replay_pack_kernel + the tile-major kernel.
#ifndef HH_PM_MAXW
#define HH_PM_MAXW 2   // waves per SIMD of the path-major kernel: 0.622 ms at 2, 0.632-0.634 at 3, 4, 5 (profiles/r03_a_path_major_ab.txt)
#endif
#ifndef HH_PM_NT
#define HH_PM_NT 1
#endif
#ifndef HH_PM_PIECES
#define HH_PM_PIECES 8  // 16-byte pieces per row and chunk: 8 = one 128-byte line, 16 = two
#endif
constexpr int kPmPieces = HH_PM_PIECES;
static_assert(kPmPieces == 8 || kPmPieces == 16, "one or two 128-byte lines per row and chunk");

template <class M, int P, bool ANTI>
__global__ __launch_bounds__(kTile)
__attribute__((amdgpu_waves_per_eu(1, HH_PM_MAXW))) void euler_pm_kernel(const SimArgs<P> a) {
  constexpr int NC = M::NCOMP;
  constexpr int NP = kPmPieces;            // pieces per row and chunk
  constexpr int ROWB = NP * 16;            // bytes of a row of the LDS image
  constexpr int RPI = 64 / NP;             // rows one LDS-DMA instruction (64 lanes x 16 B) covers
  constexpr int NI = 64 / RPI;             // instructions per chunk
  using State = typename M::State;
  using V2 = double __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(128))) double image[kTile / 64][64 * 2 * NP];

  const uint32_t tile = blockIdx.x, tid = threadIdx.x;
  const uint32_t wave = tid >> 6, lane = tid & 63;
  const uint64_t path = (uint64_t)tile * kTile + tid;
  const uint64_t last = a.n_paths - 1;
  const int T = (int)((uint64_t)a.n_steps * NC / 2);  // 16-byte pieces per row (<= 262 140)
  const uint64_t S = (uint64_t)T * 16;                  // bytes per row
  const char* base = reinterpret_cast<const char*>(a.replay);

  // rows beyond the ensemble (last tile) repeat the last trajectory: loads stay unguarded and in
  // bounds, finish_path drops the duplicates
  const uint64_t wave_path0 = (uint64_t)tile * kTile + (uint64_t)wave * 64;
  auto row_of = [&](uint32_t r) {
    const uint64_t pth = wave_path0 + r;
    return base + (pth < last ? pth : last) * S;
  };
  // slot of piece j in row r of the image: the 16 lanes one ds_read_b128 cycle serves
  // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) must land on 16 different 16-byte bank groups
  auto swz = [](uint32_t r) { return NP == 8 ? (r >> 1) & 7u : r & 15u; };

  // ---- loader role: instruction i moves rows RPI·i + lane / NP, this lane the piece in slot lane % NP
  const char* src[NI];   // source of the CURRENT chunk's piece (advanced by ROWB per chunk)
  const char* row_r[NI]; // the row's first byte
  int tb[NI];            // row piece index of that slot in chunk 0
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const uint32_t r = (uint32_t)RPI * i + lane / NP;
    const char* row = row_of(r);
    const uint32_t o_r = ((uint32_t)(uintptr_t)row & 127u) >> 4;
    const uint32_t jg = (lane % NP) ^ swz(r);  // swizzle on the source side (LDS-DMA writes lane-linearly)
    row_r[i] = row;
    tb[i] = (int)jg - (int)o_r;
    src[i] = row - ((uintptr_t)row & 127u) + jg * 16u;
  }
  // ---- consumer role: this lane's own row
  const char* my_row = row_of(lane);
  const int o = (int)(((uint32_t)(uintptr_t)my_row & 127u) >> 4);
  const uint32_t sw = swz(lane);
  const char* my_img = reinterpret_cast<const char*>(&image[wave][0]) + lane * (uint32_t)ROWB;

  // chunks this wave runs: enough for its row with the largest phase (wave-uniform)
  int om = o;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) om = max(om, __shfl_xor(om, off, 64));
  om = __builtin_amdgcn_readfirstlane(om);
  const int n_chunks = (T + om + NP - 1) / NP;
  // chunks all of whose possible row pieces NP·k − 7 … NP·k + NP − 1 lie inside the row: no guards
  const int k_full_end = (T - NP) / NP;  // full: 1 <= k <= k_full_end (none when T < 2·NP)

  auto issue = [&](int k, bool guarded) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const char* g = src[i];
      if (guarded) {
        // a piece outside the row is redirected to the row's NEAREST piece: inside the same half
        // line as the valid pieces next to it, so a partly valid line costs no second request and a
        // line shared by two rows is fetched by halves
        int t = NP * k + tb[i];
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        g = row_r[i] + (uint32_t)t * 16u;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)&image[wave][i * 128],
                                       16, 0, HH_PM_NT ? 2 : 0);
      src[i] += ROWB;
    }
  };

  State st, sa;
  M::init(st, a);
  if constexpr (ANTI) M::init(sa, a);

  auto advance = [&](const V2& d) {
    if constexpr (NC == 2) {
      M::step(st, a, d.x, d.y);
      if constexpr (ANTI) M::step(sa, a, -d.x, -d.y);  // montecarlo.jl:258: -W
    } else {
      M::step(st, a, d.x, 0.0);
      if constexpr (ANTI) M::step(sa, a, -d.x, 0.0);
      M::step(st, a, d.y, 0.0);
      if constexpr (ANTI) M::step(sa, a, -d.y, 0.0);
    }
  };

  if (n_chunks > 0) issue(0, true);
  for (int k = 0; k < n_chunks; ++k) {
    // the compiler orders LDS reads behind every outstanding LDS-DMA (s_waitcnt vmcnt(0))
    V2 d[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j)
      d[j] = *reinterpret_cast<const V2*>(my_img + (((uint32_t)j ^ sw) << 4));
    // WAR: the next chunk's DMA overwrites the image — not before these reads have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const bool full = k >= 1 && k <= k_full_end;
    if (k + 1 < n_chunks) issue(k + 1, !(k + 1 <= k_full_end));
    if (full) {
#pragma unroll
      for (int j = 0; j < NP; ++j) advance(d[j]);
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int t = NP * k - o + j;
        if (t >= 0 && t < T) advance(d[j]);
      }
    }
  }

  double acc[4 + P];
#pragma unroll
  for (int i = 0; i < 4 + P; ++i) acc[i] = 0.0;
  finish_path<P, ANTI>(st, sa, a, path, acc);
  block_reduce_publish<4 + P, kTile / 64, 2>(acc, a.records + (size_t)tile * kRecStride, a.accum != nullptr, a.map.n > 0);
  if (a.accum && reduces_records(tile, a)) finish_records<kTile, P>(a);
}

// ------------------------------------------------------------------------------------------
// exact lognormal law (montecarlo.jl:293-303, 384-390, 412-414, 454-459)
// ------------------------------------------------------------------------------------------

template <int P>
struct ExactState {
  DualT<P> x;
};

template <int P, bool REPLAY, bool ANTI, int PAIRS>
__global__ __launch_bounds__(kTile) void exact_gbm_kernel(const SimArgs<P> a) {
  const uint32_t chunk = blockIdx.x;
  const uint32_t tid = threadIdx.x;

  double acc[4 + P];
#pragma unroll
  for (int i = 0; i < 4 + P; ++i) acc[i] = 0.0;

  // pair j of this lane: trajectories chunk·512·PAIRS + j·512 + 2·tid, + 1 (a wave's samples are contiguous)
#pragma unroll 2
  for (int jp = 0; jp < PAIRS; ++jp) {
    const uint64_t path0 = ((uint64_t)chunk * PAIRS + jp) * (2 * kTile) + (uint64_t)tid * 2;
    double z[2];
    exact_pair_normals<REPLAY>(a, path0, z);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ExactState<P> st, sa;
      st.x.v = fma(a.law_sd.v, z[j], a.law_mu.v);
      if constexpr (ANTI) sa.x.v = 2 * a.law_mu.v - st.x.v;  // montecarlo.jl:387
      if constexpr (P > 0) {
#pragma unroll
        for (int k = 0; k < P; ++k) {
          st.x.d[k] = fma(a.law_sd.d[k], z[j], a.law_mu.d[k]);
          if constexpr (ANTI) sa.x.d[k] = 2 * a.law_mu.d[k] - st.x.d[k];
        }
      }
      finish_path<P, ANTI>(st, sa, a, path0 + j, acc);
    }
  }
  block_reduce_publish<4 + P, kTile / 64, 2>(acc, a.records + (size_t)chunk * kRecStride, a.accum != nullptr, a.map.n > 0);
  if (a.accum && reduces_records(chunk, a)) finish_records<kTile, P>(a);
}

// ------------------------------------------------------------------------------------------
// record reduction: one workgroup per accumulator slot, fixed summation order
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void reduce_records_kernel(const double* __restrict__ rec,
                                                              uint32_t n, double n_paths,
                                                              double* __restrict__ accum,
                                                              const PartialMap map,
                                                              const uint32_t* __restrict__ n_dev) {
  __shared__ double sm[257];
  const int slot = blockIdx.x;
  rec += (size_t)blockIdx.y * n * kRecStride;  // group = one payoff of a basket (the ALLOCATED count is the groups' stride)
  if (n_dev) n = min(n, *n_dev);  // (uniform) the records a Broadie–Kaya chain filled: bk_live_records()
  accum += (size_t)blockIdx.y * kRecStride;
  double out;
  const int k = slot - HH_ACC_DSUM;
  if (map.n > 0 && k >= 0 && k < HH_MAX_PARTIALS) {
    out = 0.0;
    if (k < map.n) {
      for (int j = 0; j < map.n_active; ++j)
        if (map.w[k][j] != 0.0) out = fma(map.w[k][j], sum_slot(rec, n, HH_ACC_DSUM + j, sm), out);
      if (map.xdT[k] != 0.0) out = fma(map.xdT[k], sum_slot(rec, n, kRecItmS, sm), out);
      if (map.dK[k] != 0.0) out -= map.dK[k] * sum_slot(rec, n, kRecItmS + 1, sm);
    }
  } else if (map.sim && (slot == kRecItmS || slot == kRecItmS + 1)) {
    out = 0.0;  // internal sums do not leave the device
  } else {
    out = sum_slot(rec, n, slot, sm);
  }
  if (threadIdx.x == 0) accum[slot] = (slot == HH_ACC_NPATHS) ? n_paths : out;
}

// ------------------------------------------------------------------------------------------
// REPLAY buffer helpers
// ------------------------------------------------------------------------------------------

constexpr int kFillSteps = 16;  // steps per workgroup of the fill kernel

template <int NC>
__global__ __launch_bounds__(kTile) void wiener_fill_kernel(double rho, double rho_c,
                                                            double sqrt_dt, uint32_t n_steps,
                                                            uint64_t n_paths,
                                                            const uint64_t* __restrict__ seeds,
                                                            double* __restrict__ dst) {
  const uint32_t tile = blockIdx.x;
  const uint32_t s0 = blockIdx.y * kFillSteps;
  const uint64_t path = (uint64_t)tile * kTile + threadIdx.x;
  const bool live = path < n_paths;
  const uint64_t key = live ? seeds[path] : 0ull;
  double* out = dst + (size_t)tile * n_steps * NC * kTile + threadIdx.x;
  for (uint32_t s = s0; s < s0 + kFillSteps && s < n_steps; ++s) {
    double z1, z2;
    if constexpr (NC == 2) {
      normal_pair(key, s, 0u, 0u, kDomEuler, z1, z2);
      const double d1 = sqrt_dt * z1;
      const double d2 = sqrt_dt * fma(rho, z1, rho_c * z2);
      out[((size_t)s * 2 + 0) * kTile] = live ? d1 : 0.0;
      out[((size_t)s * 2 + 1) * kTile] = live ? d2 : 0.0;
    } else {
      normal_pair(key, s >> 1, 0u, 0u, kDomEuler, z1, z2);
      const double d = sqrt_dt * ((s & 1u) ? z2 : z1);
      out[(size_t)s * kTile] = live ? d : 0.0;
    }
  }
}

#ifndef HH_PACK_STEPS
#define HH_PACK_STEPS 4  // steps per workgroup of the path-major -> tile-major repack (4.6 TB/s read+write; 8: 4.2, 16: 2.8)
#endif
constexpr int kPackSteps = HH_PACK_STEPS;

// src[path][step][comp] -> dst[tile][step][comp][256], transposed through LDS so that both the
// reads (along step·comp) and the writes (along path) are contiguous per wave.
template <int NC>
__global__ __launch_bounds__(256) void replay_pack_kernel(uint64_t n_paths, uint32_t n_steps,
                                                          const double* __restrict__ src,
                                                          double* __restrict__ dst) {
  constexpr int COLS = kPackSteps * NC;
  __shared__ double sm[kTile][COLS + 1];
  const uint32_t tile = blockIdx.x;
  const uint32_t s0 = blockIdx.y * kPackSteps;
  const int tid = threadIdx.x;
  const int col = tid % COLS, row0 = tid / COLS;
  constexpr int ROWS_PER_IT = 256 / COLS;
  for (int row = row0; row < kTile; row += ROWS_PER_IT) {
    const uint64_t path = (uint64_t)tile * kTile + row;
    const uint32_t s = s0 + col / NC;
    double v = 0.0;
    if (path < n_paths && s < n_steps) v = src[((size_t)path * n_steps + s0) * NC + col];
    sm[row][col] = v;
  }
  __syncthreads();
  double* out = dst + (size_t)tile * n_steps * NC * kTile + tid;
  for (int c = 0; c < COLS; ++c) {
    const uint32_t s = s0 + c / NC;
    if (s < n_steps) out[((size_t)s * NC + (c % NC)) * kTile] = sm[tid][c];
  }
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------

static inline double seed_of(const double* p, uint32_t k, uint32_t n) {
  return (p && k < n) ? p[k] : 0.0;
}

static const double* basis_seeds(const hh_model& m, int b) {
  return b == kBasisV0 ? m.dV0 : b == kBasisKappa ? m.dkappa : b == kBasisTheta ? m.dtheta : m.dsigma;
}

static PartialMap classify_partials(const hh_model& m, const hh_config& c) {
  PartialMap pm{};
  const uint32_t np = c.n_partials < HH_MAX_PARTIALS ? c.n_partials : HH_MAX_PARTIALS;
  pm.n = (int)np;
  const bool euler = c.strategy == HH_EULER_MARUYAMA;
  const bool heston = c.dynamics == HH_HESTON;
  const double dt = m.T / (double)(c.n_steps ? c.n_steps : 1);
  const double sqT = sqrt(m.T), tmul = c.compat_sqrt_alpha ? sqT : m.T;
  // parameters some requested direction differentiates along, among those the model reads
  for (int b = 0; b < kMaxBasis; ++b) {
    if (b != kBasisSigma && !(heston && euler)) continue;
    bool used = false;
    for (uint32_t k = 0; k < np; ++k) used = used || seed_of(basis_seeds(m, b), k, np) != 0.0;
    if (used) pm.basis[pm.n_active++] = b;
  }
  for (uint32_t k = 0; k < np; ++k) {
    for (int j = 0; j < pm.n_active; ++j) pm.w[k][j] = seed_of(basis_seeds(m, pm.basis[j]), k, np);
    pm.dK[k] = seed_of(m.dstrike, k, np);
    // ∂x_T from the spot and rate seeds: x0' = dS0/S0, then M drift updates fma(dt, dr, ·) (Euler)
    // or dr·tmul (exact law); the diffusion carries nothing along these
    const double dr = seed_of(m.dr_drift, k, np);
    double xd = seed_of(m.dS0, k, np) / m.S0;
    if (euler) {
      if (dr != 0.0)
        for (uint32_t s = 0; s < c.n_steps; ++s) xd = fma(dt, dr, xd);
    } else {
      xd = xd + dr * tmul;
    }
    pm.xdT[k] = xd;
  }
  return pm;
}

// Pairs of trajectories per lane of the exact-law kernels: 1 / 2 / 4 in a small ensemble (kExactPairsSmall),
// kExactPairs / kExactPairsHuge in a large / huge one.  A function of n_paths alone — every launcher and every
// record count in the library asks here.
int exact_pairs_per_lane(uint64_t n_paths) {
  static const int forced = [] {
    const char* e = getenv("HEDGEHOG_MC_EXACT_PAIRS");  // a measurement: 1, 2, 4, 8 or 64 for every size
    const int v = e ? atoi(e) : 0;
    return (v == 1 || v == 2 || v == 4 || v == kExactPairs || v == kExactPairsHuge) ? v : 0;
  }();
  if (forced) return forced;
  return n_paths >= (uint64_t)2048 * 512 * kExactPairsHuge ? kExactPairsHuge
         : n_paths >= (uint64_t)2048 * 512 * kExactPairs   ? kExactPairs
                                                           : kExactPairsSmall;
}

int count_active_partials(const hh_model& m, const hh_config& c) {
  return classify_partials(m, c).n_active;
}

template <int P>
static SimArgs<P> make_args(const hh_model& m, const hh_config& c, const DevicePtrs& p,
                            const PartialMap& pm) {
  SimArgs<P> a{};
  const double sig2h = 0.5 * m.sigma * m.sigma;
  a.x0.v = log(m.S0);
  a.v0.v = m.V0;
  a.kappa.v = m.kappa;
  a.theta.v = m.theta;
  a.sigma.v = m.sigma;
  a.r.v = m.r_drift;
  a.gdrift.v = m.r_drift - sig2h;
  a.strike.v = m.strike;
  // exact law (montecarlo.jl:302): Normal(log S0 + (r - σ²/2)·√α, σ·√α); compat flag keeps the √α
  const double sqT = sqrt(m.T);
  const double tmul = c.compat_sqrt_alpha ? sqT : m.T;
  a.law_mu.v = a.x0.v + a.gdrift.v * tmul;
  a.law_sd.v = m.sigma * sqT;
  for (int j = 0; j < P && j < pm.n_active; ++j) {  // carried slot j: unit seed on its parameter
    const int b = pm.basis[j];
    const double dsig = b == kBasisSigma ? 1.0 : 0.0;
    a.v0.d[j] = b == kBasisV0 ? 1.0 : 0.0;
    a.kappa.d[j] = b == kBasisKappa ? 1.0 : 0.0;
    a.theta.d[j] = b == kBasisTheta ? 1.0 : 0.0;
    a.sigma.d[j] = dsig;
    a.gdrift.d[j] = -m.sigma * dsig;
    a.law_mu.d[j] = a.gdrift.d[j] * tmul;
    a.law_sd.d[j] = dsig * sqT;
  }
  a.dt = m.T / (double)(c.n_steps ? c.n_steps : 1);  // montecarlo.jl:349
  a.sqrt_dt = sqrt(a.dt);
  a.rho = m.rho;
  a.rho_c = sqrt(1.0 - m.rho * m.rho);
  a.cp = m.cp;
  a.n_paths = c.n_paths;
  a.path_offset = c.path_offset;
  a.n_steps = c.n_steps;
  a.n_tiles = sim_records(c);
  a.tail_from = a.n_tiles > (uint32_t)HH_REPLAY_TAIL_TILES ? a.n_tiles - (uint32_t)HH_REPLAY_TAIL_TILES : 0u;
  a.seeds = p.seeds;
  a.replay = p.replay;
  a.terminal = p.terminal;
  a.terminal_d = p.terminal_d;
  a.records = p.records;
  a.accum = p.accum;
  a.acc_n_paths = (double)c.n_paths;
  a.finish_state = p.finish_state;
  a.finish_spin_ticks = p.finish_spin_ticks < 0 ? kFinishSpinTicksDefault : (unsigned long long)p.finish_spin_ticks;
  a.reducer_tile = p.finish_tile_first ? 0u : a.n_tiles - 1u;
  a.map = pm;
  return a;
}

SimArgs<0> make_args0(const hh_model& m, const hh_config& c, const DevicePtrs& p) {
  hh_config c0 = c;
  c0.n_partials = 0;
  return make_args<0>(m, c0, p, classify_partials(m, c0));
}

template <class M, int P, bool REPLAY, bool ANTI>
static int launch_euler_t(const SimArgs<P>& a, hipStream_t s) {
#ifdef HH_REPLAY_VARIANTS
  constexpr int PPT = !REPLAY ? 1 : (P == 0 && !ANTI) ? HH_REPLAY_PPT : (P == 0 ? HH_REPLAY_PPT_ANTI : HH_REPLAY_PPT_DUAL);
  constexpr int RING = !REPLAY ? 0 : ANTI ? HH_REPLAY_LDS_ANTI : P > 0 ? HH_REPLAY_LDS_DUAL : HH_REPLAY_LDS;
  constexpr bool PIPE = HH_REPLAY_PIPE != 0;
  // With the standard ring (32 KiB of LDS per Heston workgroup) a CU holds 4 workgroups, the chip 1024; a
  // grid of at most half that cannot fill it, so each wave gets a deeper, pipelined ring instead
  if constexpr (RING > 0 && HH_REPLAY_LDS_DEEP > 0) {
    if (a.n_tiles <= 512u) {
      hipLaunchKernelGGL((euler_kernel<M, P, REPLAY, ANTI, PPT, HH_REPLAY_LDS_DEEP, true>), dim3(a.n_tiles),
                         dim3(kTile / PPT), 0, s, a);
      return (int)hipGetLastError();
    }
  }
#else
  constexpr int PPT = 1, RING = 0;  // one trajectory per lane, the register pipeline
  constexpr bool PIPE = false;
#endif
  hipLaunchKernelGGL((euler_kernel<M, P, REPLAY, ANTI, PPT, RING, PIPE>), dim3(a.n_tiles), dim3(kTile / PPT), 0, s, a);
  return (int)hipGetLastError();
}

template <class M, int P>
static int launch_euler_pm(const SimArgs<P>& a, bool anti, hipStream_t s) {
  auto kernel = anti ? euler_pm_kernel<M, P, true> : euler_pm_kernel<M, P, false>;
  hipLaunchKernelGGL(kernel, dim3(a.n_tiles), dim3(kTile), 0, s, a);
  return (int)hipGetLastError();
}

template <class M, int P>
static int launch_euler_m(const SimArgs<P>& a, bool replay, bool anti, hipStream_t s, bool path_major = false) {
  if (replay && path_major) return launch_euler_pm<M, P>(a, anti, s);
#if defined(HH_REPLAY_VARIANTS) && HH_ANTI_SPLIT
  if constexpr (P == 0) {
    if (replay && anti) {
      hipLaunchKernelGGL((euler_pair_split_kernel<M>), dim3(a.n_tiles), dim3(2 * kTile), 0, s, a);
      return (int)hipGetLastError();
    }
  }
#endif
  if (replay) return anti ? launch_euler_t<M, P, true, true>(a, s) : launch_euler_t<M, P, true, false>(a, s);
  return anti ? launch_euler_t<M, P, false, true>(a, s) : launch_euler_t<M, P, false, false>(a, s);
}

template <int P>
static int launch_sim_p(const hh_model& m, const hh_config& c, const DevicePtrs& p,
                        const PartialMap& pm, hipStream_t s) {
  const SimArgs<P> a = make_args<P>(m, c, p, pm);
  const bool anti = c.antithetic != 0;
  const bool replay = c.noise_mode == HH_NOISE_REPLAY;
  if (c.strategy == HH_EXACT_LAW) {
    auto pick = [&](auto replay_c, auto anti_c) {
      constexpr bool R = decltype(replay_c)::value, A = decltype(anti_c)::value;
      const int pairs = exact_pairs_per_lane(c.n_paths);
      return pairs == kExactPairsHuge ? exact_gbm_kernel<P, R, A, kExactPairsHuge>
             : pairs == kExactPairs   ? exact_gbm_kernel<P, R, A, kExactPairs>
             : pairs == 4             ? exact_gbm_kernel<P, R, A, 4>
             : pairs == 2             ? exact_gbm_kernel<P, R, A, 2>
                                      : exact_gbm_kernel<P, R, A, 1>;
    };
    using T = std::true_type;
    using F = std::false_type;
    auto kernel = replay ? (anti ? pick(T{}, T{}) : pick(T{}, F{})) : (anti ? pick(F{}, T{}) : pick(F{}, F{}));
    hipLaunchKernelGGL(kernel, dim3(a.n_tiles), dim3(kTile), 0, s, a);
    return (int)hipGetLastError();
  }
  const bool direct = replay && p.replay_path_major;
  if (c.dynamics == HH_LOGNORMAL) return launch_euler_m<GbmModel<P>, P>(a, replay, anti, s, direct);
  if (c.em_split) return launch_euler_m<HestonModel<P, true>, P>(a, replay, anti, s, direct);
  return launch_euler_m<HestonModel<P, false>, P>(a, replay, anti, s, direct);
}

int launch_simulation(const hh_model& m, const hh_config& c, const DevicePtrs& p, hipStream_t s) {
  const PartialMap pm = classify_partials(m, c);
  switch (pad_partials((uint32_t)pm.n_active)) {
    case 0: return launch_sim_p<0>(m, c, p, pm, s);
    case 1: return launch_sim_p<1>(m, c, p, pm, s);
    case 2: return launch_sim_p<2>(m, c, p, pm, s);
    case 3: return launch_sim_p<3>(m, c, p, pm, s);
    default: return launch_sim_p<4>(m, c, p, pm, s);
  }
}

int launch_reduce_records(const double* records, uint32_t n_records, double n_paths, double* accum,
                          hipStream_t s, uint32_t n_groups, const hh_model* m, const hh_config* c,
                          bool basket, const uint32_t* n_records_dev) {
  PartialMap pm{};
  if (m && c && (basket || c->strategy != HH_BROADIE_KAYA)) {
    if (c->n_partials) pm = classify_partials(*m, *c);
    pm.sim = 1;
    if (basket)  // strike partials are not carried through a basket
      for (double& d : pm.dK) d = 0.0;
  }
  hipLaunchKernelGGL(reduce_records_kernel, dim3(kRecStride, n_groups), dim3(256), 0, s, records,
                     n_records, n_paths, accum, pm, n_records_dev);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// basket: K payoffs reduced over one set of terminal samples.  The reference prices a basket as K
// independent solves (basket.jl:35-38); with fixed seeds, payoffs sharing an expiry see the SAME
// trajectories, so one simulation + this kernel is result-equivalent.
// ------------------------------------------------------------------------------------------

// Block -> (chunk, payoff).  Every payoff re-reads the same terminal samples; workgroups are dealt
// round-robin over the 8 XCDs by their linear index (MI355X_MICROARCH.md, dispatch), so the chunks
// are spread such that chunk c is ALWAYS handled on XCD c mod 8, whatever the payoff: each XCD's
// private L2 then holds one eighth of the samples (1 MB of 8 at 10^6 trajectories) for all payoffs,
// instead of every L2 streaming all of them.  Placement is for speed only: results do not depend on it.
#ifndef HH_BASKET_XCD
#define HH_BASKET_XCD 1
#endif
#ifndef HH_BASKET_KB
#define HH_BASKET_KB 4
#endif
constexpr int kBasketKB = HH_BASKET_KB;  // payoffs a workgroup evaluates on each sample it loads

// One workgroup = one chunk of samples x kBasketKB payoffs: a sample (and its tangents) is loaded once
// and every payoff of the group is evaluated on it from registers, so the re-read of the samples
// shrinks by kBasketKB and the kernel is bound by its fp64 selects and adds.  Each payoff keeps its own
// accumulators, lane order and record, so its sums are those of a one-payoff launch bit for bit.
template <int P>
__global__ __launch_bounds__(256) void basket_payoff_kernel(const BasketArgs b, const uint32_t n_payoffs) {
#if HH_BASKET_XCD
  const uint32_t per_xcd = (b.n_chunks + 7u) / 8u;           // chunks an XCD owns
  const uint32_t xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const uint32_t grp = idx / per_xcd, chunk = (idx % per_xcd) * 8u + xcd;
  if (chunk >= b.n_chunks) return;                           // padding of the last group of eight
#else
  const uint32_t chunk = blockIdx.x, grp = blockIdx.y;
#endif
  const uint32_t k0 = grp * kBasketKB;
  double strike[kBasketKB], cp[kBasketKB];
#pragma unroll
  for (int g = 0; g < kBasketKB; ++g) {                      // a short last group repeats its last payoff
    const uint32_t k = min(k0 + g, n_payoffs - 1);
    strike[g] = b.strikes[k];
    cp[g] = b.cps[k];
  }
  const uint64_t n_total = b.antithetic ? 2 * b.n_paths : b.n_paths;
  double acc[kBasketKB][4 + P];
#pragma unroll
  for (int g = 0; g < kBasketKB; ++g)
#pragma unroll
    for (int i = 0; i < 4 + P; ++i) acc[g][i] = 0.0;
  const uint64_t i0 = (uint64_t)chunk * kBasketChunk;
  for (uint32_t j = threadIdx.x; j < (uint32_t)kBasketChunk; j += 256) {
    const uint64_t i = i0 + j;
    if (i >= b.n_paths) break;
    const double S = b.terminal[i];
    double Sa = 0.0, d[P > 0 ? P : 1], da[P > 0 ? P : 1];
    if constexpr (P > 0) {  // terminal_d holds dS_T = S·∂x of the ACTIVE directions
#pragma unroll
      for (int q = 0; q < P; ++q) d[q] = b.terminal_d[(uint64_t)q * n_total + i];
    }
    if (b.antithetic) {
      Sa = b.terminal[b.n_paths + i];
      if constexpr (P > 0) {
#pragma unroll
        for (int q = 0; q < P; ++q) da[q] = b.terminal_d[(uint64_t)q * n_total + b.n_paths + i];
      }
    }
#pragma unroll
    for (int g = 0; g < kBasketKB; ++g) {
      const double m = cp[g] * (S - strike[g]);
      const bool itm = m > 0.0;
      double p = itm ? m : 0.0;
      double wS = itm ? cp[g] * S : 0.0, wN = itm ? cp[g] : 0.0;
      double pd[P > 0 ? P : 1];
      if constexpr (P > 0) {
#pragma unroll
        for (int q = 0; q < P; ++q) pd[q] = itm ? cp[g] * d[q] : 0.0;
      }
      if (b.antithetic) {
        const double ma = cp[g] * (Sa - strike[g]);
        const bool itma = ma > 0.0;
        p = (p + (itma ? ma : 0.0)) / 2;
        wS = (wS + (itma ? cp[g] * Sa : 0.0)) / 2;
        wN = (wN + (itma ? cp[g] : 0.0)) / 2;
        if constexpr (P > 0) {
#pragma unroll
          for (int q = 0; q < P; ++q) pd[q] = (pd[q] + (itma ? cp[g] * da[q] : 0.0)) / 2;
        }
      }
      acc[g][0] += p;
      acc[g][1] = fma(p, p, acc[g][1]);
      if constexpr (P > 0) {
#pragma unroll
        for (int q = 0; q < P; ++q) acc[g][2 + q] += pd[q];
      }
      acc[g][2 + P] += wS;
      acc[g][3 + P] += wN;
    }
  }
#pragma unroll
  for (int g = 0; g < kBasketKB; ++g) {
    if (k0 + g >= n_payoffs) break;                          // uniform over the workgroup
    if (g) __syncthreads();                                  // the reduction's LDS staging is reused
    block_reduce_store<4 + P, 4, 2>(acc[g], b.records + ((size_t)(k0 + g) * b.n_chunks + chunk) * kRecStride);
  }
}

// a basket on Broadie–Kaya samples: the simulation's own counters (fall-backs, series terms) belong
// to every payoff's result
__global__ void copy_bk_counters_kernel(const double* __restrict__ src, double* __restrict__ accum,
                                        uint32_t n_groups) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_groups) return;
  for (int i = HH_ACC_BK_NEWTON_FAIL; i <= HH_ACC_BK_CF_TERMS; ++i)
    accum[(size_t)k * kRecStride + i] = src[i];
}

int launch_copy_bk_counters(const double* src, double* accum, uint32_t n_groups, hipStream_t s) {
  hipLaunchKernelGGL(copy_bk_counters_kernel, dim3((n_groups + 255) / 256), dim3(256), 0, s, src, accum,
                     n_groups);
  return (int)hipGetLastError();
}

int launch_basket_payoffs(const BasketArgs& b, uint32_t n_payoffs, uint32_t n_active_partials,
                          hipStream_t s) {
  const uint32_t n_groups = (n_payoffs + kBasketKB - 1) / kBasketKB;
  // a 1-D grid: 2^31 workgroups is 3.5·10^13 payoff evaluations; beyond that the caller splits the basket
  if ((uint64_t)((b.n_chunks + 7u) / 8u) * 8u * n_groups > 0x7fffffffull) return (int)hipErrorInvalidConfiguration;
#if HH_BASKET_XCD
  const dim3 grid(((b.n_chunks + 7u) / 8u) * 8u * n_groups), block(256);
#else
  const dim3 grid(b.n_chunks, n_groups), block(256);
#endif
  switch (pad_partials(n_active_partials)) {
    case 0: hipLaunchKernelGGL(basket_payoff_kernel<0>, grid, block, 0, s, b, n_payoffs); break;
    case 1: hipLaunchKernelGGL(basket_payoff_kernel<1>, grid, block, 0, s, b, n_payoffs); break;
    case 2: hipLaunchKernelGGL(basket_payoff_kernel<2>, grid, block, 0, s, b, n_payoffs); break;
    case 3: hipLaunchKernelGGL(basket_payoff_kernel<3>, grid, block, 0, s, b, n_payoffs); break;
    default: hipLaunchKernelGGL(basket_payoff_kernel<4>, grid, block, 0, s, b, n_payoffs); break;
  }
  return (int)hipGetLastError();
}

int launch_wiener_fill(int dynamics, double rho, double sqrt_dt, uint32_t n_steps, uint64_t n_paths,
                       const uint64_t* seeds_dev, double* dst, hipStream_t s) {
  const dim3 grid(tiles_for(n_paths), (n_steps + kFillSteps - 1) / kFillSteps);
  const double rho_c = sqrt(1.0 - rho * rho);
  if (dynamics == HH_HESTON)
    hipLaunchKernelGGL(wiener_fill_kernel<2>, grid, dim3(kTile), 0, s, rho, rho_c, sqrt_dt, n_steps,
                       n_paths, seeds_dev, dst);
  else
    hipLaunchKernelGGL(wiener_fill_kernel<1>, grid, dim3(kTile), 0, s, rho, rho_c, sqrt_dt, n_steps,
                       n_paths, seeds_dev, dst);
  return (int)hipGetLastError();
}

int launch_replay_pack(int ncomp, uint64_t n_paths, uint32_t n_steps, const double* src_dev,
                       double* dst_dev, hipStream_t s) {
  const dim3 grid(tiles_for(n_paths), (n_steps + kPackSteps - 1) / kPackSteps);
  if (ncomp == 2)
    hipLaunchKernelGGL(replay_pack_kernel<2>, grid, dim3(256), 0, s, n_paths, n_steps, src_dev,
                       dst_dev);
  else
    hipLaunchKernelGGL(replay_pack_kernel<1>, grid, dim3(256), 0, s, n_paths, n_steps, src_dev,
                       dst_dev);
  return (int)hipGetLastError();
}

}  // namespace hh
