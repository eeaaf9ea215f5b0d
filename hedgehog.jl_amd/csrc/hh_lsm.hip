// Longstaff–Schwartz American pricing on the full path grid — the consumer of simulate_paths that
// needs every time step.  Reference: solve(::PricingProblem{VanillaOption{…,
// American,…}}, ::LSM), src/pricing_methods/least_squares_montecarlo.jl:99-165, on the paths of
// sde_problem(::LognormalDynamics, ::BlackScholesExact) (montecarlo.jl:140-159, antithetic :270-284).
//
// Layout: spot grid S[step][path] (step-major: every backward step streams two contiguous rows),
// per-path stopping state tau[path] (int32), val[path] (fp64).
//
// The backward induction is serial in time but data-parallel over paths; each step is ONE launch:
//   reduce last launch's partial moment sums Σ z^k y  ->  solve the (d+1)x(d+1) normal equations
//   (every workgroup redundantly, same order => same coefficients)  ->  exercise decision for its
//   own paths  ->  its partial moment sums for the NEXT (earlier) row.
// Everything that depends on the spots only (in-the-money count, mean, std, power sums Σ z^k of every
// row) is precomputed for all rows in three launches.  The regression is done in the standardised
// variable z = (x - mean)/std: same polynomial space as Polynomials.fit(x, y, degree) (:126), so the
// fitted function is the same; its Gram matrix is well conditioned in fp64.
#include <cmath>

#include "hh_kernels.h"
#include "hh_rng.h"

namespace hh {

namespace {

constexpr int kLsmChunk = 1024;  // paths per workgroup (256 threads x 4) of the one-off kernels
// The backward-step kernels take Q paths per thread: 4, or 16 for large ensembles — every workgroup
// re-reduces all workgroups' partial moment sums at the start of a step, n_chunks²·(D+1) reads per
// exercise date, which at 2·10⁶ paths and 1024-path chunks is more traffic than the paths themselves
#ifndef HH_LSM_WIDE_Q
#define HH_LSM_WIDE_Q 8
#endif
constexpr uint64_t kLsmWideFrom = 1ull << 19;  // ensembles from this size on use Q = HH_LSM_WIDE_Q
// paths per workgroup of the one-off row-statistics kernels (grid = chunks x rows): with 1024 the
// workgroup reductions of the 3 / 2D+1 sums cost more than reading the paths
inline uint32_t lsm_one_off_paths(uint64_t ntot) { return ntot >= kLsmWideFrom ? 8192u : 1024u; }
constexpr int kLsmMaxDeg = 8;

// ---- full path grid -----------------------------------------------------------------------

template <bool ANTI>
__global__ __launch_bounds__(256) void gbm_grid_kernel(const uint64_t* __restrict__ seeds,
                                                       uint64_t n_paths, uint32_t n_steps,
                                                       double S0, double a, double b,
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
        S = S + S * (exp(fma(b, z[h], a)) - 1.0);
        grid[(size_t)(s + h + 1) * ntot + i] = S;
        if (ANTI) {  // flipped σ, same draws (montecarlo.jl:276)
          Sa = Sa + Sa * (exp(fma(-b, z[h], a)) - 1.0);
          grid[(size_t)(s + h + 1) * ntot + n_paths + i] = Sa;
        }
      }
    }
  }
}

// ---- helpers --------------------------------------------------------------------------------

template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double (&out)[N]) {
  // all threads receive the workgroup sums (fixed order: wave tree, then waves 0..3)
  __shared__ double sm[4][N];
  __syncthreads();  // protect sm from a previous use
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) sm[wave][i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = ((sm[0][i] + sm[1][i]) + sm[2][i]) + sm[3][i];
}

// sum NV-vectors written by `n_rec` workgroups (rec[r*NV + i]) — every workgroup does this in the
// same order, so all of them obtain bit-identical totals
template <int NV>
__device__ __forceinline__ void reduce_records(const double* __restrict__ rec, uint32_t n_rec,
                                               double (&tot)[NV]) {
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.0;
  for (uint32_t r = threadIdx.x; r < n_rec; r += 256) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] += rec[(size_t)r * NV + i];
  }
  block_sum<NV>(v, tot);
}

struct RowStat {
  double n, mu, sd;
};

// ---- precompute: per-row in-the-money statistics and power sums --------------------------------

__global__ __launch_bounds__(256) void lsm_stats_kernel(const double* __restrict__ grid,
                                                        uint64_t ntot, double strike, double cp,
                                                        uint32_t n_chunks, uint32_t per_wg,
                                                        double* __restrict__ rec /*[row][chunk][3]*/) {
  const uint32_t chunk = blockIdx.x, row = blockIdx.y;
  const double* S = grid + (size_t)row * ntot;
  double v[3] = {0, 0, 0};
  for (uint32_t j = 0; j < per_wg / 256; ++j) {
    const uint64_t p = (uint64_t)chunk * per_wg + j * 256 + threadIdx.x;
    if (p < ntot) {
      const double x = S[p];
      if (cp * (x - strike) > 0.0) {
        v[0] += 1.0;
        v[1] += x;
        v[2] = fma(x, x, v[2]);
      }
    }
  }
  double t[3];
  block_sum<3>(v, t);
  if (threadIdx.x == 0) {
    double* o = rec + ((size_t)row * n_chunks + chunk) * 3;
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
  }
}

// count, mean, std of the in-the-money spots of a row from (n, Σx, Σx²)
__device__ __forceinline__ RowStat rowstat_of(const double (&t)[3]) {
  RowStat r{t[0], 0.0, 1.0};
  if (t[0] > 0.0) {
    r.mu = t[1] / t[0];
    const double var = t[2] / t[0] - r.mu * r.mu;
    r.sd = var > 0.0 ? sqrt(var) : 1.0;
  }
  return r;
}

__global__ __launch_bounds__(256) void lsm_rowstat_kernel(const double* __restrict__ rec,
                                                          uint32_t n_chunks,
                                                          RowStat* __restrict__ rs) {
  const uint32_t row = blockIdx.x;
  double t[3];
  reduce_records<3>(rec + (size_t)row * n_chunks * 3, n_chunks, t);
  if (threadIdx.x == 0) rs[row] = rowstat_of(t);
}

// sharded form: the sums leave the device (all-reduce over the ranks) and come back
__global__ __launch_bounds__(256) void lsm_rowstat_from_sums_kernel(const double* __restrict__ sums,
                                                                    uint32_t rows,
                                                                    RowStat* __restrict__ rs) {
  const uint32_t row = blockIdx.x * 256 + threadIdx.x;
  if (row < rows) {
    const double t[3] = {sums[row * 3], sums[row * 3 + 1], sums[row * 3 + 2]};
    rs[row] = rowstat_of(t);
  }
}

// out[row][NV] = Σ_chunk rec[row][chunk][NV], same order as the in-kernel reductions (grid = rows)
template <int NV>
__global__ __launch_bounds__(256) void lsm_sum_records_kernel(const double* __restrict__ rec,
                                                              uint32_t n_chunks,
                                                              double* __restrict__ out) {
  const uint32_t row = blockIdx.x;
  double t[NV];
  reduce_records<NV>(rec + (size_t)row * n_chunks * NV, n_chunks, t);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) out[(size_t)row * NV + i] = t[i];
  }
}

template <int D>
__global__ __launch_bounds__(256) void lsm_pow_kernel(const double* __restrict__ grid, uint64_t ntot,
                                                      double strike, double cp, uint32_t n_chunks,
                                                      uint32_t per_wg,
                                                      const RowStat* __restrict__ rs,
                                                      double* __restrict__ rec /*[row][chunk][2D+1]*/) {
  constexpr int NV = 2 * D + 1;
  const uint32_t chunk = blockIdx.x, row = blockIdx.y;
  const double* S = grid + (size_t)row * ntot;
  const RowStat r = rs[row];
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.0;
  for (uint32_t j = 0; j < per_wg / 256; ++j) {
    const uint64_t p = (uint64_t)chunk * per_wg + j * 256 + threadIdx.x;
    if (p < ntot) {
      const double x = S[p];
      if (cp * (x - strike) > 0.0) {
        const double z = (x - r.mu) / r.sd;
        double pw = 1.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          v[i] += pw;
          pw *= z;
        }
      }
    }
  }
  double t[NV];
  block_sum<NV>(v, t);
  if (threadIdx.x == 0) {
    double* o = rec + ((size_t)row * n_chunks + chunk) * NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = t[i];
  }
}

template <int D>
__global__ __launch_bounds__(256) void lsm_powsum_kernel(const double* __restrict__ rec,
                                                         uint32_t n_chunks,
                                                         double* __restrict__ P /*[row][2D+1]*/) {
  constexpr int NV = 2 * D + 1;
  const uint32_t row = blockIdx.x;
  double t[NV];
  reduce_records<NV>(rec + (size_t)row * n_chunks * NV, n_chunks, t);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) P[(size_t)row * NV + i] = t[i];
  }
}

// ---- backward induction -----------------------------------------------------------------------

struct LsmStepArgs {
  const double* grid;
  uint64_t ntot;
  double strike, cp, ln_disc;  // ln of the per-step discount factor
  uint32_t n_steps, n_chunks;  // n_chunks: workgroups of the step kernels (256·Q paths each)
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

// contribution of this workgroup's paths to Σ z^k y of `row`, y = D^(tau - row) val
// (least_squares_montecarlo.jl:115-116), written to recB[row][chunk]
template <int D, int Q>
__device__ __forceinline__ void emit_moments(const LsmStepArgs& a, uint32_t row, const int (&tau)[Q],
                                             const double (&val)[Q], const double (&xrow)[Q]) {
  const RowStat r = a.rs[row];
  double v[D + 1];
#pragma unroll
  for (int i = 0; i <= D; ++i) v[i] = 0.0;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = (uint64_t)blockIdx.x * (256 * Q) + j * 256 + threadIdx.x;
    if (p < a.ntot) {
      const double x = xrow[j];  // S[row][p], loaded by the caller
      if (a.cp * (x - a.strike) > 0.0) {
        const double z = (x - r.mu) / r.sd;
        const double y = a.disc_pow[tau[j] - (int)row] * val[j];  // tau >= row + 1
        double pw = y;
#pragma unroll
        for (int i = 0; i <= D; ++i) {
          v[i] += pw;
          pw *= z;
        }
      }
    }
  }
  double t[D + 1];
  block_sum<D + 1>(v, t);
  if (threadIdx.x == 0) {
    double* o = a.recB + ((size_t)row * a.n_chunks + blockIdx.x) * (D + 1);
#pragma unroll
    for (int i = 0; i <= D; ++i) o[i] = t[i];
  }
}

// stopping_info = [(nsteps, payoff(S_T))] (:109), and the moment sums of row nsteps-1
template <int D, int Q>
__global__ __launch_bounds__(256) void lsm_init_kernel(const LsmStepArgs a) {
  const double* S = a.grid + (size_t)a.n_steps * a.ntot;
  const double* Sn = a.grid + (size_t)(a.n_steps >= 2 ? a.n_steps - 1 : a.n_steps) * a.ntot;
  int tau[Q];
  double val[Q], xn[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = (uint64_t)blockIdx.x * (256 * Q) + j * 256 + threadIdx.x;
    tau[j] = (int)a.n_steps;
    val[j] = xn[j] = 0.0;
    if (p < a.ntot) {
      xn[j] = Sn[p];
      const double m = a.cp * (S[p] - a.strike);
      val[j] = m > 0.0 ? m : 0.0;
      a.tau[p] = tau[j];
      a.val[p] = val[j];
    }
  }
  if (a.n_steps >= 2) emit_moments<D, Q>(a, a.n_steps - 1, tau, val, xn);
}

// one backward step at time index t (the reference's loop body for i = t+1, :112-131)
template <int D, int Q>
__global__ __launch_bounds__(256) void lsm_step_kernel(const LsmStepArgs a, uint32_t t) {
  constexpr int N = D + 1;
  const RowStat r = a.rs[t];
  __shared__ double coef[N];
  __shared__ int have_fit;
  // this workgroup's paths first: the loads are in flight while the moment sums are reduced and the
  // normal equations solved (a serial prologue of several microseconds in every workgroup)
  const double* S = a.grid + (size_t)t * a.ntot;
  const double* Sn = a.grid + (size_t)(t >= 2 ? t - 1 : t) * a.ntot;  // row of the next step
  int tau[Q];
  double val[Q], xs[Q], xn[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = (uint64_t)blockIdx.x * (256 * Q) + j * 256 + threadIdx.x;
    tau[j] = 0;
    val[j] = xs[j] = xn[j] = 0.0;
    if (p < a.ntot) {
      tau[j] = a.tau[p];
      val[j] = a.val[p];
      xs[j] = S[p];
      xn[j] = Sn[p];
    }
  }
  double B[N];
  if (a.B_given) {  // summed over the ranks by the host between two launches
#pragma unroll
    for (int i = 0; i < N; ++i) B[i] = a.B_given[i];
  } else {
    reduce_records<N>(a.recB + (size_t)t * a.n_chunks * N, a.n_chunks, B);
  }
  if (threadIdx.x == 0) {
    have_fit = 0;
    if (r.n > 0.0) {  // isempty(in_the_money) && continue (:120)
      // normal equations G c = B, G_jk = Σ z^(j+k); Gaussian elimination with partial pivoting
      const double* P = a.P + (size_t)t * (2 * D + 1);
      double M[N][N + 1];
#pragma unroll
      for (int j = 0; j < N; ++j) {
#pragma unroll
        for (int k = 0; k < N; ++k) M[j][k] = P[j + k];
        M[j][N] = B[j];
      }
      double scale = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) scale = fmax(scale, fabs(M[j][j]));
      bool dead[N];
#pragma unroll
      for (int c = 0; c < N; ++c) {
        int piv = c;
        double best = fabs(M[c][c]);
#pragma unroll
        for (int j = 0; j < N; ++j)
          if (j > c && fabs(M[j][c]) > best) {
            best = fabs(M[j][c]);
            piv = j;
          }
#pragma unroll
        for (int j = 0; j < N; ++j)
          if (j == piv && piv != c) {
#pragma unroll
            for (int k = 0; k <= N; ++k) {
              const double tmp = M[c][k];
              M[c][k] = M[j][k];
              M[j][k] = tmp;
            }
          }
        // rank deficiency (fewer distinct in-the-money spots than coefficients): drop the column
        dead[c] = !(best > 1e-13 * scale);
        if (!dead[c]) {
          const double inv = 1.0 / M[c][c];
#pragma unroll
          for (int j = 0; j < N; ++j)
            if (j > c) {
              const double f = M[j][c] * inv;
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
        cf[c] = dead[c] ? 0.0 : s / M[c][c];
      }
#pragma unroll
      for (int c = 0; c < N; ++c) coef[c] = cf[c];
      have_fit = 1;
    }
    if (blockIdx.x == 0) a.counters[r.n > 0.0 ? 0 : 1] += 1.0;
  }
  __syncthreads();

#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const uint64_t p = (uint64_t)blockIdx.x * (256 * Q) + j * 256 + threadIdx.x;
    if (p < a.ntot) {
      const double x = xs[j];
      const double pay = a.cp * (x - a.strike);
      if (have_fit && pay > 0.0) {
        const double z = (x - r.mu) / r.sd;
        double cont = coef[D];  // cont_value = poly(x) (:127), Horner in z
#pragma unroll
        for (int c = D - 1; c >= 0; --c) cont = fma(cont, z, coef[c]);
        if (pay > cont) {  // update_stopping_info! (:163-164)
          tau[j] = (int)t;
          val[j] = pay;
          a.tau[p] = tau[j];
          a.val[p] = pay;
        }
      }
    }
  }
  if (t >= 2) emit_moments<D, Q>(a, t - 1, tau, val, xn);
}

// discounted_values = discount^t * val (:133): per-workgroup Σ and Σ² into 16-double records
__global__ __launch_bounds__(256) void lsm_final_kernel(const int32_t* __restrict__ tau,
                                                        const double* __restrict__ val,
                                                        uint64_t ntot, double ln_disc,
                                                        double* __restrict__ records) {
  double v[2] = {0, 0};
  for (int j = 0; j < 4; ++j) {
    const uint64_t p = (uint64_t)blockIdx.x * kLsmChunk + j * 256 + threadIdx.x;
    if (p < ntot) {
      const double d = exp(ln_disc * (double)tau[p]) * val[p];
      v[0] += d;
      v[1] = fma(d, d, v[1]);
    }
  }
  double t[2];
  block_sum<2>(v, t);
  if (threadIdx.x == 0) {
    double* rec = records + (size_t)blockIdx.x * kRecStride;
    for (int i = 0; i < kRecStride; ++i) rec[i] = 0.0;
    rec[HH_ACC_SUM] = t[0];
    rec[HH_ACC_SUMSQ] = t[1];
  }
}

// ---- launch sequences ---------------------------------------------------------------------------

// where the pieces of the caller's scratch buffer are (lsm_scratch_doubles() doubles)
struct LsmLayout {
  double* rec_stats;  // [rows][ch1][3]
  RowStat* rs;        // [rows]
  double* rec_pow;    // [rows][ch1][2D+1]
  double* P;          // [rows][2D+1]
  double* recB;       // [rows][ch][D+1]
  double* disc_pow;   // [rows]
  double* counters;   // [2]
  uint32_t rows, ch1, per_wg;
};

LsmLayout lsm_layout(double* scratch, uint64_t ntot, uint32_t n_steps, int degree) {
  const size_t rows = (size_t)n_steps + 1, ch = lsm_chunks(ntot), nv = 2 * (size_t)degree + 1;
  LsmLayout L{};
  L.rec_stats = scratch;
  L.rs = reinterpret_cast<RowStat*>(L.rec_stats + rows * ch * 3);
  L.rec_pow = reinterpret_cast<double*>(L.rs) + rows * 3;
  L.P = L.rec_pow + rows * ch * nv;
  L.recB = L.P + rows * nv;
  L.disc_pow = L.recB + rows * ch * (degree + 1);
  L.counters = L.disc_pow + rows;
  L.rows = (uint32_t)rows;
  L.per_wg = lsm_one_off_paths(ntot);
  L.ch1 = (uint32_t)((ntot + L.per_wg - 1) / L.per_wg);
  return L;
}

LsmStepArgs lsm_step_args(const LsmLayout& L, const double* grid, uint64_t ntot, uint32_t n_steps,
                          double strike, double cp, double step_discount, int32_t* tau, double* val) {
  LsmStepArgs a{};
  a.grid = grid; a.ntot = ntot; a.strike = strike; a.cp = cp; a.ln_disc = log(step_discount);
  a.n_steps = n_steps; a.tau = tau; a.val = val; a.rs = L.rs; a.P = L.P; a.recB = L.recB;
  a.disc_pow = L.disc_pow; a.counters = L.counters;
  const bool wide = ntot >= kLsmWideFrom;
  a.n_chunks = wide ? (uint32_t)((ntot + 256 * HH_LSM_WIDE_Q - 1) / (256 * HH_LSM_WIDE_Q))
                    : lsm_chunks(ntot);
  return a;
}

template <int D>
void launch_pow(const LsmLayout& L, const LsmStepArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(lsm_pow_kernel<D>, dim3(L.ch1, L.rows), dim3(256), 0, s, a.grid, a.ntot, a.strike,
                     a.cp, L.ch1, L.per_wg, a.rs, L.rec_pow);
}

template <int D>
void launch_init(const LsmStepArgs& a, hipStream_t s) {
  if (a.ntot >= kLsmWideFrom)
    hipLaunchKernelGGL((lsm_init_kernel<D, HH_LSM_WIDE_Q>), dim3(a.n_chunks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((lsm_init_kernel<D, 4>), dim3(a.n_chunks), dim3(256), 0, s, a);
}

template <int D>
void launch_step(const LsmStepArgs& a, uint32_t t, hipStream_t s) {
  if (a.ntot >= kLsmWideFrom)
    hipLaunchKernelGGL((lsm_step_kernel<D, HH_LSM_WIDE_Q>), dim3(a.n_chunks), dim3(256), 0, s, a, t);
  else
    hipLaunchKernelGGL((lsm_step_kernel<D, 4>), dim3(a.n_chunks), dim3(256), 0, s, a, t);
}

// the whole induction on one device
template <int D>
int run_lsm(const LsmLayout& L, const LsmStepArgs& a, hipStream_t s) {
  launch_pow<D>(L, a, s);
  hipLaunchKernelGGL(lsm_powsum_kernel<D>, dim3(L.rows), dim3(256), 0, s, L.rec_pow, L.ch1, L.P);
  launch_init<D>(a, s);
  for (uint32_t t = a.n_steps - 1; t >= 1; --t)  // for i = nsteps:-1:2, t = i-1 (:112-113)
    launch_step<D>(a, t, s);
  return (int)hipGetLastError();
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
      hipLaunchKernelGGL(lsm_powsum_kernel<D>, dim3(L.rows), dim3(256), 0, s, L.rec_pow, L.ch1, vec_out);
      break;
    case kLsmPhaseInit: {  // global power sums in; stopping at expiry; local moment sums of row n-1 out
      hipError_t e = hipMemcpyAsync(L.P, vec_in, (size_t)L.rows * NV * sizeof(double),
                                    hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return (int)e;
      launch_init<D>(a, s);
      if (a.n_steps >= 2)
        hipLaunchKernelGGL(lsm_sum_records_kernel<N>, dim3(1), dim3(256), 0, s,
                           a.recB + (size_t)(a.n_steps - 1) * a.n_chunks * N, a.n_chunks, vec_out);
      break;
    }
    case kLsmPhaseStep:  // global moment sums of row t in; decisions at t; local sums of row t-1 out
      a.B_given = vec_in;
      launch_step<D>(a, t, s);
      if (t >= 2)
        hipLaunchKernelGGL(lsm_sum_records_kernel<N>, dim3(1), dim3(256), 0, s,
                           a.recB + (size_t)(t - 1) * a.n_chunks * N, a.n_chunks, vec_out);
      break;
    default: return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

}  // namespace

uint32_t lsm_chunks(uint64_t ntot) { return (uint32_t)((ntot + kLsmChunk - 1) / kLsmChunk); }

// scratch sizes in doubles, for the caller (hh_api.hip) to allocate
size_t lsm_scratch_doubles(uint64_t ntot, uint32_t n_steps, int degree) {
  const size_t rows = (size_t)n_steps + 1, ch = lsm_chunks(ntot);
  const size_t nv = 2 * (size_t)degree + 1;
  // rec_stats [rows][ch][3] | rowstat [rows][3] | rec_pow [rows][ch][nv] | P [rows][nv] |
  // recB [rows][ch][degree+1] | disc_pow [rows] | counters [2]
  return rows * ch * 3 + rows * 3 + rows * ch * nv + rows * nv + rows * ch * (degree + 1) + rows + 2;
}

int launch_gbm_grid(const uint64_t* seeds_dev, uint64_t n_paths, uint32_t n_steps, double S0,
                    double r, double sigma, double T, int anti, double* grid, hipStream_t s) {
  const double dt = T / (double)n_steps;
  const double a = (r - 0.5 * sigma * sigma) * dt, b = sigma * sqrt(dt);
  const dim3 g((unsigned)((n_paths + 255) / 256)), blk(256);
  if (anti)
    hipLaunchKernelGGL(gbm_grid_kernel<true>, g, blk, 0, s, seeds_dev, n_paths, n_steps, S0, a, b,
                       grid);
  else
    hipLaunchKernelGGL(gbm_grid_kernel<false>, g, blk, 0, s, seeds_dev, n_paths, n_steps, S0, a, b,
                       grid);
  return (int)hipGetLastError();
}

// Backward induction on a device-resident grid.  `scratch` has lsm_scratch_doubles() doubles,
// `records` lsm_chunks() x kRecStride; on return `records` holds the per-workgroup Σ, Σ² of the
// discounted stopped values and scratch's last two doubles the regressed / skipped row counts.
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

int launch_lsm(const double* grid, uint64_t ntot, uint32_t n_steps, double strike, double cp,
               double step_discount, int degree, int32_t* tau, double* val, double* scratch,
               double* records, hipStream_t s) {
  if (degree < 1 || degree > kLsmMaxDeg) return (int)hipErrorInvalidValue;
  const LsmLayout L = lsm_layout(scratch, ntot, n_steps, degree);
  hipError_t e = hipMemsetAsync(L.counters, 0, 2 * sizeof(double), s);
  if (e != hipSuccess) return (int)e;
  const dim3 b(256);
  hipLaunchKernelGGL(lsm_stats_kernel, dim3(L.ch1, L.rows), b, 0, s, grid, ntot, strike, cp, L.ch1,
                     L.per_wg, L.rec_stats);
  hipLaunchKernelGGL(lsm_rowstat_kernel, dim3(L.rows), b, 0, s, L.rec_stats, L.ch1, L.rs);
  const LsmStepArgs a = lsm_step_args(L, grid, ntot, n_steps, strike, cp, step_discount, tau, val);
  hipLaunchKernelGGL(lsm_disc_kernel, dim3((L.rows + 255) / 256), b, 0, s, a.ln_disc, n_steps,
                     L.disc_pow);
  int rc = 0;
#define HH_CALL(D) run_lsm<D>(L, a, s)
  HH_LSM_DISPATCH(degree, HH_CALL)
#undef HH_CALL
  if (rc) return rc;
  hipLaunchKernelGGL(lsm_final_kernel, dim3(lsm_chunks(ntot)), b, 0, s, tau, val, ntot, a.ln_disc,
                     records);
  return (int)hipGetLastError();
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
    hipError_t e = hipMemsetAsync(L.counters, 0, 2 * sizeof(double), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(lsm_stats_kernel, dim3(L.ch1, L.rows), b, 0, s, grid, ntot, strike, cp, L.ch1,
                       L.per_wg, L.rec_stats);
    hipLaunchKernelGGL(lsm_sum_records_kernel<3>, dim3(L.rows), b, 0, s, L.rec_stats, L.ch1, vec_out);
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
