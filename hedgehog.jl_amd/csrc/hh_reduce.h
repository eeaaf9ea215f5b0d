// The fixed summation order of the record reduction (mean(payoffs), montecarlo.jl:490), shared by every kernel that
// adds workgroup records: reduce_records_kernel (hh_kernels.hip), the reducers folded into the simulation kernels
// (hh_sim.h, finish_records) and the Broadie–Kaya tail kernel (hh_bk.hip).  Internal.
#pragma once
#include <hip/hip_runtime.h>

#include "hh_kernels.h"

namespace hh {

// The binary tree of sum_slot() over 256 partial sums, by ONE wave without a barrier: the steps 128 and 64
// on four LDS words per lane, the rest by shuffles — the same adds in the same order.  Result in every lane.
__device__ __forceinline__ double tree256(const double* __restrict__ p) {
  const int l = threadIdx.x & 63;
  double a = p[l] + p[l + 128];
  const double b = p[l + 64] + p[l + 192];
  a += b;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  return __shfl(a, 0, 64);
}

__device__ __forceinline__ double sum_slot(const double* __restrict__ rec, uint32_t n, int slot,
                                           double* sm) {
  const int tid = threadIdx.x;
  double t = 0.0;
  // Records b = tid, tid + 256, … added in that order.  The kernel is pure latency — a strided load per add:
  // SIXTEEN loads are in flight before the first add, so that the 3907 records of 10^6 trajectories are ONE
  // round trip per thread (eight made it 6.5 µs; one dependent load per add 8 µs); + 0.0 leaves a sum unchanged.
  for (uint32_t b = tid; b < n; b += 256 * 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const uint32_t i = b + 256u * u;
      v[u] = i < n ? rec[(size_t)i * kRecStride + slot] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) t += v[u];
  }
  __syncthreads();  // sm may still be read from the slot before
  sm[tid] = t;
  __syncthreads();
  // the binary tree over the 256 partial sums — steps 128 and 64 on LDS, the rest by shuffles in wave 0
  // (tree256, hh_sim.h: the same adds in the same order as eight barrier-separated LDS steps)
  double r = 0.0;
  if (tid < 64) r = tree256(sm);
  if (tid == 0) sm[256] = r;
  __syncthreads();
  return sm[256];
}

}  // namespace hh
