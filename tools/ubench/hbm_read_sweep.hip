// Sweep of the pure-read ceiling (see hbm_read.hip) over workgroup shape, steps per iteration and
// occupancy cap (LDS padding), on the 10^6-path REPLAY buffer (3907 tiles x 252 steps x 2 x 2 KiB).
// Build: hipcc -O3 --offload-arch=gfx950 hbm_read_sweep.hip -o hbm_read_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

// THREADS x (256/THREADS) paths per lane; CH steps per iteration; PAD KiB of LDS per workgroup
template <int THREADS, int CH, int PAD>
__global__ __launch_bounds__(THREADS) void read_kernel(const double* __restrict__ src, uint32_t n_steps,
                                                       double* __restrict__ out) {
  constexpr int PPT = 256 / THREADS;           // 1, 2 or 4 doubles per lane and (step, comp)
  constexpr int NV = PPT >= 2 ? PPT / 2 : 1;   // 16-B loads per lane and (step, comp)
  __shared__ double pad[PAD > 0 ? PAD * 128 : 1];
  const uint32_t tile = blockIdx.x, tid = threadIdx.x;
  const double* base = src + (size_t)tile * n_steps * 2 * 256 + (size_t)tid * PPT;
  double acc = 0.0;
  for (uint32_t s = 0; s + CH <= n_steps; s += CH) {
    if constexpr (PPT == 1) {
      double v[CH * 2];
#pragma unroll
      for (int i = 0; i < CH * 2; ++i)
        v[i] = __builtin_nontemporal_load(base + ((size_t)s * 2 + i) * 256);
#pragma unroll
      for (int i = 0; i < CH * 2; ++i) acc += v[i];
    } else {
      d2 v[CH * 2 * NV];
#pragma unroll
      for (int i = 0; i < CH * 2; ++i)
#pragma unroll
        for (int q = 0; q < NV; ++q)
          v[i * NV + q] = __builtin_nontemporal_load(
              reinterpret_cast<const d2*>(base + ((size_t)s * 2 + i) * 256 + 2 * q));
#pragma unroll
      for (int i = 0; i < CH * 2 * NV; ++i) acc += v[i].x + v[i].y;
    }
  }
  if (PAD > 0 && acc == 1.2345e300) pad[tid] = acc;
  if (acc == 1.2345e300) out[tile * THREADS + tid] = acc + (PAD > 0 ? pad[tid ^ 1] : 0.0);
}

template <int THREADS, int CH, int PAD>
static void run(const double* src, double* out, uint32_t n_tiles, uint32_t n_steps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  double mean = 0;
  int cnt = 0;
  for (int it = 0; it < 50; ++it) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((read_kernel<THREADS, CH, PAD>), dim3(n_tiles), dim3(THREADS), 0, 0, src, n_steps, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float t;
    (void)hipEventElapsedTime(&t, e0, e1);
    if (it >= 15) { mean += t; ++cnt; }
  }
  mean /= cnt;
  const double bytes = (double)n_tiles * (n_steps / CH * CH) * 2 * 256 * 8;
  const int waves = THREADS / 64;
  const int wg_per_cu = PAD > 0 ? (160 / (PAD + 1) < 32 / waves ? 160 / (PAD + 1) : 32 / waves) : 32 / waves;
  printf("threads %3d  steps/iter %d  pad %2d KiB (~%2d waves/CU, %4.1f KiB in flight/wave): %.4f ms %5.0f GB/s\n",
         THREADS, CH, PAD, wg_per_cu * waves, CH * 2 * 2048.0 / waves / 1024.0, mean, bytes / mean / 1e6);
}

int main() {
  const uint32_t n_steps = 252, n_tiles = 3907;
  const size_t n = (size_t)n_tiles * n_steps * 2 * 256;
  double *src = nullptr, *out = nullptr;
  if (hipMalloc(&src, n * sizeof(double)) != hipSuccess) return 1;
  (void)hipMalloc(&out, (size_t)n_tiles * 256 * sizeof(double));
  (void)hipMemset(src, 0, n * sizeof(double));
#define R(T, C, P) run<T, C, P>(src, out, n_tiles, n_steps)
  R(128, 2, 0); R(128, 2, 20); R(128, 2, 24); R(128, 2, 28); R(128, 2, 32); R(128, 2, 40); R(128, 2, 52);
  R(128, 1, 32); R(128, 4, 32); R(128, 4, 40); R(128, 4, 52); R(128, 6, 52);
  R(64, 1, 20); R(64, 2, 20); R(64, 2, 16); R(64, 2, 26); R(64, 4, 20); R(64, 4, 26);
  R(256, 2, 0); R(256, 2, 40); R(256, 4, 40); R(256, 4, 52); R(256, 6, 79);
  return 0;
}
