// Device-side pin of fm::sqrt_lean (hedgehog.jl_amd/csrc/hh_math.h): prints, for a list of arguments, the bits
// of sqrt_lean(w) and of sqrt(w) as the DEVICE computes them, then the largest ulp distance over 10^6 random
// arguments in [2^-767, 2^1000].  tests/test_gpu_math_device.py holds the output against the contract the
// header states.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "hh_math.h"

__global__ void probe(const double* w, double* lean, double* ref, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  lean[i] = hh::fm::sqrt_lean(w[i]);
  ref[i] = sqrt(w[i]);
}

static uint64_t bits(double x) {
  uint64_t b;
  memcpy(&b, &x, 8);
  return b;
}

int main() {
  std::vector<double> w = {0.0,    4.9406564584124654e-324, 1e-310, 0x1p-1022, 0x1p-768, 0x1p-767, 0x1.8p-767, 1e-200,
                           0.25,   1.0,  2.0,  3.0,  1e300,  0x1.fffffffffffffp+1023, INFINITY, NAN, -1.0, -0.0, -1e-300};
  const int n_named = (int)w.size();
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < 1000000; ++i) {  // mantissa and exponent uniformly over [2^-767, 2^1000)
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const uint64_t e = 1023 - 767 + (s >> 11) % (767 + 1000);
    const uint64_t b = (e << 52) | (s & 0xFFFFFFFFFFFFFull);
    double x;
    memcpy(&x, &b, 8);
    w.push_back(x);
  }
  const int n = (int)w.size();
  double *dw, *dl, *dr;
  if (hipMalloc(&dw, n * 8) != hipSuccess || hipMalloc(&dl, n * 8) != hipSuccess || hipMalloc(&dr, n * 8) != hipSuccess) return 2;
  if (hipMemcpy(dw, w.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess) return 3;
  hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, dw, dl, dr, n);
  std::vector<double> l(n), r(n);
  if (hipMemcpy(l.data(), dl, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
  if (hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
  for (int i = 0; i < n_named; ++i)
    printf("arg %016llx lean %016llx sqrt %016llx\n", (unsigned long long)bits(w[i]), (unsigned long long)bits(l[i]),
           (unsigned long long)bits(r[i]));
  long long worst = 0;
  for (int i = n_named; i < n; ++i) {
    const long long d = (long long)bits(l[i]) - (long long)bits(r[i]);
    if (llabs(d) > worst) worst = llabs(d);
  }
  printf("random_worst_ulp %lld\n", worst);
  return 0;
}
