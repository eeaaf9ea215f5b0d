// Micro-benchmark: issue cost of the instruction classes of the GENERATE loop on gfx950 —
// v_fma_f64, v_mad_u64_u32, v_mul_hi_u32 + v_mul_lo_u32, v_xor_b32 — as wave-instructions per clock
// per SIMD, with enough independent chains and waves to saturate the pipe.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITERS = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, double seed) {
  uint32_t a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 7, a3 = a1 + 11;
  uint64_t m0 = a0, m1 = a1, m2 = a2, m3 = a3;
  double d0 = seed + threadIdx.x, d1 = d0 * 1.0000001, d2 = d0 * 0.9999999, d3 = d0 + 0.5;
  for (int i = 0; i < ITERS; ++i) {
    if (KIND == 0) {  // 4 independent v_fma_f64
      d0 = fma(d0, 1.0000000001, 1e-9); d1 = fma(d1, 0.9999999999, 1e-9);
      d2 = fma(d2, 1.0000000002, 1e-9); d3 = fma(d3, 0.9999999998, 1e-9);
    } else if (KIND == 1) {  // 4 independent v_mad_u64_u32
      m0 = (uint64_t)(uint32_t)m0 * 0xD2511F53u + (m0 >> 32);
      m1 = (uint64_t)(uint32_t)m1 * 0xCD9E8D57u + (m1 >> 32);
      m2 = (uint64_t)(uint32_t)m2 * 0xD2511F53u + (m2 >> 32);
      m3 = (uint64_t)(uint32_t)m3 * 0xCD9E8D57u + (m3 >> 32);
    } else if (KIND == 2) {  // 4 x (mul_hi + mul_lo)
      a0 = __umulhi(a0, 0xD2511F53u) ^ (a0 * 0xD2511F53u);
      a1 = __umulhi(a1, 0xCD9E8D57u) ^ (a1 * 0xCD9E8D57u);
      a2 = __umulhi(a2, 0xD2511F53u) ^ (a2 * 0xD2511F53u);
      a3 = __umulhi(a3, 0xCD9E8D57u) ^ (a3 * 0xCD9E8D57u);
    } else {  // 4 independent xor/add pairs
      a0 = (a0 ^ a1) + 0x9e3779b9u; a1 = (a1 ^ a2) + 0xbb67ae85u;
      a2 = (a2 ^ a3) + 0x9e3779b9u; a3 = (a3 ^ a0) + 0xbb67ae85u;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = m0 + m1 + m2 + m3 + a0 + a1 + a2 + a3 +
                                       (uint64_t)(d0 + d1 + d2 + d3);
}

template <int KIND>
double run(const char* name, int insts_per_iter) {
  uint64_t* out;
  const int blocks = 256 * 8;  // 8 waves per SIMD
  hipMalloc(&out, blocks * 256 * sizeof(uint64_t));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_insts = (double)blocks * 4 * ITERS * insts_per_iter;
  const double per_simd_per_us = wave_insts / 1024.0 / (ms * 1e3);
  printf("%-28s %8.3f ms  %7.1f wave-instr/us/SIMD  (= %5.2f cycles each at 2.4 GHz)\n", name, ms,
         per_simd_per_us, 2400.0 / per_simd_per_us);
  hipFree(out);
  return ms;
}

int main() {
  run<0>("v_fma_f64", 4);
  run<1>("v_mad_u64_u32", 4);
  run<2>("v_mul_hi_u32+v_mul_lo_u32+xor", 12);
  run<3>("v_xor_b32+v_add_u32", 8);
  return 0;
}
