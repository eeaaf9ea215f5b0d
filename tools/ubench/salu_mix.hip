// Micro-benchmark: what the scalar moves of a literal-constant Horner chain cost on gfx950.
// hh_math.h's fma_c() reads its fp64 literal from an SGPR pair, which the compiler fills with two s_mov_b32
// right before the v_fma_f64 (hh_bk.hip is built without machine LICM, so nothing is hoisted).  A SIMD issues at
// most one VALU and one scalar instruction per 4-cycle slot, from different waves: is a chain of
// (s_mov, s_mov, v_fma_f64) bound by the scalar unit?  Variants per Horner step:
//   0: v_fma_f64 alone (constant resident)     1: 2 s_mov_b32 + v_fma_f64     2: 1 s_mov_b32 + v_fma_f64
//   3: as 1 with two independent chains        4: the constants by one s_load_dwordx16 per 8 steps
// at 4 and 8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 salu_mix.hip -o salu_mix ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITERS = 2048;
constexpr int STEPS = 16;  // Horner steps per iteration

#define STEP0(d, z) asm volatile("v_fma_f64 %0, %0, %1, s[20:21]" : "+v"(d) : "v"(z) : "s20", "s21");
#define STEP1(d, z, lo, hi)                                                                              \
  asm volatile("s_mov_b32 s20, " #lo "\n s_mov_b32 s21, " #hi "\n v_fma_f64 %0, %0, %1, s[20:21]" \
               : "+v"(d) : "v"(z) : "s20", "s21");
#define STEP2(d, z, hi) \
  asm volatile("s_mov_b32 s21, " #hi "\n v_fma_f64 %0, %0, %1, s[20:21]" : "+v"(d) : "v"(z) : "s20", "s21");

template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, double seed, const double* __restrict__ tab) {
  double d = seed + threadIdx.x * 1e-9, e = d * 0.5, z = 1e-3 * seed;
  asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x3fc55555" ::: "s20", "s21");
  for (int i = 0; i < ITERS; ++i) {
    if (KIND == 0) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) { STEP0(d, z) }
    } else if (KIND == 1) {
#pragma unroll
      for (int s = 0; s < STEPS / 4; ++s) {
        STEP1(d, z, 0x11111111, 0x3f811111) STEP1(d, z, 0x16c16c17, 0x3f56c16c)
        STEP1(d, z, 0x1a01a01a, 0x3f2a01a0) STEP1(d, z, 0xa556c734, 0x3ec71de3)
      }
    } else if (KIND == 2) {
#pragma unroll
      for (int s = 0; s < STEPS / 4; ++s) {
        STEP2(d, z, 0x3f811111) STEP2(d, z, 0x3f56c16c) STEP2(d, z, 0x3f2a01a0) STEP2(d, z, 0x3ec71de3)
      }
    } else if (KIND == 3) {
#pragma unroll
      for (int s = 0; s < STEPS / 4; ++s) {
        STEP1(d, z, 0x11111111, 0x3f811111) STEP1(e, z, 0x16c16c17, 0x3f56c16c)
        STEP1(d, z, 0x1a01a01a, 0x3f2a01a0) STEP1(e, z, 0xa556c734, 0x3ec71de3)
      }
    } else {
      // 8 constants by one scalar load, waited for, then 8 steps; twice
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        asm volatile(
            "s_load_dwordx16 s[36:51], %2, 0x0\n s_waitcnt lgkmcnt(0)\n"
            "v_fma_f64 %0, %0, %1, s[36:37]\n v_fma_f64 %0, %0, %1, s[38:39]\n"
            "v_fma_f64 %0, %0, %1, s[40:41]\n v_fma_f64 %0, %0, %1, s[42:43]\n"
            "v_fma_f64 %0, %0, %1, s[44:45]\n v_fma_f64 %0, %0, %1, s[46:47]\n"
            "v_fma_f64 %0, %0, %1, s[48:49]\n v_fma_f64 %0, %0, %1, s[50:51]\n"
            : "+v"(d)
            : "v"(z), "s"(tab)
            : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49",
              "s50", "s51");
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = d + e;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
  double *out, *tab;
  const int blocks = 256 * waves_per_simd;
  hipMalloc(&out, blocks * 256 * sizeof(double));
  hipMalloc(&tab, 64 * sizeof(double));
  double h[64];
  for (int i = 0; i < 64; ++i) h[i] = 1.0 / (i + 2);
  hipMemcpy(tab, h, sizeof h, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0, tab);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0, tab);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fmas = (double)blocks * 4 * ITERS * STEPS;
  const double per_simd_per_us = fmas / 1024.0 / (ms * 1e3);
  printf("%-44s %d waves/SIMD %8.3f ms  %6.1f fma/us/SIMD (= %5.2f cycles per step at 2.4 GHz)\n", name,
         waves_per_simd, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
  hipFree(out); hipFree(tab);
}

int main() {
  for (int w : {4, 8}) {
    run<0>("v_fma_f64, constant resident", w);
    run<1>("2 s_mov_b32 + v_fma_f64", w);
    run<2>("1 s_mov_b32 + v_fma_f64", w);
    run<3>("2 s_mov_b32 + v_fma_f64, two chains", w);
    run<4>("s_load_dwordx16 + wait + 8 v_fma_f64", w);
  }
  return 0;
}
