// Host-side known-answer check of hedgehog.jl_amd/csrc/hh_rng.h: the Philox4x32-10 the kernels draw with — the same
// source, compiled for the host (its __host__ __device__ functions) — against the Random123 known-answer vectors,
// and the uniform of u01_from_bits at its two ends.  No HIP call is made: this runs on a machine without a GPU
// (tests/test_rng_host.py; under AddressSanitizer + UBSan by `make test-cpu-asan`, host code only).
#include <cstdint>
#include <cstdio>

#include "hh_rng.h"

static int bad = 0;
static void kat(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t w0,
                uint32_t w1, uint32_t w2, uint32_t w3) {
  const hh::Philox4 r = hh::philox4x32_10(c0, c1, c2, c3, k0, k1);
  const bool ok = r.c0 == w0 && r.c1 == w1 && r.c2 == w2 && r.c3 == w3;
  std::printf("philox %08x %08x %08x %08x key %08x %08x -> %08x %08x %08x %08x %s\n", c0, c1, c2, c3, k0, k1, r.c0,
              r.c1, r.c2, r.c3, ok ? "ok" : "MISMATCH");
  bad += !ok;
}

int main() {
  // Random123 kat_vectors, philox4x32 10 rounds
  kat(0u, 0u, 0u, 0u, 0u, 0u, 0x6627e8d5u, 0xe169c58du, 0xbc57ac4cu, 0x9b00dbd8u);
  kat(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x408f276du, 0x41c83b0eu,
      0xa20bc7c6u, 0x6d5451fdu);
  kat(0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u, 0xa4093822u, 0x299f31d0u, 0xd16cfe09u, 0x94fdccebu,
      0x5001e420u, 0x24126ea1u);
  const double lo = hh::u01_from_bits(0u, 0u), hi = hh::u01_from_bits(0xffffffffu, 0xffffffffu);
  const bool ends = lo == 0x1p-53 && hi == 1.0 - 0x1p-53;
  std::printf("u01 ends %a %a %s\n", lo, hi, ends ? "ok" : "MISMATCH");
  bad += !ends;
  // every counter of a short stream: distinct blocks (a stuck round function would repeat)
  uint64_t acc = 0;
  for (uint32_t i = 0; i < 100000u; ++i) {
    const hh::Philox4 r = hh::philox4x32_10(i, 0u, 0u, hh::kDomEuler, 42u, 7u);
    acc += r.c0 ^ r.c3;
  }
  std::printf("stream checksum %llu\n", (unsigned long long)acc);
  return bad ? 1 : 0;
}
