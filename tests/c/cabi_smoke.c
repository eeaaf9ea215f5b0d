/*
 * Plain-C client of libhedgehog_mc.so: exercises the boundary exactly as a foreign host (Julia's
 * ccall, cgo, …) would — no Python, no torch, only include/hedgehog_mc.h.  Built with gcc and run on
 * the GPU box by tests/test_gpu_cabi_c.py.  Prints "OK <price> <std_error>" on success.
 *
 * Checks, with library-owned device memory only (hh_device_malloc / hh_memcpy_*):
 *   - GENERATE solve == REPLAY solve of hh_wiener_fill's buffer (same draws)
 *   - accumulate + finalize split == hh_mc_solve
 *   - a fused 3-partial Greek pass returns the same price
 *   - error codes and hh_last_error for bad arguments
 *   - exact Heston grid -> LSM on the device grid; the phased (sharded) LSM with one rank == fused
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/hedgehog_mc.h"

#define CHECK(cond, msg)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      fprintf(stderr, "FAIL %s (%s:%d)\n", msg, __FILE__, __LINE__); \
      return 1;                                            \
    }                                                      \
  } while (0)

int main(void) {
  hh_ctx* ctx = NULL;
  int rc = hh_ctx_create(&ctx, 0);
  if (rc != HH_OK) {
    fprintf(stderr, "hh_ctx_create failed (%d): no HIP device — there is no CPU fallback\n", rc);
    return 2;
  }
  CHECK(hh_abi_version() == HH_ABI_VERSION, "abi version");

  const uint64_t N = 200000;
  const uint32_t M = 64;
  uint64_t* seeds = (uint64_t*)malloc(N * sizeof(uint64_t));
  for (uint64_t i = 0; i < N; ++i) seeds[i] = i + 1;

  hh_model m;
  memset(&m, 0, sizeof(m));
  m.S0 = 100; m.V0 = 0.04; m.kappa = 2; m.theta = 0.04; m.sigma = 0.3; m.rho = -0.7;
  m.r_drift = 0.03; m.T = 1.0; m.discount = exp(-0.03); m.strike = 100; m.cp = 1;
  hh_config c;
  memset(&c, 0, sizeof(c));
  c.dynamics = HH_HESTON; c.strategy = HH_EULER_MARUYAMA; c.em_split = 1;
  c.n_paths = N; c.n_steps = M; c.seeds = seeds;

  hh_result gen, rep, split, greeks;
  CHECK(hh_mc_solve(ctx, &m, &c, &gen, NULL) == HH_OK, hh_last_error(ctx));
  CHECK(gen.n_paths_done == N && gen.price > 8.5 && gen.price < 10.0, "price band");

  /* same draws through a device-resident REPLAY buffer owned by the library's allocator */
  void *d_seeds = NULL, *d_dw = NULL, *d_acc = NULL;
  const size_t n_el = hh_replay_elems(N, M, HH_HESTON);
  CHECK(hh_device_malloc(ctx, N * sizeof(uint64_t), &d_seeds) == HH_OK, hh_last_error(ctx));
  CHECK(hh_device_malloc(ctx, n_el * sizeof(double), &d_dw) == HH_OK, hh_last_error(ctx));
  CHECK(hh_device_malloc(ctx, HH_ACC_LEN * sizeof(double), &d_acc) == HH_OK, hh_last_error(ctx));
  CHECK(hh_memcpy_h2d(ctx, d_seeds, seeds, N * sizeof(uint64_t)) == HH_OK, hh_last_error(ctx));
  CHECK(hh_wiener_fill(ctx, HH_HESTON, m.rho, m.T, M, N, (const uint64_t*)d_seeds, 1,
                       (double*)d_dw) == HH_OK, hh_last_error(ctx));
  hh_config cr = c;
  cr.noise_mode = HH_NOISE_REPLAY; cr.replay = (const double*)d_dw; cr.replay_on_device = 1;
  CHECK(hh_mc_solve(ctx, &m, &cr, &rep, NULL) == HH_OK, hh_last_error(ctx));
  CHECK(fabs(rep.price - gen.price) <= 1e-13 * gen.price, "REPLAY == GENERATE");

  /* accumulate (device accumulators, no sync) + finalize on the host */
  double acc[HH_ACC_LEN];
  CHECK(hh_mc_accumulate(ctx, &m, &cr, (double*)d_acc, NULL) == HH_OK, hh_last_error(ctx));
  CHECK(hh_memcpy_d2h(ctx, acc, d_acc, sizeof(acc)) == HH_OK, hh_last_error(ctx));
  CHECK(hh_mc_finalize(&m, &cr, acc, &split) == HH_OK, "finalize");
  CHECK(split.price == rep.price && split.std_error == rep.std_error, "split == solve");

  /* three dual partials in one pass: dS0, dV0, dr (drift and discount) */
  const double dS0[3] = {1, 0, 0}, dV0[3] = {0, 1, 0}, dr[3] = {0, 0, 1};
  const double dD[3] = {0, 0, -m.T * m.discount};
  hh_model mg = m;
  mg.dS0 = dS0; mg.dV0 = dV0; mg.dr_drift = dr; mg.ddiscount = dD;
  hh_config cg = c;
  cg.n_partials = 3;
  CHECK(hh_mc_solve(ctx, &mg, &cg, &greeks, NULL) == HH_OK, hh_last_error(ctx));
  CHECK(fabs(greeks.price - gen.price) <= 1e-13 * gen.price, "greeks pass price");
  CHECK(greeks.dprice[0] > 0.5 && greeks.dprice[0] < 0.8, "delta band");
  CHECK(greeks.dprice[1] > 30 && greeks.dprice[1] < 50, "dV0 band");
  CHECK(greeks.dprice[2] > 45 && greeks.dprice[2] < 65, "rho band");

  /* error behaviour */
  hh_config bad = c;
  bad.n_paths = 0;
  CHECK(hh_mc_solve(ctx, &m, &bad, &gen, NULL) == HH_ERR_INVALID, "n_paths = 0");
  bad = c;
  bad.strategy = HH_EXACT_LAW; /* Heston has no closed-form marginal law in the reference */
  CHECK(hh_mc_solve(ctx, &m, &bad, &gen, NULL) == HH_ERR_UNSUPPORTED, "unsupported pair");
  CHECK(strlen(hh_last_error(ctx)) > 0, "error text");
  CHECK(hh_mc_solve(NULL, &m, &c, &gen, NULL) == HH_ERR_INVALID, "NULL ctx");

  /* per-date exact Heston paths left in device memory, LSM on that grid, and the same American
     put through the sharded (phased) sequence with ONE rank: bit-identical to the fused solve */
  {
    const uint64_t NG = 4000;
    const uint32_t MG = 6;
    hh_config cgd = c;
    cgd.strategy = HH_BROADIE_KAYA;
    cgd.n_paths = NG;
    cgd.n_steps = MG;
    cgd.seeds_len = N;
    hh_model mp = m;
    mp.cp = -1.0;
    void *d_spot = NULL, *d_x = NULL, *d_acc2 = NULL;
    const size_t gel = hh_lsm_grid_elems(NG, MG, 0);
    CHECK(hh_device_malloc(ctx, gel * sizeof(double), &d_spot) == HH_OK, hh_last_error(ctx));
    hh_result gr;
    CHECK(hh_heston_exact_grid(ctx, &mp, &cgd, (double*)d_spot, NULL, 1, &gr) == HH_OK,
          hh_last_error(ctx));
    CHECK(gr.n_paths_done == NG && gr.bk_cf_terms > 0, "grid counters");
    const double D = exp(-mp.r_drift * mp.T / MG);
    hh_lsm_result fused, ongrid, phased;
    CHECK(hh_lsm_solve(ctx, &mp, &cgd, 3, D, &fused, NULL, NULL, NULL) == HH_OK, hh_last_error(ctx));
    CHECK(hh_lsm_solve_grid(ctx, &mp, (const double*)d_spot, NG, MG, 3, D, &ongrid, NULL, NULL) == HH_OK,
          hh_last_error(ctx));
    CHECK(ongrid.price == fused.price && ongrid.std_error == fused.std_error, "LSM on a caller grid");
    const size_t xel = hh_lsm_shard_xchg_elems(MG, 3);
    CHECK(hh_device_malloc(ctx, xel * sizeof(double), &d_x) == HH_OK, hh_last_error(ctx));
    CHECK(hh_device_malloc(ctx, HH_ACC_LEN * sizeof(double), &d_acc2) == HH_OK, hh_last_error(ctx));
    CHECK(hh_lsm_shard_begin(ctx, &mp, &cgd, 3, D, (double*)d_x) == HH_OK, hh_last_error(ctx));
    CHECK(hh_lsm_shard_phase(ctx, HH_LSM_PHASE_POW, 0, (const double*)d_x, (double*)d_x) == HH_OK,
          hh_last_error(ctx));
    CHECK(hh_lsm_shard_phase(ctx, HH_LSM_PHASE_INIT, 0, (const double*)d_x, (double*)d_x) == HH_OK,
          hh_last_error(ctx));
    for (uint32_t t = MG - 1; t >= 1; --t)
      CHECK(hh_lsm_shard_phase(ctx, HH_LSM_PHASE_STEP, t, (const double*)d_x, (double*)d_x) == HH_OK,
            hh_last_error(ctx));
    uint32_t regressed = 0, skipped = 0;
    CHECK(hh_lsm_shard_finish(ctx, (double*)d_acc2, NULL, NULL, NULL, &regressed, &skipped) == HH_OK,
          hh_last_error(ctx));
    double acc2[HH_ACC_LEN];
    CHECK(hh_memcpy_d2h(ctx, acc2, d_acc2, sizeof(acc2)) == HH_OK, hh_last_error(ctx));
    CHECK(hh_lsm_finalize(acc2, &phased) == HH_OK, "lsm finalize");
    CHECK(phased.price == fused.price && phased.std_error == fused.std_error, "phased LSM == fused");
    CHECK(regressed + skipped == MG - 1 && phased.n_paths_total == NG, "phased LSM counters");
    CHECK(hh_lsm_shard_phase(ctx, HH_LSM_PHASE_POW, 0, (const double*)d_x, (double*)d_x) ==
              HH_ERR_INVALID, "phase without begin");
    CHECK(hh_device_free(ctx, d_spot) == HH_OK && hh_device_free(ctx, d_x) == HH_OK &&
          hh_device_free(ctx, d_acc2) == HH_OK, "free");
  }

  /* several devices behind ONE call (hh_mgpu_*), as a Julia host would drive them: no torch in this
     process, so RCCL is whatever dlopen("librccl.so.1") finds.  One device + HH_MGPU_RCCL runs the
     library's own communicator and ncclAllReduce; device 0 listed twice is refused by RCCL and takes
     the host ordered sum — same sums as the single solve up to the order of the last addition. */
  {
    const int one[1] = {0}, twice[2] = {0, 0};
    hh_mgpu *mg1 = NULL, *mg2 = NULL;
    hh_result r1, r2, r3;
    CHECK(hh_mc_solve(ctx, &m, &c, &gen, NULL) == HH_OK, hh_last_error(ctx));
    rc = hh_mgpu_create(&mg1, one, 1, HH_MGPU_RCCL);
    if (rc == HH_OK) {
      CHECK(hh_mgpu_reduce_mode(mg1) == HH_MGPU_REDUCE_RCCL && hh_mgpu_n_devices(mg1) == 1, "rccl mode");
      CHECK(hh_mgpu_solve(mg1, &m, &c, &r1, NULL) == HH_OK, hh_mgpu_last_error(mg1));
      CHECK(r1.price == gen.price && r1.sumsq_payoff == gen.sumsq_payoff, "one device over RCCL == single solve");
      hh_mgpu_destroy(mg1);
    } else {
      CHECK(rc == HH_ERR_RCCL, "RCCL required: only HH_ERR_RCCL may refuse");  /* a host without librccl */
    }
    CHECK(hh_mgpu_create(&mg2, twice, 2, HH_MGPU_AUTO) == HH_OK, "mgpu create");
    CHECK(hh_mgpu_reduce_mode(mg2) == HH_MGPU_REDUCE_HOST, "duplicate device: host ordered sum");
    CHECK(hh_mgpu_solve(mg2, &m, &c, &r2, NULL) == HH_OK, hh_mgpu_last_error(mg2));
    CHECK(r2.n_paths_done == N && fabs(r2.price - gen.price) <= 1e-13 * gen.price &&
          fabs(r2.sumsq_payoff - gen.sumsq_payoff) <= 1e-13 * gen.sumsq_payoff, "two shards == single solve");
    /* device-resident shards: the REPLAY buffer of above, cut at a tile boundary */
    uint64_t a0, b0, a1, b1;
    hh_mgpu_shard_range(N, 2, 0, 1, &a0, &b0);
    hh_mgpu_shard_range(N, 2, 1, 1, &a1, &b1);
    CHECK(a0 == 0 && b0 == a1 && b1 == N && a1 % HH_TILE_PATHS == 0, "shard ranges");
    hh_config cs[2] = {cr, cr};
    cs[0].n_paths = b0 - a0;
    cs[1].n_paths = b1 - a1;
    cs[1].replay = (const double*)d_dw + (size_t)(a1 / HH_TILE_PATHS) * M * 2 * HH_TILE_PATHS;
    CHECK(hh_mgpu_solve_shards(mg2, &m, cs, &r3, NULL) == HH_OK, hh_mgpu_last_error(mg2));
    CHECK(fabs(r3.price - rep.price) <= 1e-13 * rep.price, "device-resident shards == single REPLAY solve");
    CHECK(hh_mgpu_ctx(mg2, 2) == NULL && hh_mgpu_ctx(mg2, 1) != NULL, "per-device contexts");
    hh_mgpu_destroy(mg2);
    const int bad_ids[2] = {0, 4096};
    CHECK(hh_mgpu_create(&mg2, bad_ids, 2, HH_MGPU_AUTO) == HH_ERR_INVALID && mg2 == NULL, "bad ordinal");
  }

  CHECK(hh_device_free(ctx, d_seeds) == HH_OK && hh_device_free(ctx, d_dw) == HH_OK &&
        hh_device_free(ctx, d_acc) == HH_OK, "free");
  hh_ctx_destroy(ctx);
  free(seeds);
  printf("OK %.10f %.6f\n", rep.price, rep.std_error);
  return 0;
}
