#!/usr/bin/env python3
"""Benchmark of the hot path: Heston Euler–Maruyama Monte Carlo, 10^6 paths x 252 steps per GPU
(BASELINE.json metric; SURVEY.md §8d benchmark problem "H252").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete pass of the path over one batch: every rank integrates its 10^6
trajectories x 252 Euler steps from Wiener increments already resident in HBM (REPLAY, the mode the
HBM roofline is quoted for: 16 algorithmic bytes per path-step), reduces the discounted payoff
sums, and (N > 1) all-reduces the 16-double accumulator vector over RCCL.  Paths shard
embarrassingly: weak scaling, 10^6 paths per GPU.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      dominant kernel (euler_kernel REPLAY): algorithmic bytes / HIP-event time vs 8 TB/s
  cpu_baseline  the CPU oracle (oracle/hh_oracle.c, a port) timed on this host on a bounded sample
  generate      the same workload with the increments drawn in-kernel from Philox (VALU-bound)
  price_check   |price - CPU reference| on identical draws (bounded sample of the same buffer)
  other_configs kernel times of BASELINE configs 2, 4, 5 (N = 1 only)
  widened_rows  kernel times of the rows widened beyond the headline path: LSM, exact Heston grid
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PATH_STEP = 16.0   # two fp64 increments read once

# benchmark problem H252
H252 = dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1.0,
            strike=100.0, cp=1.0)
H252_ANALYTIC = 9.242521073959068  # Carr–Madan restatement, SURVEY.md §8c (sanity band only)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 timed steps after 20 warm-ups (0.25 s in all): the first ~25 launches of a fresh process
    # run 3-8 % slower than the steady state (clock / memory-system transient), whichever kernel form
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--paths", type=int, default=1_000_000, help="trajectories per GPU")
    ap.add_argument("--nsteps", type=int, default=252, help="Euler steps per trajectory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with 1 rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import hedgehog_jl_amd as hh
    from hedgehog_jl_amd import _ffi

    ctx = hh.Context(dev.index)  # raises without a HIP device — there is no CPU fallback
    lib, h = ctx.lib, ctx.handle
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)

    n_paths, n_steps = args.paths, args.nsteps
    # seeds[i] = global 1-based trajectory index; shard = contiguous range
    g0 = rank * n_paths
    seeds = torch.arange(g0 + 1, g0 + n_paths + 1, dtype=torch.int64, device=dev)
    model = _ffi.make_model(**H252)

    # synthetic input, generated on the device BEFORE the timed region: correlated increments
    n_el = lib.hh_replay_elems(n_paths, n_steps, _ffi.HH_HESTON)
    dW = torch.empty(n_el, dtype=torch.float64, device=dev)
    ctx.check(lib.hh_wiener_fill(h, _ffi.HH_HESTON, model.rho, model.T, n_steps, n_paths,
                                 seeds.data_ptr(), 1, dW.data_ptr()))
    accum = torch.zeros(_ffi.HH_ACC_LEN, dtype=torch.float64, device=dev)

    def config(noise):
        c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, n_paths, n_steps,
                             noise_mode=noise)
        c.seeds, c.seeds_on_device = seeds.data_ptr(), 1
        c.replay, c.replay_on_device = dW.data_ptr(), 1
        return c

    cfg_rep, cfg_gen = config(_ffi.HH_NOISE_REPLAY), config(_ffi.HH_NOISE_GENERATE)

    # Steps are independent pricing jobs: the (latency-bound, 128-byte) all-reduce of step k is
    # issued asynchronously on RCCL's stream and overlaps the simulation kernel of step k+1; two
    # accumulator buffers alternate, and every all-reduce has completed before the clock stops.
    accums = [accum, torch.zeros_like(accum)]
    pending = [None, None]

    def step(cfg, i):
        b = i & 1
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        ctx.check(lib.hh_mc_accumulate(h, C.byref(model), C.byref(cfg), accums[b].data_ptr(), None))
        if dist is not None:  # the path's one exchange: 16 doubles, SUM
            pending[b] = dist.all_reduce(accums[b], async_op=True)
        return b

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def timed(cfg, k, w):
        last = 0
        for i in range(w):
            step(cfg, i)
        drain()
        ctx.enable_timing(True)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(k):
            last = step(cfg, i)
        drain()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        kern_ms = ctx.read_timings()
        ctx.enable_timing(False)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        accum.copy_(accums[last])
        return dt, kern_ms

    dt_rep, kern_rep = timed(cfg_rep, args.steps, args.warmup)
    acc_rep = accum.cpu().numpy().copy()
    dt_gen, kern_gen = timed(cfg_gen, args.steps, args.warmup)
    acc_gen = accum.cpu().numpy().copy()

    # every collective of the run is behind us: all ranks leave the process group together, here;
    # rank 0 then measures the single-GPU extras (other configs, CPU baseline) on its own
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        return

    total_path_steps = float(world) * n_paths * n_steps
    value = total_path_steps * args.steps / dt_rep
    res = _ffi.hh_result()
    lib.hh_mc_finalize(C.byref(model), C.byref(cfg_rep), acc_rep.ctypes.data, C.byref(res))
    res_gen = _ffi.hh_result()
    lib.hh_mc_finalize(C.byref(model), C.byref(cfg_gen), acc_gen.ctypes.data, C.byref(res_gen))

    kern_s = float(np.mean(kern_rep)) * 1e-3
    achieved = BYTES_PER_PATH_STEP * n_paths * n_steps / kern_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "MC path-steps/sec (Heston Euler-Maruyama, 1e6 paths x 252 steps per GPU)",
        "value": value,
        "unit": "path-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt_rep / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (Philox-generated correlated Wiener increments, resident in HBM)",
        "config": {"workload": "HestonDynamics EulerMaruyama European call H252 "
                               "(configs[2]): %d paths x %d steps per GPU, NoVarianceReduction, "
                               "noise REPLAY" % (n_paths, n_steps),
                   "paths_per_gpu": n_paths, "n_steps": n_steps, "global_paths": world * n_paths,
                   "parallelism": "path-sharded x%d, one 16-double all-reduce" % world},
        "price": res.price,
        "std_error": res.std_error,
        "analytic_carr_madan": H252_ANALYTIC,
        "roofline": {
            "bound": "hbm", "kernel": "euler_kernel<HestonModel,REPLAY>",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": BYTES_PER_PATH_STEP * n_paths * n_steps,
            "kernel_ms_avg": kern_s * 1e3, "kernel_ms_min": float(np.min(kern_rep)),
            "launches_timed": len(kern_rep)},
        "generate": {
            "value": total_path_steps * args.steps / dt_gen, "unit": "path-steps/s",
            "ms_per_step": dt_gen / args.steps * 1e3,
            "kernel_ms_avg": float(np.mean(kern_gen)), "bound": "valu (Philox + Box-Muller, fp64)",
            "price": res_gen.price,
            "rel_diff_vs_replay": abs(res_gen.price - res.price) / abs(res.price)},
    }

    # ---- the other BASELINE.json configurations, a few launches each (rank 0, N = 1 only) -----
    if world == 1 and not args.no_extra:
        def kernel_ms(mdl, cfg, reps=5):
            ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(cfg), accum.data_ptr(), None))
            ctx.enable_timing(True)
            for _ in range(reps):
                ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(cfg), accum.data_ptr(), None))
            t = float(np.median(ctx.read_timings()))
            ctx.enable_timing(False)
            r = _ffi.hh_result()
            a = accum.cpu().numpy().copy()
            lib.hh_mc_finalize(C.byref(mdl), C.byref(cfg), a.ctypes.data, C.byref(r))
            return t, r

        sd = {"S0": [1, 0, 0], "V0": [0, 1, 0], "r_drift": [0, 0, 1],
              "discount": [0, 0, -float(np.exp(-H252["r"] * H252["T"]))]}
        m5 = _ffi.make_model(**H252, seeds=sd, n_partials=3)
        c5 = config(_ffi.HH_NOISE_REPLAY)
        c5.n_partials = 3
        t5, r5 = kernel_ms(m5, c5)
        c4 = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n_paths)
        c4.seeds, c4.seeds_on_device = seeds.data_ptr(), 1
        t4, r4 = kernel_ms(model, c4, reps=3)
        m2 = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0)
        c2 = _ffi.make_config(_ffi.HH_LOGNORMAL, _ffi.HH_EXACT_LAW, n_paths)
        c2.seeds, c2.seeds_on_device = seeds.data_ptr(), 1
        t2, r2 = kernel_ms(m2, c2)
        out["other_configs"] = {
            "config2_lognormal_exact": {"paths_per_s": n_paths / (t2 * 1e-3), "kernel_ms": t2,
                                        "price": r2.price, "analytic": 10.450583572185565},
            "config4_broadie_kaya": {"paths_per_s": n_paths / (t4 * 1e-3), "kernel_ms": t4,
                                     "price": r4.price, "std_error": r4.std_error,
                                     "cf_terms_per_path": r4.bk_cf_terms / n_paths,
                                     "bisect_fallbacks": int(r4.bk_bisect_fallback)},
            "config5_greeks_delta_dV0_rho_replay": {
                "path_steps_per_s": n_paths * n_steps / (t5 * 1e-3), "kernel_ms": t5,
                "hbm_GBs": BYTES_PER_PATH_STEP * n_paths * n_steps / (t5 * 1e-3) / 1e9,
                "greeks": [r5.dprice[k] for k in range(3)],
                "fourier": [0.65565115, 40.7248418, 56.3225943]},
        }

        # the rows widened beyond the headline path: one timing each, same C-ABI
        import math
        n_l, st_l = 1_000_000, 100
        m_l = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0, cp=-1.0)
        c_l = _ffi.make_config(_ffi.HH_LOGNORMAL, _ffi.HH_EXACT_LAW, n_l, st_l, antithetic=1)
        c_l.seeds, c_l.seeds_on_device, c_l.seeds_len = seeds.data_ptr(), 1, n_paths
        r_l = _ffi.hh_lsm_result()
        for _ in range(3):
            ctx.check(lib.hh_lsm_solve(h, C.byref(m_l), C.byref(c_l), 5, math.exp(-0.05 / st_l),
                                       C.byref(r_l), None, None, None))
        n_g, st_g = 200_000, 12
        c_g = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n_g, st_g)
        c_g.seeds, c_g.seeds_on_device, c_g.seeds_len = seeds.data_ptr(), 1, n_paths
        r_g = _ffi.hh_result()
        for _ in range(2):
            ctx.check(lib.hh_heston_exact_grid(h, C.byref(model), C.byref(c_g), None, None, 0,
                                               C.byref(r_g)))
        out["widened_rows"] = {
            "lsm_american_put_2e6_paths_x_100_dates": {"kernel_ms": r_l.kernel_ms, "price": r_l.price,
                                                       "std_error": r_l.std_error},
            "heston_exact_grid_2e5_paths_x_12_dates": {
                "kernel_ms": r_g.kernel_ms, "transitions_per_s": n_g * st_g / (r_g.kernel_ms * 1e-3),
                "cf_terms_per_transition": r_g.bk_cf_terms / (n_g * st_g)},
        }

    # ---- bounded-sample checks against the CPU oracle (rank 0, N = 1 only) ------------------
    if world == 1 and not args.no_cpu_baseline:
        from tests import oracle_ffi  # the ONLY use of the oracle here: checker + timed CPU baseline
        orc = oracle_ffi.load()
        tiles = 200                                   # 51,200 trajectories of the SAME buffer
        ns = min(n_paths, tiles * _ffi.HH_TILE_PATHS)
        n_s_el = lib.hh_replay_elems(ns, n_steps, _ffi.HH_HESTON)
        dW_s = dW[:n_s_el].cpu().numpy()
        c_s = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, ns, n_steps,
                            noise_mode=_ffi.HH_NOISE_REPLAY, replay=dW_s)
        r_gpu = _ffi.hh_result()
        ctx.check(lib.hh_mc_solve(h, C.byref(model), C.byref(c_s), C.byref(r_gpu), None))
        threads = orc.num_threads()
        r_cpu, _, _ = orc.mc_solve(model, c_s, want_terminal=False)  # also warms the CPU
        out["price_check"] = {
            "sample_paths": ns, "gpu": r_gpu.price, "cpu_ref": r_cpu.price,
            "rel_err": abs(r_gpu.price - r_cpu.price) / abs(r_cpu.price),
            "what": "same Wiener increments through the HIP kernel and the CPU oracle"}
        reps, t0 = 0, time.perf_counter()
        while True:
            orc.mc_solve(model, c_s, want_terminal=False)
            reps += 1
            el = time.perf_counter() - t0
            if el >= args.cpu_seconds and reps >= 2:
                break
        out["cpu_baseline"] = {
            "value": reps * ns * n_steps / el, "unit": "path-steps/s", "cores": threads,
            "kind": "port",
            "sample": "%d x (%d paths x %d steps, REPLAY of the same increments), %.1f s, "
                      "oracle/hh_oracle.c with OpenMP over paths" % (reps, ns, n_steps, el)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
