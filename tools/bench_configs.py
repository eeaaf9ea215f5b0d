#!/usr/bin/env python3
"""Throughput of every BASELINE.json configuration on one MI355X (kernel time by HIP events on the
launch stream; inputs resident in HBM).  Prints one JSON object; run on the GPU box:

    python tools/bench_configs.py > gpurun_out/configs.json
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hedgehog_jl_amd as hh  # noqa: E402
from hedgehog_jl_amd import _ffi  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ctx = hh.Context(0)
    lib, h = ctx.lib, ctx.handle
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    N, M = 1_000_000, 252
    seeds = torch.arange(1, N + 1, dtype=torch.int64, device=dev)
    acc = torch.zeros(16, dtype=torch.float64, device=dev)
    out = {}

    def run(tag, model, cfg, units, unit, reps=10, warm=2, extra=None):
        for _ in range(warm):
            ctx.check(lib.hh_mc_accumulate(h, C.byref(model), C.byref(cfg), acc.data_ptr(), None))
        ctx.enable_timing(True)
        for _ in range(reps):
            ctx.check(lib.hh_mc_accumulate(h, C.byref(model), C.byref(cfg), acc.data_ptr(), None))
        t = np.array(ctx.read_timings())
        ctx.enable_timing(False)
        a = acc.cpu().numpy()
        r = _ffi.hh_result()
        lib.hh_mc_finalize(C.byref(model), C.byref(cfg), a.ctypes.data, C.byref(r))
        out[tag] = {"kernel_ms_median": float(np.median(t)), "kernel_ms_min": float(t.min()),
                    "throughput": units / (np.median(t) * 1e-3), "unit": unit, "price": r.price,
                    "std_error": r.std_error}
        if cfg.n_partials:
            out[tag]["dprice"] = [r.dprice[k] for k in range(cfg.n_partials)]
        if extra:
            out[tag].update(extra(r))

    def dev_cfg(c):
        c.seeds, c.seeds_on_device = seeds.data_ptr(), 1
        return c

    # config 1: BS Euler 10^4 x 100 (examples/montecarlo_black_scholes.jl: put, S=K=1, r=.03, σ=.04)
    T366 = 366 / 365
    m1 = _ffi.make_model(S0=1.0, sigma=0.04, r=0.03, T=T366, strike=1.0, cp=-1.0)
    run("config1_bs_euler_1e4x100", m1, dev_cfg(_ffi.make_config(0, 0, 10_000, 100)), 1e4 * 100,
        "path-steps/s")
    # config 2: lognormal exact 10^6 vs BlackScholesAnalytic
    m2 = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0)
    run("config2_lognormal_exact_1e6", m2, dev_cfg(_ffi.make_config(0, 1, N)), N, "paths/s")
    out["config2_lognormal_exact_1e6"]["analytic"] = 10.450583572185565
    # config 3: Heston Euler 10^6 x 252, GENERATE and REPLAY (+ antithetic REPLAY)
    m3 = _ffi.make_model()
    dW = torch.empty(lib.hh_replay_elems(N, M, 1), dtype=torch.float64, device=dev)
    ctx.check(lib.hh_wiener_fill(h, 1, m3.rho, m3.T, M, N, seeds.data_ptr(), 1, dW.data_ptr()))

    def rep_cfg(**kw):
        c = dev_cfg(_ffi.make_config(1, 0, N, M, noise_mode=1, **kw))
        c.replay, c.replay_on_device = dW.data_ptr(), 1
        return c

    run("config3_heston_euler_generate", m3, dev_cfg(_ffi.make_config(1, 0, N, M)), N * M,
        "path-steps/s")
    run("config3_heston_euler_replay", m3, rep_cfg(), N * M, "path-steps/s")
    out["config3_heston_euler_replay"]["hbm_GBs"] = 16e-9 * out["config3_heston_euler_replay"]["throughput"]
    run("config3_heston_euler_replay_antithetic", m3, rep_cfg(antithetic=1), 2 * N * M,
        "integrated path-steps/s (2 per pair)")
    # config 5: (Δ, ∂V0, ρ) fused, 10^6 x 252
    sd = {"S0": [1, 0, 0], "V0": [0, 1, 0], "r_drift": [0, 0, 1],
          "discount": [0, 0, -float(np.exp(-0.03))]}
    m5 = _ffi.make_model(seeds=sd, n_partials=3)
    run("config5_greeks3_generate", m5, dev_cfg(_ffi.make_config(1, 0, N, M, n_partials=3)), N * M,
        "path-steps/s")
    run("config5_greeks3_replay", m5, rep_cfg(n_partials=3), N * M, "path-steps/s")
    out["config5_greeks3_replay"]["hbm_GBs"] = 16e-9 * out["config5_greeks3_replay"]["throughput"]
    out["config5_greeks3_replay"]["fourier_targets"] = [0.65565115, 40.7248418, 56.3225943]
    # the full Heston gradient in one pass (calibration's use of AD): ∂/∂(S0, V0, κ, θ, σ, r) — four
    # carried basis derivatives (V0, κ, θ, σ), two passive directions
    e6 = lambda k: [1.0 if j == k else 0.0 for j in range(6)]
    sd6 = {"S0": e6(0), "V0": e6(1), "kappa": e6(2), "theta": e6(3), "sigma": e6(4), "r_drift": e6(5),
           "discount": [0, 0, 0, 0, 0, -float(np.exp(-0.03))]}
    m6 = _ffi.make_model(seeds=sd6, n_partials=6)
    run("heston_full_gradient6_generate", m6, dev_cfg(_ffi.make_config(1, 0, N, M, n_partials=6)), N * M,
        "path-steps/s")
    run("heston_full_gradient6_replay", m6, rep_cfg(n_partials=6), N * M, "path-steps/s")
    out["heston_full_gradient6_replay"]["hbm_GBs"] = 16e-9 * out["heston_full_gradient6_replay"]["throughput"]
    del dW
    # config 4: Broadie–Kaya 10^6
    c4 = _ffi.make_config(1, 2, N)
    seed0 = torch.tensor([99], dtype=torch.int64, device=dev)
    c4.seeds, c4.seeds_on_device = seed0.data_ptr(), 1
    run("config4_broadie_kaya_1e6", m3, c4, N, "paths/s", reps=5, warm=1,
        extra=lambda r: {"cf_terms_per_path": r.bk_cf_terms / N,
                         "newton_fail": int(r.bk_newton_fail),
                         "bisect_fallback": int(r.bk_bisect_fallback),
                         "maxguess_fallback": int(r.bk_maxguess_fallback),
                         "carr_madan": 9.242521073959068})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
