#!/usr/bin/env python3
"""Wall time of the phased (sharded) LSM sequence on ONE rank against the fused hh_lsm_solve: what
the cuts themselves cost (extra launches, host round trips) before any collective is added.
GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hedgehog_jl_amd as hh

ref = hh.Date(2020, 1, 1)
payoff = hh.VanillaOption(100.0, hh.add_years(ref, 1), hh.American(), hh.Put(), hh.Spot())
prob = hh.PricingProblem(payoff, hh.BlackScholesInputs(ref, 0.05, 100.0, 0.2))
for n in (10_000, 200_000, 1_000_000):
    cfg = hh.SimulationConfig(n, steps=100, seeds=np.arange(1, n + 1, dtype=np.uint64),
                              variance_reduction=hh.Antithetic())
    method = hh.LSM(hh.LognormalDynamics(), hh.BlackScholesExact(), cfg, 5)
    out = []
    for f in (lambda: hh.solve(prob, method, stopping_info=False), lambda: hh.solve_lsm_sharded(prob, method)):
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sol = f()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        out.append((float(np.median(ts[2:])), sol.price))
    print(f"n={n:8d} x2: fused {out[0][0]:.3f} ms, phased {out[1][0]:.3f} ms "
          f"(+{(out[1][0] - out[0][0]) / 99 * 1e3:.1f} us per exercise date), same price: {out[0][1] == out[1][1]}")
