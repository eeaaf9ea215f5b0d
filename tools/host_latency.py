#!/usr/bin/env python3
"""Wall time of the HOST entry point `solve(prob, MonteCarlo(...))` against the kernel time inside it:
what a caller of the drop-in pays per solve (seed upload, launch, sample download) on BASELINE
configs 1 and 3.  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import datetime as dt

import numpy as np

import hedgehog_jl_amd as hh

ref, exp_ = dt.date(2021, 1, 1), dt.date(2022, 1, 1)


def wall(f, reps):
    f()
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t0) / reps * 1e3, r


heston = hh.PricingProblem(hh.VanillaOption(100.0, exp_, hh.European(), hh.Call(), hh.Spot()),
                           hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7))
bs = hh.PricingProblem(hh.VanillaOption(100.0, exp_, hh.European(), hh.Call(), hh.Spot()),
                       hh.BlackScholesInputs(ref, 0.05, 100.0, 0.2))
for name, prob, dyn, n, m in (("config 1: lognormal Euler 1e4 x 100", bs, hh.LognormalDynamics(), 10_000, 100),
                              ("config 3: Heston Euler 1e6 x 252", heston, hh.HestonDynamics(), 1_000_000, 252)):
    cfg = hh.SimulationConfig(n, steps=m, seeds=np.arange(1, n + 1, dtype=np.uint64))
    method = hh.MonteCarlo(dyn, hh.EulerMaruyama(), cfg)
    for label, f in (("solve(prob, method)", lambda: hh.solve(prob, method)),
                     ("solve(...).price only", lambda: hh.solve(prob, method, ensemble=False)),
                     ("solve + read .ensemble", lambda: (lambda s: (s, s.ensemble[0]))(hh.solve(prob, method))[0])):
        t, sol = wall(f, 20 if n > 100_000 else 200)
        print(f"{name}: {label:24s} {t:8.3f} ms wall  (kernel {sol.result.kernel_ms:.3f} ms, price {sol.price:.6f})",
              flush=True)
