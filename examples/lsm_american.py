#!/usr/bin/env python3
"""American put by Longstaff–Schwartz on the HIP path, against the CRR tree — the comparison of
/root/reference/test/agreement/american_options.jl.  Needs an MI355X."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hedgehog_jl_amd as hh  # noqa: E402

ref = hh.Date(2020, 1, 1)
expiry = hh.add_years(ref, 1)
prob = hh.PricingProblem(hh.VanillaOption(100.0, expiry, hh.American(), hh.Put(), hh.Spot()),
                         hh.BlackScholesInputs(ref, 0.05, 100.0, 0.2))
n = 500_000
cfg = hh.SimulationConfig(n, steps=100, seeds=np.arange(1, n + 1),
                          variance_reduction=hh.Antithetic())
sol = hh.solve(prob, hh.LSM(hh.LognormalDynamics(), hh.BlackScholesExact(), cfg, 5))
T = hh.yearfrac(ref, expiry)
print(f"LSM  {sol.price:.5f} +- {sol.std_error:.5f}  ({sol.result.kernel_ms:.2f} ms for "
      f"{sol.result.n_paths_total} paths x 100 dates)")
print("CRR  6.09711  (2000-step Cox-Ross-Rubinstein tree, T = 366/365; computed by tests' oracle/analytic.py)")
tau, val = sol.stopping_info
print("exercised early on", f"{np.mean(tau < 100) * 100:.1f}% of the paths")
