#!/usr/bin/env python3
"""Heston calibration to 51 Carr–Madan quotes — the scenario of the reference's own test
(/root/reference/test/unit/calibration.jl:38-108): true (V0, κ, θ, σ, ρ) = (0.010201, 6.21, 0.019, 0.61,
-0.7), r = 0.0319, strikes 60:5:140 x expiries 90 / 180 / 365 days, start (0.02, 3, 0.03, 0.4, -0.3),
box bounds.  The reference minimises Σ(price − quote)² with Optimization.jl + ForwardDiff through
solve(::BasketPricingProblem, ::CarrMadan) (calibration.jl:75-88); here every objective evaluation —
all 51 prices AND their 51 x 5 Jacobian — is ONE device launch (hh_carr_madan_basket_grad through Dual
inputs), driven by scipy's bounded least squares.  Needs an MI355X."""
import datetime as dt
import os
import sys
import time

import numpy as np
from scipy.optimize import least_squares

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hedgehog_jl_amd as hh  # noqa: E402
from hedgehog_jl_amd.dual import Dual  # noqa: E402

TRUE = (0.010201, 6.21, 0.019, 0.61, -0.7)
START = (0.02, 3.0, 0.03, 0.4, -0.3)
LOWER, UPPER = (1e-5, 1e-3, 1e-5, 1e-3, -0.99), (1.0, 20.0, 1.0, 5.0, 0.99)


def calibrate(verbose=False):
    ref, r, S0 = hh.Date(2020, 1, 1), 0.0319, 100.0
    payoffs = [hh.VanillaOption(float(K), ref + dt.timedelta(days=d), hh.European(), hh.Call(), hh.Spot())
               for d in (90, 180, 365) for K in np.arange(60.0, 141.0, 5.0)]
    method = hh.CarrMadan(1.0, 32.0, hh.HestonDynamics())
    market = lambda x: hh.HestonInputs(ref, r, S0, *x)
    quotes = np.array([s.price for s in hh.solve(hh.BasketPricingProblem(payoffs, market(TRUE)), method).solutions])
    evals = [0]

    def residuals_and_jacobian(x):
        evals[0] += 1
        seeded = [Dual(v, tuple(1.0 if i == j else 0.0 for i in range(5))) for j, v in enumerate(x)]
        sol = hh.solve(hh.BasketPricingProblem(payoffs, market(seeded)), method)   # one launch
        res = np.array([s.price.value for s in sol.solutions]) - quotes
        jac = np.array([s.price.partials for s in sol.solutions])
        return res, jac

    cache = {}

    def fun(x):
        cache["x"], (cache["r"], cache["J"]) = x.copy(), residuals_and_jacobian(x)
        return cache["r"]

    def jac(x):
        if "x" not in cache or not np.array_equal(cache["x"], x):
            fun(x)
        return cache["J"]

    t0 = time.perf_counter()
    out = least_squares(fun, np.array(START), jac=jac, bounds=(LOWER, UPPER), xtol=1e-12, ftol=1e-14, gtol=1e-12)
    wall = time.perf_counter() - t0
    if verbose:
        print(f"{len(payoffs)} quotes, {evals[0]} objective evaluations (one launch each), {wall * 1e3:.1f} ms")
        for name, got, want in zip(("V0", "kappa", "theta", "sigma", "rho"), out.x, TRUE):
            print(f"  {name:6s} {got: .6f}   true {want: .6f}")
        print(f"  sum of squares {2 * out.cost:.3e}")
    return out.x, 2 * out.cost, evals[0], wall


if __name__ == "__main__":
    calibrate(verbose=True)
