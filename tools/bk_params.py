#!/usr/bin/env python3
"""Broadie–Kaya chain time by parameter regime (10^6 trajectories each): the Bessel order ν = 2κθ/σ² − 1 and
the size of the Bessel arguments decide which expansion the CF kernel runs.  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hedgehog_jl_amd import _ffi

PARAMS = {
    "h252 (nu 0.78)": dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1.0),
    "intended (nu 0.33)": dict(S0=100.0, V0=0.04, kappa=1.5, theta=0.04, sigma=0.3, rho=-0.6, r=0.05, T=364 / 365),
    "q2 (nu -0.93)": dict(S0=100.0, V0=1.5, kappa=0.04, theta=0.3, sigma=-0.6, rho=0.04, r=0.05, T=364 / 365),
    "nu_one": dict(S0=100.0, V0=0.06, kappa=1.0, theta=0.09, sigma=0.3, rho=-0.4, r=0.02, T=1.5),
    "feller, nu 3": dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.2, rho=-0.7, r=0.03, T=1.0),
    "large_nu (nu 15)": dict(S0=100.0, V0=0.05, kappa=2.0, theta=0.04, sigma=0.1, rho=-0.3, r=0.02, T=0.5),
    "nu 63": dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.05, rho=-0.3, r=0.02, T=1.0),
    "nu 475 (besseli underflows)": dict(S0=100.0, V0=0.0242, kappa=3.719, theta=0.16, sigma=0.05, rho=0.4, r=0.02, T=2.154),
    "absorbed (nu -0.994)": dict(S0=100.0, V0=0.01, kappa=0.3, theta=0.01, sigma=1.0, rho=-0.9, r=0.02, T=1.9),
    "short_T (Hankel)": dict(S0=100.0, V0=0.09, kappa=1.0, theta=0.02, sigma=0.5, rho=-0.5, r=0.01, T=0.02),
    "monthly step of h252": dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1 / 12),
}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
only = sys.argv[2] if len(sys.argv) > 2 else ""  # substring of the regime's name
ctx = _ffi.get_context(0)
for name, prm in PARAMS.items():
    if only not in name:
        continue
    m = _ffi.make_model(strike=100.0, cp=1.0, **prm)
    c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, seeds=np.array([7], dtype=np.uint64))
    r = _ffi.hh_result()
    ts = []
    for _ in range(4):
        ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(r), None))
        ts.append(r.kernel_ms)
    print(f"{name:24s} {min(ts[1:]):8.3f} ms  price {r.price:10.6f}  cf terms/path {r.bk_cf_terms / n:7.2f}  "
          f"ladder {r.bk_newton_fail:7d}  max_guess {r.bk_maxguess_fallback}", flush=True)
