#!/usr/bin/env python3
"""Soak of hh_mc_solve_multi on HestonBroadieKaya: random models, ensembles and sets of bumps (some invisible to the
variance process — they share the first model's chain —, some not); every model's result and samples against its own
hh_mc_solve, bit for bit.  usage: soak_bk_multi.py [seed] [cases].  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = _ffi.Context(0)
FIELDS = ("price", "std_error", "sum_payoff", "sumsq_payoff", "n_paths_done", "bk_cf_terms", "bk_newton_fail",
          "bk_bisect_fallback", "bk_maxguess_fallback")
bad = 0
for case in range(cases):
    kappa, theta = rng.uniform(0.5, 4.0), rng.uniform(0.02, 0.12)
    sigma = rng.uniform(0.15, 0.6)
    base = dict(S0=rng.uniform(50, 150), V0=rng.uniform(0.01, 0.12), kappa=kappa, theta=theta, sigma=sigma,
                rho=rng.uniform(-0.9, 0.5), r=rng.uniform(0.0, 0.06), T=rng.uniform(0.1, 2.0),
                strike=rng.uniform(60, 140), cp=float(rng.choice([-1.0, 1.0])))
    K = int(rng.integers(2, 6))
    models = [o.make_model(**base)]
    for _ in range(K - 1):
        b = dict(base)
        for key in rng.choice(["S0", "r", "rho", "strike", "cp", "V0", "sigma", "T"], size=int(rng.integers(1, 3)), replace=False,
                              p=[0.25, 0.15, 0.15, 0.15, 0.1, 0.08, 0.06, 0.06]):
            b[key] = -b[key] if key == "cp" else b[key] * (1 + rng.choice([-1, 1]) * rng.uniform(1e-4, 5e-2))
        b["rho"] = float(np.clip(b["rho"], -0.95, 0.95))
        models.append(o.make_model(**b))
    n = int(rng.integers(300, 70_000))
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, 1, seeds=[int(rng.integers(1, 2**62))])
    if rng.uniform() < 0.2:
        c.bk_newton_maxiter = 2
    each = []
    for m in models:
        r = _ffi.hh_result()
        t = np.zeros(n)
        ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(r), t.ctypes.data))
        each.append((r, t))
    res = (_ffi.hh_result * K)()
    terms = [np.zeros(n) for _ in range(K)]
    tp = (C.c_void_p * K)(*[t.ctypes.data for t in terms])
    ctx.check(ctx.lib.hh_mc_solve_multi(ctx.handle, (_ffi.hh_model * K)(*models), K, C.byref(c), res, tp))
    for k in range(K):
        same = all(np.float64(getattr(each[k][0], f)).tobytes() == np.float64(getattr(res[k], f)).tobytes() for f in FIELDS)
        if not same or each[k][1].tobytes() != terms[k].tobytes():
            bad += 1
            print("MISMATCH", case, k, base, n, flush=True)
print(f"{cases} cases, {bad} mismatches")
