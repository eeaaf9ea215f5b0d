"""Longstaff–Schwartz American pricing — numpy restatement of
/root/reference/src/pricing_methods/least_squares_montecarlo.jl:99-165 on top of the oracle's path
grid (hho_gbm_grid in hh_oracle.c).

TEST INFRASTRUCTURE ONLY.  PARITY STATUS: per-draw parity with the reference UNPINNED (its paths
come from DiffEqNoiseProcess' GBM process on Julia RNG streams; its regression from
Polynomials.fit = least squares on the raw Vandermonde matrix, both third party).  The algorithm
below is the reference's, statement by statement; the least-squares fit is computed in the
standardised variable z = (x - mean)/std, which spans the same polynomial space (the fitted
function is the same; only its conditioning is better).  Pinned statistically against the CRR tree
restated in oracle/analytic.py, which reproduces the reference's own CRR regression values
(test/unit/binomial_tree.jl:18,26) — the comparison test/agreement/american_options.jl makes.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def gbm_grid(seeds, n_steps, S0, r, sigma, T, anti):
    lib = C.CDLL(os.path.join(_HERE, "libhh_oracle.so"))
    lib.hho_gbm_grid.restype = None
    lib.hho_gbm_grid.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_double, C.c_double,
                                 C.c_double, C.c_double, C.c_int, C.c_void_p]
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    n = len(seeds)
    out = np.empty((n_steps + 1, n * (2 if anti else 1)))
    lib.hho_gbm_grid(seeds.ctypes.data, n, n_steps, S0, r, sigma, T, int(anti), out.ctypes.data)
    return out


def lsm_solve(spot_grid, strike, cp, step_discount, degree):
    """least_squares_montecarlo.jl:99-136.  spot_grid: (nsteps+1, npaths).
    Returns dict(price, std_error, stop_time, stop_value, steps_regressed)."""
    ntimes, npaths = spot_grid.shape
    nsteps = ntimes - 1
    payoff = lambda s: np.maximum(cp * (s - strike), 0.0)
    tau = np.full(npaths, nsteps, dtype=np.int64)      # stopping_info[p][1]
    val = payoff(spot_grid[nsteps])                    # stopping_info[p][2]
    regressed = 0
    for i in range(nsteps, 1, -1):                     # for i = nsteps:-1:2
        t = i - 1
        continuation = step_discount ** (tau - t) * val
        payoff_t = payoff(spot_grid[t])                # spot_grid[i, :] (1-based row i = time t)
        itm = np.nonzero(payoff_t > 0)[0]
        if itm.size == 0:
            continue
        x = spot_grid[t, itm]
        y = continuation[itm]
        mu, sd = x.mean(), x.std()
        if not sd > 0:
            sd = 1.0
        z = (x - mu) / sd
        V = np.vander(z, degree + 1, increasing=True)
        coef, *_ = np.linalg.lstsq(V, y, rcond=None)   # Polynomials.fit(x, y, degree)
        cont_value = V @ coef
        ex = payoff_t[itm] > cont_value                # update_stopping_info!
        tau[itm[ex]] = t
        val[itm[ex]] = payoff_t[itm[ex]]
        regressed += 1
    disc = step_discount ** tau * val
    return dict(price=float(disc.mean()),
                std_error=float(disc.std(ddof=1) / np.sqrt(npaths)) if npaths > 1 else 0.0,
                stop_time=tau, stop_value=val, steps_regressed=regressed)
