"""Soak: the dual partials of a solve against central finite differences of the same solve on the same
seeds (common random numbers: the Monte Carlo price is piecewise smooth in the parameters), for random
Heston / lognormal models, Euler and the exact law, antithetic or not — every parameter the ABI can seed.
A review aid, not a pass/fail test: what it prints are the cases where the two differ by more than 2e-4 (3 %
for the variance parameters) — seen so far only where few trajectories end in the money (finite-difference
noise of the kink at the strike) or the Euler variance sits on its clip.
GPU box: python tools/soak_duals.py [seed] [cases]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from hedgehog_jl_amd import _ffi
ctx = _ffi.get_context(0); lib, h = ctx.lib, ctx.handle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
def price(kw, dyn, strat, n, steps, seeds, anti, sd=None, P=0):
    m = _ffi.make_model(**kw, seeds=sd, n_partials=P)
    c = _ffi.make_config(dyn, strat, n, steps, antithetic=anti, seeds=seeds, n_partials=P)
    r = _ffi.hh_result()
    ctx.check(lib.hh_mc_solve(h, C.byref(m), C.byref(c), C.byref(r), None))
    return r
bad = 0
for it in range(N):
    kind = rng.choice(["heston", "heston", "gbm_euler", "gbm_exact"])
    n = int(rng.integers(2000, 60000)); steps = int(rng.integers(2, 60)); anti = int(rng.random() < 0.4)
    kw = dict(S0=float(rng.uniform(50, 150)), r=float(rng.uniform(0.0, 0.08)), T=float(rng.uniform(0.2, 2.0)),
              strike=float(rng.uniform(70, 130)), cp=float(rng.choice([1.0, -1.0])))
    if kind == "heston":
        kw.update(V0=float(rng.uniform(0.02, 0.09)), kappa=float(rng.uniform(0.5, 4)), theta=float(rng.uniform(0.02, 0.09)),
                  sigma=float(rng.uniform(0.1, 0.5)), rho=float(rng.uniform(-0.9, 0.3)))
        if os.environ.get("HH_SOAK_FELLER"):  # keep the variance off its clip: 2 kappa theta >= 2 sigma^2
            kw["sigma"] = min(kw["sigma"], math.sqrt(kw["kappa"] * kw["theta"]))
        dyn, strat = 1, 0; names = ["S0", "V0", "kappa", "theta", "sigma", "strike"]
        seeds = rng.integers(1, 2**62, n).astype(np.uint64)
    else:
        kw.update(sigma=float(rng.uniform(0.1, 0.5)))
        dyn = 0; strat = 0 if kind == "gbm_euler" else 1; names = ["S0", "sigma", "strike"]
        seeds = rng.integers(1, 2**62, n if strat == 0 else 1).astype(np.uint64)
        if strat == 1: steps = 1
    kw["discount"] = math.exp(-kw["r"] * kw["T"])
    P = len(names)
    sd = {nm: [1.0 if j == i else 0.0 for j in range(P)] for i, nm in enumerate(names)}
    ad = price(kw, dyn, strat, n, steps, seeds, anti, sd, P)
    for i, nm in enumerate(names):
        x = kw[nm]; e = 1e-5 * max(abs(x), 1e-2)
        up = price({**kw, nm: x + e}, dyn, strat, n, steps, seeds, anti).price
        dn = price({**kw, nm: x - e}, dyn, strat, n, steps, seeds, anti).price
        fd = (up - dn) / (2 * e)
        # the variance parameters act through sqrt(max(v, 0)), which is not Lipschitz at the clip: there the
        # pathwise derivative (what ForwardDiff computes in the reference too) and a finite difference agree
        # only to ~sqrt(eps) of the clipped paths' share — 3 % asked when 2 kappa theta >= 2 sigma^2, nothing below
        loose = kind == "heston" and nm in ("V0", "kappa", "theta", "sigma")
        fel = 2 * kw.get("kappa", 0) * kw.get("theta", 0) / kw["sigma"] ** 2 if kind == "heston" else float("inf")
        if loose and fel < 2.0:
            continue
        tol = (3e-2 if loose else 2e-4) * max(abs(fd), abs(ad.dprice[i]), 1e-3 * ad.price / max(abs(x), 1e-2) + 1e-6)
        if not abs(fd - ad.dprice[i]) <= tol:
            bad += 1
            print("MISMATCH", kind, nm, f"feller {fel:.2f}", dict(n=n, steps=steps, anti=anti), f"AD {ad.dprice[i]:.8g} FD {fd:.8g}", {k: round(v, 4) for k, v in kw.items()}, flush=True)
print(f"{N} cases, {bad} mismatches")
