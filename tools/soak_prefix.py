"""Soak: a trajectory depends on (key, index) only — the first n1 terminal samples of an n2-trajectory solve
equal the n1-trajectory solve bit for bit, across the Broadie–Kaya slot regimes (static slots up to 2^18
trajectories, XCD-local dynamic slots beyond), the exact law and Euler GENERATE.
GPU box: python tools/soak_prefix.py [seed] [cases]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from hedgehog_jl_amd import _ffi
ctx = _ffi.get_context(0); lib, h = ctx.lib, ctx.handle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
def solve(m, dyn, strat, n, steps, seeds, anti):
    c = _ffi.make_config(dyn, strat, n, steps, antithetic=anti, seeds=seeds)
    res = _ffi.hh_result(); t = np.zeros(n * (2 if anti else 1))
    ctx.check(lib.hh_mc_solve(h, C.byref(m), C.byref(c), C.byref(res), t.ctypes.data))
    return res, t
for it in range(N):
    kind = rng.choice(["bk", "bk", "exact", "euler"])
    n2 = int(rng.choice([rng.integers(2, 5000), rng.integers(250000, 280000), rng.integers(300000, 1200000)]))
    n1 = int(rng.integers(1, n2))
    anti = int(rng.random() < 0.3) if kind != "bk" else 0
    if kind == "bk":
        m = _ffi.make_model(kappa=float(rng.uniform(0.5, 4)), theta=float(rng.uniform(0.02, 0.09)), sigma=float(rng.uniform(0.1, 0.6)),
                            rho=float(rng.uniform(-0.9, 0.2)), V0=float(rng.uniform(0.01, 0.09)), T=float(rng.uniform(0.2, 2)))
        args = (1, 2, 1); seeds = np.array([rng.integers(1, 2**62)], dtype=np.uint64)
    elif kind == "exact":
        m = _ffi.make_model(sigma=0.2); args = (0, 1, 1); seeds = np.array([rng.integers(1, 2**62)], dtype=np.uint64)
    else:
        m = _ffi.make_model(); args = (1, 0, int(rng.integers(1, 12))); seeds = rng.integers(1, 2**62, n2).astype(np.uint64)
        n2 = min(n2, 400000); n1 = min(n1, n2 - 1) or 1; seeds = seeds[:n2]
    r2, t2 = solve(m, args[0], args[1], n2, args[2], seeds, anti)
    r1, t1 = solve(m, args[0], args[1], n1, args[2], seeds if kind != "euler" else seeds[:n1], anti)
    ok = np.array_equal(t1[:n1], t2[:n1]) and (not anti or np.array_equal(t1[n1:], t2[n2:n2 + n1])) and np.all(np.isfinite(t2))
    if not ok:
        bad += 1
        d = np.nonzero(t1[:n1] != t2[:n1])[0]
        print("MISMATCH", kind, dict(n1=n1, n2=n2, anti=anti), "first diffs at", d[:5], flush=True)
print(f"{N} cases, {bad} mismatches")
