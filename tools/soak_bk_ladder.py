#!/usr/bin/env python3
"""Soak of the in-kernel bisection ladder of the Broadie–Kaya CF kernel (hh_bk.hip, wave_ladder: the failed
trajectories of a tile walked as bisection TREES by one wave) against the lane-by-lane loop of the check build
(tests/c/libhh_bk_check.so: -DHH_BK_SERIAL_LADDER=1, statement for statement sample_from_cf.jl:123-133): random models,
ensemble sizes and controls — among them the ones that push many trajectories into the ladder, end it by its cap in
the middle of a tree, or stretch it over many turns — every sample, decision word and counter bit for bit.
usage: soak_bk_ladder.py [seed] [cases].  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o
from tests.c.build_bk_check import build_bk_check

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
libs = []
for path in (_ffi.LIB_PATH, build_bk_check()):
    lib = C.CDLL(path)
    for name, res, args in _ffi.SYMBOLS:
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    h = C.c_void_p()
    assert lib.hh_ctx_create(C.byref(h), 0) == 0
    libs.append((lib, h))
FIELDS = ("price", "std_error", "sum_payoff", "sumsq_payoff", "bk_cf_terms", "bk_newton_fail", "bk_bisect_fallback",
          "bk_maxguess_fallback")
bad = ladders = skipped = 0
for case in range(cases):
    kappa, theta, sigma = rng.uniform(0.3, 4.0), rng.uniform(0.02, 0.12), rng.uniform(0.1, 0.8)
    prm = dict(S0=rng.uniform(50, 150), V0=rng.uniform(0.01, 0.12), kappa=kappa, theta=theta, sigma=sigma,
               rho=rng.uniform(-0.9, 0.5), r=rng.uniform(0.0, 0.06), T=rng.uniform(0.05, 2.0),
               strike=rng.uniform(60, 140), cp=float(rng.choice([-1.0, 1.0])))
    n = int(rng.integers(200, 40_000))
    m = o.make_model(**prm)
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, 1, seeds=[int(rng.integers(1, 2**62))],
                      path_offset=int(rng.integers(0, 1000)))
    kind = rng.integers(0, 6)
    if kind == 1:
        c.bk_newton_maxiter = int(rng.integers(2, 5))                 # most trajectories in the ladder
    elif kind == 2:
        c.bk_newton_maxiter, c.bk_bisect_maxiter = 2, int(rng.integers(1, 12))  # the cap ends it inside a tree
    elif kind == 3:
        c.bk_newton_maxiter, c.bk_atol = int(rng.integers(2, 6)), float(10.0 ** rng.uniform(-13, -6))  # many turns
    elif kind == 4:
        c.bk_newton_maxiter, c.bk_atol = 2, float(rng.uniform(0.01, 0.6))          # ends at its first levels
    out = []
    for lib, h in libs:
        r = _ffi.hh_result()
        t = np.zeros(n)
        rc = lib.hh_mc_solve(h, C.byref(m), C.byref(c), C.byref(r), t.ctypes.data)
        dec, ln = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        if rc == 0:
            assert lib.hh_bk_decisions(h, n, dec.ctypes.data, ln.ctypes.data) == 0
        out.append((rc, r, t, dec, ln))
    (rc1, r1, t1, d1, l1), (rc0, r0, t0, d0, l0) = out
    if rc1 != rc0:
        bad += 1
        print("STATUS DIFFERS", case, rc1, rc0, prm, flush=True)
        continue
    if rc1 != 0:  # a law outside the supported range: an argument error in both
        skipped += 1
        continue
    ladders += int(r1.bk_newton_fail)
    same = all(np.float64(getattr(r1, f)).tobytes() == np.float64(getattr(r0, f)).tobytes() for f in FIELDS)
    if not (same and t1.tobytes() == t0.tobytes() and d1.tobytes() == d0.tobytes() and l1.tobytes() == l0.tobytes()):
        bad += 1
        print("MISMATCH", case, "kind", int(kind), prm, n, "samples", int((t1 != t0).sum()), "words", int((d1 != d0).sum()),
              flush=True)
print(f"{cases} cases ({skipped} refused by both), {ladders} trajectories through the ladder, {bad} mismatches")
