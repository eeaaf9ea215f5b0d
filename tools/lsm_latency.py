#!/usr/bin/env python3
"""LSM solve time against ensemble size (GBM-process paths, 100 exercise dates, degree 5):
event time of the whole launch sequence and wall time of hh_lsm_solve.  GPU box only."""
import ctypes as C
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hedgehog_jl_amd import _ffi

ctx = _ffi.get_context(0)
lib, h = ctx.lib, ctx.handle
steps, degree = 100, 5
for n in (10_000, 50_000, 200_000, 1_000_000):
    m = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0, cp=-1.0)
    c = _ffi.make_config(0, 1, n, steps, antithetic=1, seeds=np.arange(1, n + 1, dtype=np.uint64))
    res = _ffi.hh_lsm_result()
    D = math.exp(-0.05 / steps)
    ks, ws = [], []
    for _ in range(8):
        t0 = time.perf_counter()
        ctx.check(lib.hh_lsm_solve(h, C.byref(m), C.byref(c), degree, D, C.byref(res), None, None, None))
        ws.append((time.perf_counter() - t0) * 1e3)
        ks.append(res.kernel_ms)
    print(f"n={n:8d} x2 antithetic, {steps} dates: kernel sequence {np.median(ks[2:]):.3f} ms, "
          f"wall {np.median(ws[2:]):.3f} ms, price {res.price:.4f}", flush=True)
