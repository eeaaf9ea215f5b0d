#!/usr/bin/env python3
"""LSM solve time against ensemble size (GBM-process paths, 100 exercise dates, degree 5) in both
forms of the backward induction — ONE persistent launch vs one launch per exercise date
(HH_OPT_LSM_FORM): event time of everything the call enqueued and wall time of hh_lsm_solve, plus
a check that the two forms give the same price bit for bit.  GPU box only.

Environment (the exit-time diagnosis of profiles/README.md): HH_LSM_FORMS=persistent|per-date (default
both), HH_CLOSE=1 destroys the context before the interpreter exits, HH_DEVICE_RESET=1 also calls
hipDeviceReset() then (the runtime gives back its queues — the cooperative one included — while the
profiler is still alive), HEDGEHOG_MC_NO_TORCH=1 keeps PyTorch's bundled HIP runtime out of the process,
HH_DUMP_MAPS=<file> writes /proc/self/maps there as the last statement."""
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
sizes = [int(x) for x in sys.argv[1:]] or [10_000, 50_000, 131_072, 200_000, 500_000, 1_000_000]
FORMS = {"persistent": ((_ffi.HH_LSM_FORM_PERSISTENT, "one launch"),), "per-date": ((_ffi.HH_LSM_FORM_PER_DATE, "per date"),)}
forms = FORMS.get(os.environ.get("HH_LSM_FORMS", ""),
                  ((_ffi.HH_LSM_FORM_PERSISTENT, "one launch"), (_ffi.HH_LSM_FORM_PER_DATE, "per date")))
for n in sizes:
    m = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0, cp=-1.0)
    c = _ffi.make_config(0, 1, n, steps, antithetic=1, seeds=np.arange(1, n + 1, dtype=np.uint64))
    D = math.exp(-0.05 / steps)
    line, prices = [], []
    for form, name in forms:
        ctx.set_option(_ffi.HH_OPT_LSM_FORM, form)
        res = _ffi.hh_lsm_result()
        ks, ws = [], []
        for _ in range(8):
            t0 = time.perf_counter()
            ctx.check(lib.hh_lsm_solve(h, C.byref(m), C.byref(c), degree, D, C.byref(res), None, None, None))
            ws.append((time.perf_counter() - t0) * 1e3)
            ks.append(res.kernel_ms)
        line.append(f"{name} (ran as form {res.form}): {np.median(ks[2:]):.3f} ms events, "
                    f"{np.median(ws[2:]):.3f} ms wall")
        prices.append(res.price)
    ctx.set_option(_ffi.HH_OPT_LSM_FORM, _ffi.HH_LSM_FORM_AUTO)
    print(f"n={n:8d} x2 antithetic, {steps} dates: " + " | ".join(line) +
          f" | price {prices[0]:.6f} identical={prices[0] == prices[-1]}", flush=True)
if os.environ.get("HH_CLOSE") == "1":
    ctx.close()
    _ffi._contexts.clear()
if os.environ.get("HH_DEVICE_RESET") == "1":
    hip = C.CDLL(None)  # the HIP runtime the library is linked against (global scope)
    print("hipDeviceReset ->", hip.hipDeviceReset(), flush=True)
if os.environ.get("HH_DUMP_MAPS"):
    open(os.environ["HH_DUMP_MAPS"], "w").write(open("/proc/self/maps").read())
