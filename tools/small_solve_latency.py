#!/usr/bin/env python3
"""Wall-clock latency of small synchronous solves through the C-ABI (calibration-style loops)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hedgehog_jl_amd as hh  # noqa: E402
from hedgehog_jl_amd import _ffi  # noqa: E402

ctx = hh.Context(0)
lib, h = ctx.lib, ctx.handle
for label, n, steps, dyn in (("BS Euler 1e4x100", 10_000, 100, 0), ("Heston Euler 1e4x100", 10_000, 100, 1),
                             ("Heston Euler 1e5x252", 100_000, 252, 1)):
    m = _ffi.make_model(sigma=0.2 if dyn == 0 else 0.3)
    seeds_h = np.arange(1, n + 1, dtype=np.uint64)
    seeds_d = torch.from_numpy(seeds_h.view(np.int64)).cuda()
    for where in ("host seeds", "device seeds"):
        c = _ffi.make_config(dyn, 0, n, steps, seeds=seeds_h)
        if where == "device seeds":
            c.seeds, c.seeds_on_device = seeds_d.data_ptr(), 1
        r = _ffi.hh_result()
        for _ in range(20):
            lib.hh_mc_solve(h, C.byref(m), C.byref(c), C.byref(r), None)
        t0 = time.perf_counter()
        reps = 300
        for _ in range(reps):
            lib.hh_mc_solve(h, C.byref(m), C.byref(c), C.byref(r), None)
        dt = (time.perf_counter() - t0) / reps
        print(f"{label:22s} {where:13s}: {dt * 1e6:7.1f} us per solve (kernel+reduce events {r.kernel_ms * 1e3:6.1f} us)")
