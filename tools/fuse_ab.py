#!/usr/bin/env python3
"""A/B of HH_OPT_FUSE_REDUCE: the record reduction inside the simulation kernel (1) against the separate
reduce_records_kernel (0), same build, same process.  Per row: wall time per step of a back-to-back loop
(what a pricing service sees) and the HIP-event time of what one hh_mc_accumulate enqueues."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hedgehog_jl_amd as hh  # noqa: E402
from hedgehog_jl_amd import _ffi  # noqa: E402

ctx = hh.get_context(0)
lib, h = ctx.lib, ctx.handle
N, M = 1_000_000, 252
seeds = _ffi.DeviceBuffer(ctx, 8 * N).upload(np.arange(1, N + 1, dtype=np.uint64))
dW = _ffi.DeviceBuffer(ctx, 8 * lib.hh_replay_elems(N, M, _ffi.HH_HESTON))
model = _ffi.make_model()
ctx.check(lib.hh_wiener_fill(h, _ffi.HH_HESTON, model.rho, model.T, M, N, seeds.ptr, 1, dW.ptr))
acc = _ffi.DeviceBuffer(ctx, 8 * _ffi.HH_ACC_LEN)


def cfg(dyn, strat, n, steps, noise, anti=0):
    c = _ffi.make_config(dyn, strat, n, steps, noise_mode=noise, antithetic=anti)
    c.seeds, c.seeds_on_device, c.seeds_len = seeds.ptr, 1, N
    if noise == _ffi.HH_NOISE_REPLAY:
        c.replay, c.replay_on_device = dW.ptr, 1
    return c


m2 = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0)
rows = [
    ("heston_euler_replay_1e6x252", model, cfg(1, 0, N, M, 1), 200),
    ("heston_euler_replay_anti_1e6x252", model, cfg(1, 0, N, M, 1, 1), 100),
    ("heston_euler_generate_1e6x252", model, cfg(1, 0, N, M, 0), 60),
    ("lognormal_exact_1e6", m2, cfg(0, 1, N, 1, 0), 400),
    ("lognormal_exact_1e7", m2, cfg(0, 1, 10 * N, 1, 0), 100),
    ("lognormal_exact_1e8", m2, cfg(0, 1, 100 * N, 1, 0), 20),
    ("lognormal_euler_1e4x100 (config 1)", m2, cfg(0, 0, 10_000, 100, 0), 400),
    ("heston_euler_generate_1e4x100", model, cfg(1, 0, 10_000, 100, 0), 400),
]


def run(mdl, c, k, fused):
    ctx.set_option(_ffi.HH_OPT_FUSE_REDUCE, fused)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.03:  # the clock this kernel holds
        for _ in range(8):
            ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(c), acc.ptr, None))
        ctx.synchronize()
    ctx.enable_timing(True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(c), acc.ptr, None))
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / k * 1e3
    ev = ctx.read_timings()
    ctx.enable_timing(False)
    return wall, float(np.mean(ev)), acc.download(np.empty(_ffi.HH_ACC_LEN))


out = {}
print("HH_GRID_WG_PER_CU =", os.environ.get("HH_GRID_WG_PER_CU", "(auto)"), flush=True)
for name, mdl, c, k in rows:
    r = {}
    for rep in range(3):
        for fused in (0, 1):
            wall, ev, a = run(mdl, c, k, fused)
            e = r.setdefault("fused" if fused else "separate", {"wall_ms": [], "event_ms": []})
            e["wall_ms"].append(round(wall, 5))
            e["event_ms"].append(round(ev, 5))
            e["sum"] = float(a[0])
    r["same_bits"] = bool(r["fused"]["sum"] == r["separate"]["sum"])
    r["wall_ratio_fused_over_separate"] = min(r["fused"]["wall_ms"]) / min(r["separate"]["wall_ms"])
    out[name] = r
    print(name, json.dumps(r), flush=True)
ctx.set_option(_ffi.HH_OPT_FUSE_REDUCE, 2)
ctx.synchronize()
