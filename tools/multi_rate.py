#!/usr/bin/env python3
"""hh_mc_accumulate_multi at the headline size (10^6 x 252, Heston Euler): ms per pass of K models on the same
draws against K one-model passes, GENERATE and REPLAY.  $HEDGEHOG_MC_LIB selects an A/B build."""
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
base = _ffi.make_model()
ctx.check(lib.hh_wiener_fill(h, _ffi.HH_HESTON, base.rho, base.T, M, N, seeds.ptr, 1, dW.ptr))
acc = _ffi.DeviceBuffer(ctx, 8 * _ffi.HH_ACC_LEN * 16)
eps = 1e-3
bumps = [dict(S0=100 * (1 + eps)), dict(S0=100 * (1 - eps)), dict(), dict(V0=0.04 * (1 + eps))]


def cfg(noise, anti=0):
    c = _ffi.make_config(1, 0, N, M, noise_mode=noise, antithetic=anti)
    c.seeds, c.seeds_on_device, c.seeds_len = seeds.ptr, 1, N
    if noise == _ffi.HH_NOISE_REPLAY:
        c.replay, c.replay_on_device = dW.ptr, 1
    return c


def timed(call, reps):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.03:
        for _ in range(4):
            call()
        ctx.synchronize()
    ctx.enable_timing(True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ev = ctx.read_timings()
    ctx.enable_timing(False)
    return wall, float(np.mean(ev))


print("lib:", os.environ.get("HEDGEHOG_MC_LIB", "(default)"), flush=True)
for noise, name in ((_ffi.HH_NOISE_GENERATE, "generate"), (_ffi.HH_NOISE_REPLAY, "replay")):
    for anti in (0, 1):
        c = cfg(noise, anti)
        one = _ffi.make_model(**bumps[0])
        w1, e1 = timed(lambda: ctx.check(lib.hh_mc_accumulate(h, C.byref(one), C.byref(c), acc.ptr, None)), 40)
        row = {"one_model_ms": round(e1, 4), "one_model_wall_ms": round(w1, 4)}
        for K in (2, 3, 4):
            arr = (_ffi.hh_model * K)(*[_ffi.make_model(**bumps[k]) for k in range(K)])
            w, e = timed(lambda: ctx.check(lib.hh_mc_accumulate_multi(h, arr, K, C.byref(c), acc.ptr, None)), 30)
            row[f"K{K}_ms"] = round(e, 4)
            row[f"K{K}_wall_ms"] = round(w, 4)
            row[f"K{K}_vs_K_solves"] = round(e / (K * e1), 3)
        print(name, "anti" if anti else "plain", json.dumps(row), flush=True)
ctx.synchronize()
