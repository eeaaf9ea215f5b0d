#!/usr/bin/env python3
"""Where the tail of a self-reducing launch goes (a -DHH_FINISH_STAMPS=1 build: $HEDGEHOG_MC_LIB): ticks of
the 100 MHz clock at which the reducing workgroup had its own record out, had read all records, had its sums —
against a stamp a following one-thread kernel... (the kernel's end is not visible from inside: the event time
of the launch minus the stamps' span bounds it)."""
import ctypes as C
import os
import sys

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
m2 = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0)


def cfg(dyn, strat, n, steps, noise):
    c = _ffi.make_config(dyn, strat, n, steps, noise_mode=noise)
    c.seeds, c.seeds_on_device, c.seeds_len = seeds.ptr, 1, N
    if noise == _ffi.HH_NOISE_REPLAY:
        c.replay, c.replay_on_device = dW.ptr, 1
    return c


for name, mdl, c in (("heston_euler_replay_1e6x252", model, cfg(1, 0, N, M, 1)),
                     ("heston_euler_generate_1e6x252", model, cfg(1, 0, N, M, 0)),
                     ("lognormal_exact_1e6", m2, cfg(0, 1, N, 1, 0)),
                     ("lognormal_euler_1e4x100", m2, cfg(0, 0, 10_000, 100, 0))):
    rows = []
    for _ in range(30):
        ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(c), acc.ptr, None))
    ctx.enable_timing(True)
    for _ in range(10):
        ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(c), acc.ptr, None))
        ctx.synchronize()
        a = acc.download(np.empty(_ffi.HH_ACC_LEN))
        rows.append((a[12] - a[11], a[13] - a[12]))
    ev = ctx.read_timings()
    ctx.enable_timing(False)
    r = np.array(rows) / 100.0
    print(f"{name}: launch {np.median(ev) * 1e3:.1f} us; own record out -> all records read "
          f"{np.median(r[:, 0]):.2f} us (min {r[:, 0].min():.2f}, max {r[:, 0].max():.2f}); -> sums done {np.median(r[:, 1]):.2f} us")
