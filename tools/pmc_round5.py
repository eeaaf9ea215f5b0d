#!/usr/bin/env python3
"""The kernels new or changed in round 5, a few launches each, for rocprofv3 PMC passes (SQ_INSTS_VALU,
GRBM_GUI_ACTIVE) and kernel traces:

    rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_round5.py
    python tools/valu_insts_r5.py OUT/*/*_counter_collection.csv r05_x        # merges into profiles/valu_insts.json
"""
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
base = _ffi.make_model()
ctx.check(lib.hh_wiener_fill(h, _ffi.HH_HESTON, base.rho, base.T, M, N, seeds.ptr, 1, dW.ptr))
acc = _ffi.DeviceBuffer(ctx, 8 * _ffi.HH_ACC_LEN * 4)
eps = 1e-3
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def cfg(dyn, strat, n, steps, noise):
    c = _ffi.make_config(dyn, strat, n, steps, noise_mode=noise)
    c.seeds, c.seeds_on_device, c.seeds_len = seeds.ptr, 1, N
    if noise == _ffi.HH_NOISE_REPLAY:
        c.replay, c.replay_on_device = dW.ptr, 1
    return c


two = (_ffi.hh_model * 2)(_ffi.make_model(S0=100 * (1 + eps)), _ffi.make_model(S0=100 * (1 - eps)))
m2 = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0)
jobs = [
    lambda: lib.hh_mc_accumulate(h, C.byref(base), C.byref(cfg(1, 0, N, M, 0)), acc.ptr, None),        # GENERATE
    lambda: lib.hh_mc_accumulate(h, C.byref(base), C.byref(cfg(1, 0, N, M, 1)), acc.ptr, None),        # REPLAY
    lambda: lib.hh_mc_accumulate_multi(h, two, 2, C.byref(cfg(1, 0, N, M, 0)), acc.ptr, None),         # 2 models, GENERATE
    lambda: lib.hh_mc_accumulate_multi(h, two, 2, C.byref(cfg(1, 0, N, M, 1)), acc.ptr, None),         # 2 models, REPLAY
    lambda: lib.hh_mc_accumulate(h, C.byref(m2), C.byref(cfg(0, 1, N, 1, 0)), acc.ptr, None),          # exact law 1e6
    lambda: lib.hh_mc_accumulate(h, C.byref(m2), C.byref(cfg(0, 1, 100 * N, 1, 0)), acc.ptr, None),    # exact law 1e8
    lambda: lib.hh_mc_accumulate(h, C.byref(base), C.byref(cfg(1, 2, N, 1, 0)), acc.ptr, None),        # Broadie–Kaya
]
for job in jobs:
    for _ in range(REPS):
        ctx.check(job())
    ctx.synchronize()
print("done", flush=True)
