#!/usr/bin/env python3
"""REPLAY Heston Euler kernel time against ensemble size (H252, 252 steps): per-launch HIP-event
times of 12 back-to-back launches per size, buffers resident in HBM.  GPU box only."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hedgehog_jl_amd import _ffi

ctx = _ffi.get_context(0)
lib, h = ctx.lib, ctx.handle
m = _ffi.make_model()
steps = 252
out = {}
for n in [int(x) for x in (sys.argv[1:] or ["100000", "1000000", "4000000", "10000000"])]:
    seeds = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
    rep = torch.empty(lib.hh_replay_elems(n, steps, 1), dtype=torch.float64, device="cuda")
    ctx.check(lib.hh_wiener_fill(h, 1, m.rho, m.T, steps, n, seeds.data_ptr(), 1, rep.data_ptr()))
    acc = torch.zeros(16, dtype=torch.float64, device="cuda")
    c = _ffi.make_config(1, 0, n, steps, noise_mode=1)
    c.replay, c.replay_on_device, c.replay_len = rep.data_ptr(), 1, rep.numel()
    ctx.enable_timing(True)
    for _ in range(int(os.environ.get('HH_SIZES_LAUNCHES', '12'))):
        ctx.check(lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc.data_ptr(), None))
    ctx.synchronize()
    t = np.array(ctx.read_timings())
    t = t[len(t) // 4:]  # past the first launches
    ctx.enable_timing(False)
    gbs = 16e-9 * n * steps / (t * 1e-3)
    out[n] = dict(ms=[round(float(x), 4) for x in t], GBs_median=round(float(np.median(gbs)), 1),
                  GBs_best=round(float(gbs.max()), 1))
    print(n, out[n], flush=True)
    del rep, seeds
    torch.cuda.empty_cache()
print(json.dumps(out))
