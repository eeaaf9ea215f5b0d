#!/usr/bin/env python3
"""Exact Heston grid (hh_heston_exact_grid): dates batched into one kernel chain vs one chain per date
(HH_OPT_GRID_FORM), and the batched chain with its pairs sorted by the size of their Bessel argument (HH_OPT_GRID_ORDER);
kernel time by HIP events.  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hedgehog_jl_amd as hh
from hedgehog_jl_amd import _ffi

H252 = dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1.0)
for n, steps in ((20_000, 12), (200_000, 12), (200_000, 50), (1_000_000, 12), (50_000, 252)):
    seeds = torch.arange(1, n + 1, dtype=torch.int64, device="cuda:0")
    line = f"n={n:8d} x {steps:3d} dates:"
    for form, order, name in ((_ffi.HH_GRID_FORM_PER_DATE, 0, "chain per date"), (_ffi.HH_GRID_FORM_BATCHED, 0, "dates batched"),
                              (_ffi.HH_GRID_FORM_BATCHED, 1, "batched, pairs sorted by V0*VT")):
        ctx = hh.Context(0)
        ctx.check(ctx.lib.hh_ctx_set_option(ctx.handle, _ffi.HH_OPT_GRID_FORM, form))
        ctx.check(ctx.lib.hh_ctx_set_option(ctx.handle, _ffi.HH_OPT_GRID_ORDER, order))
        m = _ffi.make_model(**H252, strike=100.0)
        c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, steps)
        c.seeds, c.seeds_on_device, c.seeds_len = seeds.data_ptr(), 1, n
        r = _ffi.hh_result()
        t = []
        for _ in range(4):
            ctx.check(ctx.lib.hh_heston_exact_grid(ctx.handle, C.byref(m), C.byref(c), None, None, 0, C.byref(r)))
            t.append(r.kernel_ms)
        line += f"  {name} {min(t[1:]):8.3f} ms ({n * steps / min(t[1:]) * 1e-3:6.1f} M transitions/s, {r.bk_cf_terms / (n * steps):5.1f} terms)"
        ctx.close()
    print(line, flush=True)
