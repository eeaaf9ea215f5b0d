#!/usr/bin/env python3
"""Cost of reducing K payoffs on ONE set of terminal samples (hh_mc_solve_basket): 10^6 exact
lognormal samples (a 0.01 ms simulation, so the call is almost all payoff reduction), K = 64 … 2048
strikes.  GB/s = 8 B x trajectories x K re-read of the samples.  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hedgehog_jl_amd import _ffi

ctx = _ffi.get_context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
seeds = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
m = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0)
c = _ffi.make_config(0, 1, n)
c.seeds, c.seeds_on_device = seeds.data_ptr(), 1
for K in (1, 8, 82, 256, 2048):
    strikes = np.linspace(50.0, 150.0, K)
    cps = np.where(np.arange(K) % 2 == 0, 1.0, -1.0)
    res = (_ffi.hh_result * K)()
    ts = []
    for _ in range(8):
        ctx.check(ctx.lib.hh_mc_solve_basket(ctx.handle, C.byref(m), C.byref(c), strikes.ctypes.data,
                                             cps.ctypes.data, K, res, None))
        ts.append(res[0].kernel_ms)
    t = float(np.median(ts[2:]))
    print(f"{n} samples x {K} payoffs: {t:.3f} ms -> {8.0 * n * K / (t * 1e-3) / 1e9:.0f} GB/s of sample re-reads, "
          f"price[0] {res[0].price:.6f}", flush=True)
