#!/usr/bin/env python3
"""Kernel time of the fused dual-number Greeks against the number of ACTIVE directions (those that
reach the variance / diffusion: V0, κ, θ, σ) on H252, 10^6 x 252, REPLAY and GENERATE.  Passive
directions (S0, r, strike) cost nothing per path-step.  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hedgehog_jl_amd import _ffi

ctx = _ffi.get_context(0)
lib, h = ctx.lib, ctx.handle
N, M = 1_000_000, 252
seeds = torch.arange(1, N + 1, dtype=torch.int64, device="cuda")
m0 = _ffi.make_model()
rep = torch.empty(lib.hh_replay_elems(N, M, 1), dtype=torch.float64, device="cuda")
ctx.check(lib.hh_wiener_fill(h, 1, m0.rho, m0.T, M, N, seeds.data_ptr(), 1, rep.data_ptr()))
acc = torch.zeros(16, dtype=torch.float64, device="cuda")
names = ["V0", "kappa", "theta", "sigma"]
for n_act in (0, 1, 2, 3, 4):
    for extra_passive in ((0,) if n_act == 0 else (0, 2)):
        P = n_act + extra_passive
        sd = {}
        for k in range(n_act):
            sd[names[k]] = [1.0 if j == k else 0.0 for j in range(P)]
        if extra_passive:
            sd["S0"] = [1.0 if j == n_act else 0.0 for j in range(P)]
            sd["r_drift"] = [1.0 if j == n_act + 1 else 0.0 for j in range(P)]
        m = _ffi.make_model(seeds=sd, n_partials=P) if P else _ffi.make_model()
        row = [f"active={n_act} passive={extra_passive} P={P}"]
        for noise in (1, 0):
            c = _ffi.make_config(1, 0, N, M, noise_mode=noise, n_partials=P)
            c.seeds, c.seeds_on_device = seeds.data_ptr(), 1
            c.replay, c.replay_on_device, c.replay_len = rep.data_ptr(), 1, rep.numel()
            ctx.enable_timing(True)
            for _ in range(12):
                ctx.check(lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc.data_ptr(), None))
            ctx.synchronize()
            t = np.median(ctx.read_timings()[2:])
            ctx.enable_timing(False)
            row.append(f"{'REPLAY' if noise else 'GENERATE'} {t:.3f} ms")
        print("  ".join(row), flush=True)
