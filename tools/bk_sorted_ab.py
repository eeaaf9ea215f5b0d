#!/usr/bin/env python3
"""Upper bound of what ordering the trajectories buys the Broadie–Kaya chain: the same draws [V_T | u | Z]
(REPLAY, device-resident) in random order, sorted by V_T, and sorted by (coarse V_T, u); kernel time of the
chain.  GPU box only."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy import stats

from hedgehog_jl_amd import _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0  # 1/12: a date of the 12-date exact grid (short maturity: long series)
H = dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=T, strike=100.0, cp=1.0)
s2 = H["sigma"] ** 2
em = -math.expm1(-H["kappa"] * H["T"])
d = 4 * H["kappa"] * H["theta"] / s2
lam = 4 * H["kappa"] * math.exp(-H["kappa"] * H["T"]) * H["V0"] / (s2 * em)
rng = np.random.default_rng(5)
VT = s2 * em / (4 * H["kappa"]) * stats.ncx2.rvs(d, lam, size=n, random_state=rng)
u = rng.uniform(1e-6, 1 - 1e-6, n)
Z = rng.standard_normal(n)
ctx = _ffi.get_context(0)
m = _ffi.make_model(**H)


def run(order, label):
    draws = torch.from_numpy(np.concatenate([VT[order], u[order], Z[order]])).cuda()
    c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, 1, noise_mode=_ffi.HH_NOISE_REPLAY)
    c.replay, c.replay_on_device, c.replay_len = draws.data_ptr(), 1, draws.numel()
    res = _ffi.hh_result()
    t = []
    for _ in range(12):
        ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(res), None))
        t.append(res.kernel_ms)
    print(f"{label:36s} chain {np.median(t[3:]):.4f} ms (min {min(t[3:]):.4f})  price {res.price:.6f}  "
          f"cf_terms/path {res.bk_cf_terms / n:.2f} newton_fail {res.bk_newton_fail}", flush=True)


idx = np.arange(n)
run(idx, "random order")
run(np.argsort(VT), "sorted by V_T")
key = np.floor(np.log2(VT) * 8).astype(np.int64)
run(np.lexsort((u, key)), "sorted by (log2 V_T in 1/8 bins, u)")
run(np.argsort(key, kind="stable"), "stable sort by log-scale bin")
run(np.argsort(u), "sorted by u")
# what a sort INSIDE each tile of 256 trajectories (no pass over the ensemble) would buy: by V_T, and by a 16-bin key of it
tiles = idx // 256
run(np.lexsort((VT, tiles)), "each tile of 256 sorted by V_T")
edges = np.quantile(VT, np.linspace(0, 1, 17)[1:-1])
run(np.lexsort((idx, np.searchsorted(edges, VT), tiles)), "each tile by a 16-quantile bin of V_T")
run(np.lexsort((idx, np.clip(key - key.min(), 0, None) // 2, tiles)), "each tile by log2 V_T in 1/4 bins")
run(idx, "random order again")
