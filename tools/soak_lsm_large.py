"""Soak: the LSM induction against the numpy oracle at ensemble sizes the randomised test does not reach
(every chunk-size tier, ragged last chunks), oracle run on the GPU's own grid (the grid itself is checked
in tests/).  GPU box: python tools/soak_lsm_large.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from hedgehog_jl_amd import _ffi
from oracle import lsm_oracle
from tests.test_gpu_lsm import gpu_lsm
ctx = _ffi.get_context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(N):
    n = int(rng.choice([rng.integers(5000, 40000), rng.integers(100000, 140000), rng.integers(250000, 300000), rng.integers(500000, 600000)]))
    anti = int(rng.random() < 0.5)
    if n * (1 + anti) > 700000: anti = 0
    steps = int(rng.integers(2, 16)); degree = int(rng.integers(1, 7)); cp = float(rng.choice([1.0, -1.0]))
    S0 = float(rng.uniform(20, 200)); K = S0 * float(rng.uniform(0.8, 1.25)); r = float(rng.uniform(0.005, 0.12))
    sigma = float(rng.uniform(0.08, 0.6)); T = float(rng.uniform(0.1, 2.5))
    seeds = rng.integers(0, 2**63, n).astype(np.uint64)
    res, tau, val, grid, D = gpu_lsm(ctx, S0, K, r, sigma, T, cp, seeds, steps, anti, degree)
    ref = lsm_oracle.lsm_solve(grid, K, cp, D, degree)
    same = tau == ref["stop_time"]
    pay_max = np.maximum(cp * (grid - K), 0.0).max(axis=0)
    slack = float(np.sum(pay_max[~same])) / grid.shape[1]
    ok = same.mean() >= 0.995 and np.allclose(val[same], ref["stop_value"][same], rtol=1e-12, atol=1e-13 * S0) and \
        abs(res.price - ref["price"]) <= slack + 1e-11 * max(ref["price"], 1e-3 * S0)
    print(("ok  " if ok else "FAIL"), dict(n=n, anti=anti, steps=steps, degree=degree, cp=cp), f"agree {same.mean():.6f} price {res.price:.6f} ref {ref['price']:.6f} form {res.form}", flush=True)
    bad += (not ok)
print(f"{N} cases, {bad} failures")
