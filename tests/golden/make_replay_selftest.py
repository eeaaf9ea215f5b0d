#!/usr/bin/env python3
"""Generates tests/golden/replay_selftest/: a SMALL file set in the exchange format of
julia/parity_replay.jl, with the CPU oracle standing in for the reference (so it pins the FORMAT and
the tool tools/check_reference_replay.py — it is NOT reference output)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hedgehog_jl_amd import _ffi  # noqa: E402
from tests import oracle_ffi as o  # noqa: E402

out = os.path.join(ROOT, "tests", "golden", "replay_selftest")
os.makedirs(out, exist_ok=True)
orc = o.load()
n, steps = 500, 10
seeds = np.arange(1, n + 1, dtype=np.uint64)
tile = orc.wiener_fill(1, -0.7, 1.0, steps, seeds)
pm = tile.reshape(-1, steps, 2, 256).transpose(0, 3, 1, 2).reshape(-1, steps, 2)[:n].copy()
m = _ffi.make_model()
c = _ffi.make_config(1, 0, n, steps, em_split=1, noise_mode=1, replay=pm, replay_layout=1)
r, t, _ = orc.mc_solve(m, c)
pm.astype("<f8").tofile(os.path.join(out, "dW.bin"))
t.astype("<f8").tofile(os.path.join(out, "ST.bin"))
json.dump(dict(n_paths=n, n_steps=steps, S0=100.0, strike=100.0, r=0.03, V0=0.04, kappa=2.0,
               theta=0.04, sigma=0.3, rho=-0.7, T=1.0, cp=1.0, price=r.price, dW="dW.bin",
               ST="ST.bin", layout="path-major [path][step][comp] float64 LE",
               generated_by="tests/golden/make_replay_selftest.py (CPU oracle, em_split=1; format "
                            "self-test, NOT reference output)"),
          open(os.path.join(out, "meta.json"), "w"), indent=1)
print("wrote", out, r.price)
