#!/usr/bin/env python3
"""Generates tests/golden/replay_selftest/: a SMALL file set in the exchange format of
julia/parity_replay.jl (manifest.json + little-endian .bin files, one case per kernel family), with
the CPU oracles standing in for the reference — so it pins the FORMAT, the REPLAY seams of the C-ABI
and tools/check_reference_replay.py.  It is NOT reference output (the manifest says so)."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hedgehog_jl_amd import _ffi  # noqa: E402
from oracle import bk_oracle, lsm_oracle  # noqa: E402
from tests import oracle_ffi as o  # noqa: E402

out = os.path.join(ROOT, "tests", "golden", "replay_selftest")
os.makedirs(out, exist_ok=True)
for f in os.listdir(out):
    os.remove(os.path.join(out, f))
orc = o.load()
cases = []
H = dict(S0=100.0, strike=100.0, r=0.03, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, T=1.0, cp=1.0)
B = dict(S0=100.0, strike=100.0, r=0.05, sigma=0.2, T=1.0, cp=1.0)


def wbin(name, a, dtype="<f8"):
    np.ascontiguousarray(a).astype(dtype).tofile(os.path.join(out, name))
    return name


def path_major(dyn, rho, T, steps, seeds):
    nc = 2 if dyn == 1 else 1
    tile = orc.wiener_fill(dyn, rho, T, steps, seeds)
    return tile.reshape(-1, steps, nc, 256).transpose(0, 3, 1, 2).reshape(-1, steps, nc)[:len(seeds)].copy()


def euler(name, prm, dyn, n, steps, anti=0, greeks=()):
    seeds = np.arange(1, n + 1, dtype=np.uint64)
    pm = path_major(dyn, prm.get("rho", 0.0), prm["T"], steps, seeds)
    P = len(greeks)
    sd = {g: [1.0 if j == k else 0.0 for j in range(P)] for k, g in enumerate(greeks)}
    m = _ffi.make_model(**prm, seeds=sd, n_partials=P) if P else _ffi.make_model(**prm)
    c = _ffi.make_config(dyn, 0, n, steps, antithetic=anti, em_split=1, noise_mode=1, replay=pm,
                         replay_layout=1, n_partials=P)
    r, t, _ = orc.mc_solve(m, c)
    cases.append(dict(name=name, kind="euler", dynamics="heston" if dyn == 1 else "lognormal", n_paths=n,
                      n_steps=steps, antithetic=bool(anti), model=prm, price=r.price,
                      dW=wbin(name + ".dW.bin", pm), ST=wbin(name + ".ST.bin", t),
                      greeks={g: r.dprice[k] for k, g in enumerate(greeks)},
                      layout="dW: path-major [path][step][comp]; ST: [n_paths] (+ [n_paths] mirrored)"))


euler("em_split_probe", H, 1, 64, 8)  # the manifest's first case (julia/parity_replay.jl)
euler("heston_euler", H, 1, 500, 10)
euler("heston_euler_antithetic", H, 1, 300, 8, anti=1)
euler("heston_euler_greeks", H, 1, 300, 8, greeks=("S0", "V0"))
euler("lognormal_euler", B, 0, 400, 6, greeks=("S0",))

n = 1000
z = np.random.default_rng(1).standard_normal(n)
m = _ffi.make_model(**B)
c = _ffi.make_config(0, 1, n, noise_mode=1, replay=z, compat_sqrt_alpha=1)
r, t, _ = orc.mc_solve(m, c)
cases.append(dict(name="exact_lognormal", kind="exact_lognormal", n_paths=n, model=B, compat_sqrt_alpha=True,
                  price=r.price, z=wbin("exact_lognormal.z.bin", z), ST=wbin("exact_lognormal.ST.bin", t)))

import scipy.stats as st  # noqa: E402
n = 200
rng = np.random.default_rng(2)
em1 = -math.expm1(-H["kappa"] * H["T"])
d = 4 * H["kappa"] * H["theta"] / H["sigma"]**2
lam = 4 * H["kappa"] * math.exp(-H["kappa"] * H["T"]) * H["V0"] / (H["sigma"]**2 * em1)
VT = H["sigma"]**2 * em1 / (4 * H["kappa"]) * st.ncx2.rvs(d, lam, size=n, random_state=rng)
draws = np.stack([VT, rng.uniform(size=n), rng.standard_normal(n)])
ref = bk_oracle.mc_solve(**H, discount=math.exp(-H["r"] * H["T"]), n_paths=n, seed0=0, replay=draws)
cases.append(dict(name="broadie_kaya", kind="bk", n_paths=n, model=H, price=ref["price"],
                  draws=wbin("broadie_kaya.draws.bin", draws), ST=wbin("broadie_kaya.ST.bin", ref["terminal"]),
                  layout="draws: [V_T | u | Z], n_paths each"))

n, steps, degree = 1000, 10, 3
grid = lsm_oracle.gbm_grid(np.arange(1, n + 1, dtype=np.uint64), steps, 100.0, 0.05, 0.2, 1.0, 0)
disc = math.exp(-0.05 / steps)
ref = lsm_oracle.lsm_solve(grid, 100.0, -1.0, disc, degree)
cases.append(dict(name="lsm_put", kind="lsm", n_paths=n, n_steps=steps, degree=degree, strike=100.0, cp=-1.0,
                  step_discount=disc, price=ref["price"], grid=wbin("lsm_put.grid.bin", grid),
                  tau=wbin("lsm_put.tau.bin", ref["stop_time"], "<i4"), val=wbin("lsm_put.val.bin", ref["stop_value"]),
                  layout="grid: [n_steps+1][n_paths]; tau Int32, val Float64: stopping_info"))

json.dump(dict(generated_by="tests/golden/make_replay_selftest.py (CPU oracles, Euler cases with em_split=1; "
                            "format self-test, NOT reference output)", cases=cases),
          open(os.path.join(out, "manifest.json"), "w"), indent=1)
print("wrote", out, [c["name"] for c in cases])
