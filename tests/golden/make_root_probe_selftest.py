#!/usr/bin/env python3
"""Generates tests/golden/root_probe_selftest/: the `bk_root_probe` case of julia/parity_replay.jl — every abscissa
inverse_cdf (sample_from_cf.jl:105-135) asks of its CDF, for a handful of (V_T, u) pairs — written by the CPU
restatement in EACH of its eight readings of Roots.jl's two find_zero calls (oracle/bk_oracle.py: root form x
bracketing form x caps), one case per reading.  It pins the exchange FORMAT and the deciding logic of
tools/check_reference_replay.py (which must name, for every file set, the reading that wrote it, or say what the
sample cannot tell apart).  It is NOT reference output (the manifest says so)."""
import json
import math
import os
import sys

import numpy as np
import scipy.stats as st

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bk_oracle as B  # noqa: E402

out = os.path.join(ROOT, "tests", "golden", "root_probe_selftest")
os.makedirs(out, exist_ok=True)
for f in os.listdir(out):
    os.remove(os.path.join(out, f))
H = dict(S0=100.0, strike=100.0, r=0.03, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, T=1.0, cp=1.0)
n = 24
rng = np.random.default_rng(11)
em1 = -math.expm1(-H["kappa"] * H["T"])
d = 4 * H["kappa"] * H["theta"] / H["sigma"]**2
lam = 4 * H["kappa"] * math.exp(-H["kappa"] * H["T"]) * H["V0"] / (H["sigma"]**2 * em1)
VT = H["sigma"]**2 * em1 / (4 * H["kappa"]) * st.ncx2.rvs(d, lam, size=n, random_state=rng)
U = rng.uniform(size=n)
U[16:] = [1e-9 * (k + 1) if k % 2 == 0 else 1 - 1e-9 * (k + 1) for k in range(n - 16)]  # the ladder runs for these
dist = B.LogHestonDistribution(H["S0"], H["V0"], H["kappa"], H["theta"], H["sigma"], H["rho"], H["r"], H["T"])


def wbin(name, a, dtype="<f8"):
    np.ascontiguousarray(a).astype(dtype).tofile(os.path.join(out, name))
    return name


cases = []
for rf in (0, 1):
    for bf in (0, 1):
        for cp in (0, 1):
            name = f"bk_root_probe_r{rf}b{bf}c{cp}"
            g0, gm, hh, sol, counts, xs_all = [], [], [], [], [], []
            for i in range(n):
                xs, setup = [], {}
                it = B.HestonCFIterator(float(VT[i]), dist)
                sol.append(B.sample_from_cf(float(U[i]), it, root_form=rf, bracket_form=bf, caps=cp, xs=xs, setup=setup))
                g0.append(setup["initial_guess"]); gm.append(setup["max_guess"]); hh.append(setup["h"])
                counts.append(len(xs)); xs_all += xs
            cases.append(dict(name=name, kind="bk_root_probe", n=n, model=H, written_with=dict(root_form=rf, bracket_form=bf, caps=cp),
                              VT=wbin(name + ".VT.bin", VT), u=wbin(name + ".u.bin", U),
                              initial_guess=wbin(name + ".guess.bin", g0), max_guess=wbin(name + ".max_guess.bin", gm),
                              h=wbin(name + ".h.bin", hh), sol=wbin(name + ".sol.bin", sol),
                              counts=wbin(name + ".counts.bin", counts, "<i4"), threw=wbin(name + ".threw.bin", [0] * n, "<i4"),
                              xs=wbin(name + ".xs.bin", xs_all),
                              layout="VT, u, initial_guess, max_guess, h, sol: Float64[n]; counts, threw: Int32[n]; xs: Float64[sum(counts)]"))
json.dump(dict(generated_by="tests/golden/make_root_probe_selftest.py (the CPU restatement in each of its readings; format "
                            "self-test, NOT reference output)", cases=cases),
          open(os.path.join(out, "manifest.json"), "w"), indent=1)
print("wrote", out, [c["name"] for c in cases])
