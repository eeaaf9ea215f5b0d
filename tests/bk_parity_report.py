#!/usr/bin/env python3
"""Per-regime report of the Broadie–Kaya per-trajectory parity (GPU box; the checker side of
tests/test_gpu_bk.py run as a script): how many trajectories took another decision sequence than the
oracle (flipped stopping tests), and the worst relative difference of a sample among the matched and
among the flipped ones.  Calibrates MATCHED_RTOL / FLIPPED_RTOL of the test; the output is committed
under profiles/."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hedgehog_jl_amd import _ffi
from oracle import bk_oracle
from tests.test_gpu_bk import PARAMS, gpu_bk, gpu_decisions

ctx = _ffi.get_context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for seed in (2024, 7):
    for name, prm in PARAMS.items():
        res, term, D = gpu_bk(ctx, prm, n, seed=seed, offset=5)
        dec, ln = gpu_decisions(ctx, n)
        ref = bk_oracle.mc_solve(**prm, discount=D, n_paths=n, seed0=seed, path_offset=5)
        rel = np.abs(term - ref["terminal"]) / ref["terminal"]
        same = (dec == ref["decisions"]) & (ln == ref["series_len"])
        fl = ~same
        only_len = (dec == ref["decisions"]) & (ln != ref["series_len"])
        print(f"{name:18s} seed {seed:5d}: matched {same.sum():4d}/{n} worst rel {rel[same].max():.2e} median {np.median(rel[same]):.1e} | "
              f"flipped {fl.sum():3d} (series length only: {only_len.sum()}) worst rel {rel[fl].max() if fl.any() else 0:.2e} | "
              f"price rel {abs(res.price - ref['price']) / ref['price']:.1e}", flush=True)
