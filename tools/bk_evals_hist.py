#!/usr/bin/env python3
"""Broadie–Kaya, config 4: how many CDF evaluations the secant of each trajectory makes (decision word, bits 0-7),
and what a WAVE of 64 consecutive trajectories pays for it (it runs until its last lane is done): the divergence of
the inversion phase, and what handing the unconverged lanes of a tile to one wave after k evaluations would save.
GPU box only.  usage: bk_evals_hist.py [n_paths]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hedgehog_jl_amd import _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = _ffi.get_context(0)
m = _ffi.make_model()
c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, 1, seeds=np.arange(1, n + 1, dtype=np.uint64))
r = _ffi.hh_result()
ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(r), None))
dec, ln = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
ctx.check(ctx.lib.hh_bk_decisions(ctx.handle, n, dec.ctypes.data, ln.ctypes.data))
ev = (dec & 0xFF).astype(np.int64)
failed = (dec >> 8) & 3
print(f"n = {n}, price {r.price:.6f}, newton_fail {r.bk_newton_fail}, series length mean {ln.mean():.2f}")
h = np.bincount(ev, minlength=12)
for k, v in enumerate(h):
    if v:
        print(f"  {k:2d} evaluations: {v:8d}  {v / n:7.4f}   of them to the ladder: {(failed[ev == k] != 0).mean():.3f}")
full = n // 256 * 256
w = ev[:full].reshape(-1, 64)
t = ev[:full].reshape(-1, 256)
print(f"mean evaluations per trajectory {ev.mean():.3f}; per wave (max of 64) {w.max(1).mean():.3f}; "
      f"active lanes in the secant {ev.mean() / w.max(1).mean():.3f}")
lw = ln[:full].reshape(-1, 64)
print(f"series length: mean {ln.mean():.3f}, per wave (max of 64) {lw.max(1).mean():.3f}")
for k in range(3, 10):
    left = (t > k).sum(1)                       # a tile's trajectories not done after k evaluations
    waves_after = np.ceil(left / 64.0)          # compacted: waves that go on
    cost = 4 * k + (np.maximum(t.max(1) - k, 0) * waves_after)  # wave-evaluations of the tile (upper bound: all go to the cap)
    print(f"  hand-over after {k}: {left.mean():6.1f} trajectories of a tile left ({(left > 64).mean():.3f} of tiles more than one wave); "
          f"wave-evaluations per tile {cost.mean():6.2f} against {w.max(1).reshape(-1, 4).sum(1).mean():6.2f} now")
