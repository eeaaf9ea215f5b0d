#!/usr/bin/env python3
"""PCIe-inclusive rate of the REPLAY path when the boundary is handed HOST increments (never the
bench `value`; recorded in DESIGN.md §5).  10^6 x 252 Heston increments = 4.03 GB per call."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hedgehog_jl_amd as hh  # noqa: E402
from hedgehog_jl_amd import _ffi  # noqa: E402

ctx = hh.Context(0)
lib, h = ctx.lib, ctx.handle
N, M = 1_000_000, 252
seeds = torch.arange(1, N + 1, dtype=torch.int64, device="cuda")
m = _ffi.make_model()
n_el = lib.hh_replay_elems(N, M, 1)
dW = torch.empty(n_el, dtype=torch.float64, device="cuda")
ctx.check(lib.hh_wiener_fill(h, 1, m.rho, m.T, M, N, seeds.data_ptr(), 1, dW.data_ptr()))
ctx.synchronize()
host_pageable = dW.cpu().numpy()
host_pinned = torch.empty(n_el, dtype=torch.float64, pin_memory=True)
host_pinned.copy_(dW.cpu())
for label, ptr in (("pageable", host_pageable.ctypes.data), ("pinned", host_pinned.data_ptr())):
    c = _ffi.make_config(1, 0, N, M, noise_mode=1)
    c.replay = ptr
    r = _ffi.hh_result()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.check(lib.hh_mc_solve(h, C.byref(m), C.byref(c), C.byref(r), None))
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    print(f"{label:9s}: {t * 1e3:8.1f} ms per solve = {N * M / t:.3e} path-steps/s "
          f"({16 * N * M / t / 1e9:.1f} GB/s over PCIe), price {r.price:.6f}")
