#!/usr/bin/env python3
"""BASELINE config 4 alone (Broadie–Kaya, 10^6 trajectories, H252 parameters), a few launches — the
workload to put under `rocprofv3 --kernel-trace --stats` for the per-kernel split of its chain
(bk_draw / bk_series / bk_invert / bk_scan / bk_fallback).  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hedgehog_jl_amd import _ffi

if len(sys.argv) > 1 and sys.argv[1] == "grid":  # the per-date exact Heston grid instead (2e5 x 12)
    n_g, st_g = 200_000, 12
    ctx = _ffi.Context(0)
    m = _ffi.make_model()
    c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n_g, st_g,
                         seeds=np.arange(1, n_g + 1, dtype=np.uint64))
    r = _ffi.hh_result()
    for _ in range(3):
        ctx.check(ctx.lib.hh_heston_exact_grid(ctx.handle, C.byref(m), C.byref(c), None, None, 0, C.byref(r)))
    print(f"exact Heston grid {n_g} x {st_g}: {r.kernel_ms:.3f} ms, {n_g * st_g / (r.kernel_ms * 1e-3):.3e} transitions/s, "
          f"cf terms per transition {r.bk_cf_terms / (n_g * st_g):.2f}")
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
ctx = _ffi.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
acc = torch.zeros(16, dtype=torch.float64, device=dev)
seed0 = torch.tensor([99], dtype=torch.int64, device=dev)
m = _ffi.make_model()
c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n)
c.seeds, c.seeds_on_device = seed0.data_ptr(), 1
ctx.enable_timing(True)
for _ in range(12):
    ctx.check(ctx.lib.hh_mc_accumulate(ctx.handle, C.byref(m), C.byref(c), acc.data_ptr(), None))
t = np.array(ctx.read_timings())
r = _ffi.hh_result()
a = acc.cpu().numpy()
ctx.lib.hh_mc_finalize(C.byref(m), C.byref(c), a.ctypes.data, C.byref(r))
print(f"BK {n} paths: chain {np.median(t[2:]):.4f} ms (min {t.min():.4f}); price {r.price:.6f} "
      f"newton_fail {r.bk_newton_fail} bisect {r.bk_bisect_fallback} cf_terms/path {r.bk_cf_terms / n:.3f}")
