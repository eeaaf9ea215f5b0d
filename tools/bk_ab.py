#!/usr/bin/env python3
"""Interleaved A/B of Broadie–Kaya build variants (hedgehog.jl_amd/lib/variants/libhh_bk_*.so) against the
shipped library in ONE process: config 4 (10^6 trajectories) chain time by HIP events.  GPU box only."""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from hedgehog_jl_amd import _ffi

GRID = len(sys.argv) > 1 and sys.argv[1] == "grid"  # the exact Heston grid 2e5 x 12 instead of config 4
n = 200_000 if GRID else int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
libs = {"shipped": _ffi.LIB_PATH}
for f in sorted(glob.glob(os.path.join(ROOT, "hedgehog.jl_amd", "lib", "variants", "libhh_bk_*.so"))):
    if "stamps" in os.path.basename(f):  # diagnostic builds with in-kernel stamps (tools/bk_tile_timeline.py)
        continue
    libs[os.path.basename(f)[9:-3]] = f
seed0 = torch.tensor([99], dtype=torch.int64, device="cuda")
acc = torch.zeros(16, dtype=torch.float64, device="cuda")
m = _ffi.make_model()
c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, 12 if GRID else 1)
if GRID:
    seeds = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
    c.seeds, c.seeds_on_device = seeds.data_ptr(), 1
else:
    c.seeds, c.seeds_on_device = seed0.data_ptr(), 1
ctxs = {}
for tag, path in libs.items():
    lib = C.CDLL(path)
    for name, res, args in _ffi.SYMBOLS:
        if not hasattr(lib, name):  # an older build kept for comparison
            continue
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    h = C.c_void_p()
    assert lib.hh_ctx_create(C.byref(h), 0) == 0
    lib.hh_ctx_enable_timing(h, 1)
    ctxs[tag] = (lib, h)
times = {t: [] for t in ctxs}
sums = {}
for r in range(6):
    for tag, (lib, h) in ctxs.items():
        if GRID:
            res = _ffi.hh_result()
            for _ in range(3):
                assert lib.hh_heston_exact_grid(h, C.byref(m), C.byref(c), None, None, 0, C.byref(res)) == 0
                if r:
                    times[tag].append(res.kernel_ms)
            sums[tag] = float(res.bk_cf_terms)
            continue
        for _ in range(6):
            assert lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc.data_ptr(), None) == 0
        buf = (C.c_double * 256)()
        k = C.c_int32()
        lib.hh_ctx_read_timings(h, buf, 256, C.byref(k))
        if r:
            times[tag] += [buf[i] for i in range(k.value)]
        sums[tag] = float(acc[0].item())
for tag, t in times.items():
    print(f"{tag:10s} n={n}: median {np.median(t):.4f} ms  min {min(t):.4f}  same_sum={sums[tag] == sums['shipped']}")
