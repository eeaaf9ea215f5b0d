#!/usr/bin/env python3
"""Interleaved A/B of GENERATE-mode Euler solves (hh_mc_accumulate, kernel time by the library's timing hook) between
the shipped library and variants (hedgehog.jl_amd/lib/variants/libhh_bk_<tag>.so named on the command line), over a
few shapes, ONE process.  GPU box only.  usage: generate_ab.py <tag> [<tag> …]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from hedgehog_jl_amd import _ffi

libs = {"shipped": _ffi.LIB_PATH}
for tag in sys.argv[1:]:
    libs[tag] = os.path.join(ROOT, "hedgehog.jl_amd", "lib", "variants", f"libhh_bk_{tag}.so")
ctxs = {}
for tag, path in libs.items():
    lib = C.CDLL(path)
    for name, res, args in _ffi.SYMBOLS:
        if hasattr(lib, name):
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
    h = C.c_void_p()
    assert lib.hh_ctx_create(C.byref(h), 0) == 0
    lib.hh_ctx_enable_timing(h, 1)
    ctxs[tag] = (lib, h)
acc = torch.zeros(16, dtype=torch.float64, device="cuda")
HES, GBM, EM = _ffi.HH_HESTON, _ffi.HH_LOGNORMAL, _ffi.HH_EULER_MARUYAMA
shapes = [(HES, 1_000_000, 252, 0, 0), (HES, 1_000_000, 252, 1, 0), (HES, 1_000_000, 252, 0, 3), (GBM, 1_000_000, 252, 0, 0),
          (HES, 500_000, 252, 0, 0), (HES, 300_000, 1000, 0, 0), (HES, 2_000_000, 64, 0, 0), (HES, 4_000_000, 252, 0, 0),
          (HES, 270_000, 252, 0, 0), (HES, 400_000, 252, 0, 0)]
for dyn, n, steps, anti, P in shapes:
    seeds = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
    kw = dict(seeds={"S0": [1.0, 0, 0], "V0": [0, 1.0, 0], "r_drift": [0, 0, 1.0]}, n_partials=3) if P else {}
    m = _ffi.make_model(**({"sigma": 0.2} if dyn == GBM else {}), **kw)
    c = _ffi.make_config(dyn, EM, n, steps, antithetic=anti, n_partials=P)
    c.seeds, c.seeds_on_device = seeds.data_ptr(), 1
    times = {t: [] for t in ctxs}
    for r in range(5):
        for tag, (lib, h) in ctxs.items():
            for _ in range(4):
                assert lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc.data_ptr(), None) == 0
            buf, k = (C.c_double * 256)(), C.c_int32()
            lib.hh_ctx_read_timings(h, buf, 256, C.byref(k))
            if r:
                times[tag] += [buf[i] for i in range(k.value)]
    base = np.median(times["shipped"])
    line = f"{'Heston' if dyn == HES else 'GBM':6s} {n:8d} x {steps:4d} anti={anti} P={P}: " + "  ".join(
        f"{t} {np.median(v):.4f} ms ({(np.median(v) / base - 1) * 100:+.2f} %)" for t, v in times.items())
    print(line, flush=True)
