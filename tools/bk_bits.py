#!/usr/bin/env python3
"""Bitwise comparison of Broadie–Kaya terminal samples between the shipped library and variants
(hedgehog.jl_amd/lib/variants/libhh_bk_*.so): hh_mc_solve on the parameter sets of tests/test_gpu_bk.py, one small
ensemble each, every sample compared as a 64-bit pattern.  What tools/bk_ab.py's `same_sum` cannot show.  GPU box only."""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from hedgehog_jl_amd import _ffi
from tests.test_gpu_bk import PARAMS
from tests import oracle_ffi as o

libs = {"shipped": _ffi.LIB_PATH}
for f in sorted(glob.glob(os.path.join(ROOT, "hedgehog.jl_amd", "lib", "variants", "libhh_bk_*.so"))):
    libs[os.path.basename(f)[9:-3]] = f
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
out = {}
for tag, path in libs.items():
    lib = C.CDLL(path)
    for name, res, args in _ffi.SYMBOLS:
        if not hasattr(lib, name):  # an older build kept for comparison
            continue
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    h = C.c_void_p()
    assert lib.hh_ctx_create(C.byref(h), 0) == 0
    for name, prm in PARAMS.items():
        m = o.make_model(**prm)
        c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, seeds=[4242])
        res = _ffi.hh_result()
        term = np.zeros(n)
        assert lib.hh_mc_solve(h, C.byref(m), C.byref(c), C.byref(res), term.ctypes.data) == 0
        out[(tag, name)] = term.view(np.uint64).copy()
for tag in libs:
    if tag == "shipped":
        continue
    for name in PARAMS:
        a, b = out[("shipped", name)], out[(tag, name)]
        d = int((a != b).sum())
        rel = np.abs(a.view(np.float64) - b.view(np.float64)) / np.abs(a.view(np.float64))
        print(f"{tag:10s} {name:18s} differing samples {d:6d} of {n}  worst rel {rel.max():.2e}")
