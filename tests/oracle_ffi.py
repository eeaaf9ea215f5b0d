"""ctypes access to oracle/libhh_oracle.so for the tests (the oracle is the checker, never the
product).  Shares the hh_model / hh_config / hh_result layouts of include/hedgehog_mc.h."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from hedgehog_jl_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# HH_SANITIZE=1 (make test-cpu-asan): the AddressSanitizer + UBSan build of the same file, loaded into a python that
# runs with libasan preloaded
SANITIZE = os.environ.get("HH_SANITIZE") == "1"
ORACLE_LIB = os.path.join(ORACLE_DIR, "_asan", "libhh_oracle.so") if SANITIZE else os.path.join(ORACLE_DIR, "libhh_oracle.so")

_vp = C.c_void_p


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.hho_mc_solve.restype = C.c_int
        lib.hho_mc_solve.argtypes = [C.POINTER(_ffi.hh_model), C.POINTER(_ffi.hh_config),
                                     C.POINTER(_ffi.hh_result), _vp, _vp, C.c_int]
        lib.hho_replay_elems.restype = C.c_size_t
        lib.hho_replay_elems.argtypes = [C.c_uint64, C.c_uint32, C.c_int32]
        lib.hho_wiener_fill.restype = None
        lib.hho_wiener_fill.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_uint32, C.c_uint64,
                                        _vp, _vp]
        lib.hho_replay_pack.restype = None
        lib.hho_replay_pack.argtypes = [C.c_int32, C.c_uint64, C.c_uint32, _vp, _vp]
        lib.hho_philox4x32_10.restype = None
        lib.hho_philox4x32_10.argtypes = [_vp, _vp, _vp]
        lib.hho_normal_pair.restype = None
        lib.hho_normal_pair.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.hho_num_threads.restype = C.c_int

    def philox(self, ctr, key):
        c = np.asarray(ctr, dtype=np.uint32)
        k = np.asarray(key, dtype=np.uint32)
        o = np.zeros(4, dtype=np.uint32)
        self.lib.hho_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
        return o

    def normal_pair(self, key, c0, c1=0, c2=0, dom=0):
        a, b = C.c_double(), C.c_double()
        self.lib.hho_normal_pair(key, c0, c1, c2, dom, C.byref(a), C.byref(b))
        return a.value, b.value

    def replay_elems(self, n_paths, n_steps, dynamics):
        return self.lib.hho_replay_elems(n_paths, n_steps, dynamics)

    def wiener_fill(self, dynamics, rho, T, n_steps, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        out = np.zeros(self.replay_elems(len(seeds), n_steps, dynamics), dtype=np.float64)
        self.lib.hho_wiener_fill(dynamics, rho, T, n_steps, len(seeds), seeds.ctypes.data,
                                 out.ctypes.data)
        return out

    def replay_pack(self, dynamics, n_paths, n_steps, src):
        src = np.ascontiguousarray(src, dtype=np.float64)
        out = np.zeros(self.replay_elems(n_paths, n_steps, dynamics), dtype=np.float64)
        self.lib.hho_replay_pack(dynamics, n_paths, n_steps, src.ctypes.data, out.ctypes.data)
        return out

    def mc_solve(self, model, cfg, want_terminal=True, n_threads=0):
        """-> (hh_result, terminal ndarray | None, accum ndarray)"""
        res = _ffi.hh_result()
        n = cfg.n_paths * (2 if cfg.antithetic else 1)
        term = np.zeros(n, dtype=np.float64) if want_terminal else None
        acc = np.zeros(_ffi.HH_ACC_LEN, dtype=np.float64)
        rc = self.lib.hho_mc_solve(C.byref(model), C.byref(cfg), C.byref(res),
                                   term.ctypes.data if term is not None else None,
                                   acc.ctypes.data, n_threads)
        if rc != 0:
            raise RuntimeError(f"oracle hho_mc_solve rc={rc}")
        return res, term, acc

    def num_threads(self):
        return self.lib.hho_num_threads()


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"] + (["asan"] if SANITIZE else []), check=True)


def load() -> Oracle:
    src = os.path.join(ORACLE_DIR, "hh_oracle.c")
    if not os.path.exists(ORACLE_LIB) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(ORACLE_LIB)):
        build()
    return Oracle(C.CDLL(ORACLE_LIB))


# struct builders live with the ctypes binding; re-exported for the tests
make_model = _ffi.make_model
make_config = _ffi.make_config
