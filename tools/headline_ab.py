#!/usr/bin/env python3
"""The headline launch (10^6 x 252 Heston Euler, REPLAY and GENERATE) timed through hh_mc_accumulate for the
libraries named on the command line, interleaved in one process-per-library sequence on ONE box — what a change
of the kernels cost or gained, free of box-to-box clock differences.  usage: headline_ab.py <lib.so> [<lib.so> …]
(each library runs in a child process; this file with --child does the timing).  HH_AB_TIMING=1: with the library's
timing hook on, as bench.py's timed loop runs it (hh_ctx_enable_timing) — then also the hook's mean kernel time."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import ctypes as C

    import numpy as np
    sys.path.insert(0, ROOT)
    from hedgehog_jl_amd import _ffi
    import torch  # noqa: F401  (the HIP runtime the library shares)
    lib = C.CDLL(os.environ["HEDGEHOG_MC_LIB"], mode=C.RTLD_GLOBAL)  # only entry points every round has
    for name, res, args in _ffi.SYMBOLS:
        if hasattr(lib, name):
            getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    h = C.c_void_p()
    assert lib.hh_ctx_create(C.byref(h), 0) == 0
    N, M = 1_000_000, 252
    p = C.c_void_p()
    assert lib.hh_device_malloc(h, 8 * N, C.byref(p)) == 0
    seeds = np.arange(1, N + 1, dtype=np.uint64)
    assert lib.hh_memcpy_h2d(h, p, seeds.ctypes.data, 8 * N) == 0
    dW, acc = C.c_void_p(), C.c_void_p()
    assert lib.hh_device_malloc(h, 8 * lib.hh_replay_elems(N, M, 1), C.byref(dW)) == 0
    assert lib.hh_device_malloc(h, 8 * 16, C.byref(acc)) == 0
    m = _ffi.make_model()
    assert lib.hh_wiener_fill(h, 1, m.rho, m.T, M, N, p, 1, dW) == 0
    out = {}
    for name, noise, reps in (("replay", 1, 300), ("generate", 0, 100)):
        c = _ffi.make_config(1, 0, N, M, noise_mode=noise)
        c.seeds, c.seeds_on_device, c.replay, c.replay_on_device = p.value, 1, dW.value, 1
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            for _ in range(8):
                assert lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc, None) == 0
            lib.hh_ctx_synchronize(h)
        walls, kern = [], []
        hook = os.environ.get("HH_AB_TIMING") == "1"
        reps = min(reps, 200) if hook else reps  # the hook keeps 256 slots
        for _ in range(3):
            lib.hh_ctx_synchronize(h)
            if hook:
                lib.hh_ctx_enable_timing(h, 1)
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc, None)
            lib.hh_ctx_synchronize(h)
            walls.append((time.perf_counter() - t0) / reps * 1e3)
            if hook:
                buf, k = (C.c_double * 256)(), C.c_int32()
                lib.hh_ctx_read_timings(h, buf, 256, C.byref(k))
                kern.append(sum(buf[i] for i in range(k.value)) / max(k.value, 1))
                lib.hh_ctx_enable_timing(h, 0)
        out[name + "_wall_ms_per_solve"] = [round(w, 5) for w in walls]
        if hook:
            out[name + "_hook_kernel_ms"] = [round(x, 5) for x in kern]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        libs = sys.argv[1:]
        for rnd in range(2):
            for lib in libs:
                env = dict(os.environ, HEDGEHOG_MC_LIB=os.path.abspath(lib))
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                print(os.path.basename(lib), "round", rnd, line[-1] if line else p.stderr[-400:], flush=True)
