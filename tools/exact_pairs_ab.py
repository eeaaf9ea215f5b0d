#!/usr/bin/env python3
"""Exact lognormal law (BASELINE config 2): pairs of trajectories per lane x who adds the records, at the sizes
named on the command line (default 10^6) — each combination in a child process of its own ($HEDGEHOG_MC_EXACT_PAIRS
is read once per process), the list run twice over, on ONE box.  Wall time per solve of a back-to-back
hh_mc_accumulate loop (no host synchronisation inside), microseconds.  GPU box only.
usage: exact_pairs_ab.py [n_paths …]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(sizes):
    import ctypes as C

    import numpy as np
    sys.path.insert(0, ROOT)
    from hedgehog_jl_amd import _ffi
    ctx = _ffi.Context(0)
    fuse = int(os.environ["HH_AB_FUSE"])
    ctx.set_option(_ffi.HH_OPT_FUSE_REDUCE, fuse)
    seed = _ffi.DeviceBuffer(ctx, 8).upload(np.array([77], dtype=np.uint64))
    acc = _ffi.DeviceBuffer(ctx, 8 * _ffi.HH_ACC_LEN)
    m = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0)
    out = {}
    for n in sizes:
        c = _ffi.make_config(_ffi.HH_LOGNORMAL, _ffi.HH_EXACT_LAW, n, 1)
        c.seeds, c.seeds_on_device = seed.ptr, 1
        for _ in range(50):
            ctx.check(ctx.lib.hh_mc_accumulate(ctx.handle, C.byref(m), C.byref(c), acc.ptr, None))
        ctx.synchronize()
        walls = []
        reps = 400 if n <= 10_000_000 else 100
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.lib.hh_mc_accumulate(ctx.handle, C.byref(m), C.byref(c), acc.ptr, None)
            ctx.synchronize()
            walls.append((time.perf_counter() - t0) / reps * 1e6)
        a = acc.download(np.empty(_ffi.HH_ACC_LEN))
        out[str(n)] = {"us": round(min(walls), 2), "sum": a[_ffi.HH_ACC_SUM].hex()}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child([int(a) for a in sys.argv[1:] if a != "--child"])
    else:
        sizes = [a for a in sys.argv[1:]] or ["1000000"]
        for rnd in range(2):
            for pairs in ("1", "2", "4", "8"):
                for fuse in ("0", "1"):
                    env = dict(os.environ, HEDGEHOG_MC_EXACT_PAIRS=pairs, HH_AB_FUSE=fuse)
                    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", *sizes], env=env,
                                       capture_output=True, text=True)
                    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                    print(f"round {rnd} pairs/lane {pairs} {'in-kernel reducer' if fuse == '1' else 'separate kernel  '}",
                          line[-1] if line else p.stderr[-400:], flush=True)
