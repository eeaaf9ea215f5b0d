#!/usr/bin/env python3
"""Where a date of the persistent LSM induction spends its time.  Builds diagnostic variants of the
library and runs each in a child process on the same ensembles (GPU box only; hipcc builds here,
~40 s per variant):

  -DHH_LSM_STAMPS=1           thread 0 of one workgroup stamps the phases of every date with
                              s_memrealtime (100 MHz); totals read back by hh_lsm_debug_read
  -DHH_LSM_WG=1024            1024-thread workgroups x 8 trajectories per lane (128 registers per
                              lane) instead of 512 x 16 (256 registers)
  -DHH_LSM_DEBUG=bits         1 no waiting in the all-gather, 2 no solve, 4 no partial sums /
                              reductions / publish — RESULTS WRONG, only the clock is read
usage: lsm_breakdown.py [n_pairs ...]"""
import ctypes as C
import importlib.util
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = ["B total+publish, next row issue (hidden)", "all-gather", "solve", "decisions", "moment sums + stats",
          "power sums (hidden)", "A total+publish", "loop overhead"]


def child(sizes):
    sys.path.insert(0, ROOT)
    import numpy as np
    from hedgehog_jl_amd import _ffi
    ctx = _ffi.get_context(0)
    lib, h = ctx.lib, ctx.handle
    steps, degree = 100, 5
    for n in sizes:
        m = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0, cp=-1.0)
        c = _ffi.make_config(0, 1, n, steps, antithetic=1, seeds=np.arange(1, n + 1, dtype=np.uint64))
        res = _ffi.hh_lsm_result()
        ks = []
        for _ in range(6):
            ctx.check(lib.hh_lsm_solve(h, C.byref(m), C.byref(c), degree, math.exp(-0.05 / steps),
                                       C.byref(res), None, None, None))
            ks.append(res.kernel_ms)
        st = (C.c_double * 8)()
        ctx.check(lib.hh_lsm_debug_read(h, 2 * n, steps, degree, st))
        us = [st[k] / 100.0 / (steps - 1) for k in range(8)]  # 100 ticks per µs, per date
        print(f"  n={n:8d} x2: {np.median(ks[2:]):.3f} ms (form {res.form}) price {res.price:.5f}; per date µs: "
              + ", ".join(f"{p} {u:.2f}" for p, u in zip(PHASES, us)) + f" | Σ {sum(us):.2f}", flush=True)


def main():
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "hedgehog.jl_amd", "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    sizes = [a for a in sys.argv[1:] if a.isdigit()] or ["10000", "1000000"]
    flagsets = [a for a in sys.argv[1:] if a.startswith("-D")]
    variants = [f.split(",") for f in flagsets] or [["-DHH_LSM_STAMPS=1"], ["-DHH_LSM_STAMPS=1", "-DHH_LSM_WG=1024"]]
    vdir = os.path.join(ROOT, "hedgehog.jl_amd", "lib", "variants")
    os.makedirs(vdir, exist_ok=True)
    for i, flags in enumerate(variants):
        lib = os.path.join(vdir, f"libhh_lsmdbg{i}.so")
        if not os.path.exists(lib) or os.environ.get("HH_REBUILD"):  # cross-compiled here, run on the GPU box
            b.build_library(extra_flags=tuple(flags), out=lib)
        if os.environ.get("HH_BUILD_ONLY"):
            continue
        print("== " + " ".join(flags), flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", *sizes],
                           env=dict(os.environ, HEDGEHOG_MC_LIB=lib), capture_output=True, text=True)
        print(r.stdout + (r.stderr[-2000:] if r.returncode else ""), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child([int(a) for a in sys.argv[2:]])
    else:
        main()
