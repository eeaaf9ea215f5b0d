#!/usr/bin/env python3
"""Per-trajectory parity of the HIP path against draws exported from the REFERENCE itself
(julia/parity_replay.jl).  Usage, on the MI355X box:  python tools/check_reference_replay.py meta.json

Feeds the exported increments through HH_NOISE_REPLAY (path-major layout) for both Euler step forms
and prints max |S_gpu - S_ref| / S_ref and |price_gpu - price_ref| / price_ref; the form that matches
to ~1e-12 is the integrator's.  This is the check that would turn "parity unpinned" into "pinned"."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import hedgehog_jl_amd as hh  # noqa: E402
from hedgehog_jl_amd import _ffi  # noqa: E402


def main():
    meta = json.load(open(sys.argv[1]))
    base = os.path.dirname(os.path.abspath(sys.argv[1]))
    n, steps = meta["n_paths"], meta["n_steps"]
    dW = np.fromfile(os.path.join(base, meta["dW"]), dtype="<f8")
    assert dW.size == n * steps * 2, "dW.bin must hold n_paths*n_steps*2 doubles ([path][step][comp])"
    S_ref = np.fromfile(os.path.join(base, meta["ST"]), dtype="<f8")
    ctx = hh.get_context(0)
    m = _ffi.make_model(S0=meta["S0"], V0=meta["V0"], kappa=meta["kappa"], theta=meta["theta"],
                        sigma=meta["sigma"], rho=meta["rho"], r=meta["r"], T=meta["T"],
                        strike=meta["strike"], cp=meta["cp"])
    for split in (1, 0):
        c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, n, steps, em_split=split,
                             noise_mode=_ffi.HH_NOISE_REPLAY, replay=dW,
                             replay_layout=_ffi.HH_REPLAY_PATH_MAJOR)
        res = _ffi.hh_result()
        term = np.zeros(n)
        ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(res),
                                      term.ctypes.data))
        print(f"em_split={split}: max rel |S_gpu-S_ref| = {np.max(np.abs(term - S_ref) / S_ref):.3e}, "
              f"price rel err = {abs(res.price - meta['price']) / abs(meta['price']):.3e}")


if __name__ == "__main__":
    main()
