#!/usr/bin/env python3
"""Interleaved A/B timing of build variants of the path-major REPLAY kernel (euler_pm_kernel) in ONE
process, against the tile-major kernel and the repack route, on H252 10^6 x 252.

    python tools/tune_pm.py build            # here (cross-compile the variants)
    python tools/tune_pm.py run [rounds]     # on the GPU box
    HH_TUNE_MODE = price | anti | greeks1 | greeks3 ;  HH_VARIANTS = json {tag: [flags]} ;
    HH_TUNE_LAYOUT = tile: the variants run the tile-major kernel (A/B of euler_kernel builds)
"""
import ctypes as C
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "hedgehog.jl_amd", "lib", "variants")

VARIANTS = {
    "default": [],
    "w2": ["-DHH_PM_MAXW=2"],
    "w3": ["-DHH_PM_MAXW=3"],
    "w5": ["-DHH_PM_MAXW=5"],
    "w4_plain": ["-DHH_PM_NT=0"],
    "w2_plain": ["-DHH_PM_MAXW=2", "-DHH_PM_NT=0"],
}
if os.environ.get("HH_VARIANTS"):
    VARIANTS = json.loads(os.environ["HH_VARIANTS"])


def _builder():
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "hedgehog.jl_amd", "_build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build():
    b = _builder()
    os.makedirs(VDIR, exist_ok=True)
    for tag, flags in VARIANTS.items():
        print(tag, b.build_library(extra_flags=tuple(flags), out=os.path.join(VDIR, f"libhh_{tag}.so")))


def run(rounds=5):
    import numpy as np
    import torch
    from hedgehog_jl_amd import _ffi
    n, steps = int(os.environ.get("HH_TUNE_PATHS", 1_000_000)), int(os.environ.get("HH_TUNE_STEPS", 252))
    dev = torch.device("cuda", 0)
    seeds = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
    libs = {}
    for tag in VARIANTS:
        # tag "main" = the shipped library itself (PMC passes: rocprofv3 ... -- python3 tools/tune_pm.py run 1)
        lib = C.CDLL(_ffi.LIB_PATH if tag == "main" else os.path.join(VDIR, f"libhh_{tag}.so"))
        for name, res, args in _ffi.SYMBOLS:
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        h = C.c_void_p()
        assert lib.hh_ctx_create(C.byref(h), 0) == 0
        lib.hh_ctx_enable_timing(h, 1)
        libs[tag] = (lib, h)
    lib0, h0 = next(iter(libs.values()))
    mode = os.environ.get("HH_TUNE_MODE", "price")
    sd = {"price": None, "anti": None, "greeks1": {"V0": [1.0]},
          "greeks3": {"S0": [1, 0, 0], "V0": [0, 1, 0], "r_drift": [0, 0, 1],
                      "discount": [0, 0, -float(np.exp(-0.03))]}}[mode]
    P = {"greeks1": 1, "greeks3": 3}.get(mode, 0)
    m = _ffi.make_model(seeds=sd, n_partials=P)
    dW = torch.empty(lib0.hh_replay_elems(n, steps, 1), dtype=torch.float64, device=dev)
    assert lib0.hh_wiener_fill(h0, 1, m.rho, m.T, steps, n, seeds.data_ptr(), 1, dW.data_ptr()) == 0
    lib0.hh_ctx_synchronize(h0)
    pm = dW.view(-1, steps, 2, 256).permute(0, 3, 1, 2).reshape(-1, steps, 2)[:n].contiguous()
    torch.cuda.synchronize()
    acc = torch.zeros(16, dtype=torch.float64, device=dev)

    def cfg(layout, buf):
        c = _ffi.make_config(1, 0, n, steps, noise_mode=1, antithetic=int(mode == "anti"), n_partials=P,
                             replay_layout=layout)
        c.replay, c.replay_on_device = buf.data_ptr(), 1
        return c

    c_pm, c_tile = cfg(1, pm), cfg(0, dW)
    c_var = c_tile if os.environ.get("HH_TUNE_LAYOUT") == "tile" else c_pm  # which kernel the variants differ in
    cases = [(tag, lib, h, c_var) for tag, (lib, h) in libs.items()] + [("tile-major", lib0, h0, c_tile)]
    times = {t: [] for t, *_ in cases}
    sums = {}
    for r in range(rounds + 1):
        for tag, lib, h, c in cases:
            for _ in range(10):
                assert lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc.data_ptr(), None) == 0, \
                    lib.hh_last_error(h)
            buf = (C.c_double * 256)()
            k = C.c_int32()
            lib.hh_ctx_read_timings(h, buf, 256, C.byref(k))
            if r > 0:
                times[tag] += [buf[i] for i in range(k.value)]
            sums[tag] = float(acc[0].item())
    ref = sums["tile-major"]
    for tag, t in times.items():
        t = np.array(t)
        gbs = 16.0 * n * steps / (np.median(t) * 1e-3) / 1e9
        print(f"{tag:12s} median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}  "
              f"-> {gbs:7.1f} GB/s ({gbs / 80:.1f}% of 8 TB/s)  same_sum={sums[tag] == ref}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
