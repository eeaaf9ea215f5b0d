#!/usr/bin/env python3
"""Interleaved A/B timing of REPLAY-kernel build variants in ONE process (guide rule 24).
Builds hedgehog.jl_amd/lib/variants/libhh_<tag>.so with -D knobs, then on the GPU box times each
variant round-robin on the H252 10^6 x 252 workload and prints median / min kernel ms.

    python tools/tune_replay.py build            # here (cross-compile)
    python tools/tune_replay.py run [rounds]     # on the GPU box
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "hedgehog.jl_amd", "lib", "variants")
CSRC = os.path.join(ROOT, "hedgehog.jl_amd", "csrc")

VARIANTS = {
    "c4_w1": ["-DHH_REPLAY_CHUNK=4", "-DHH_REPLAY_MINW=1"],
    "c4_w4": ["-DHH_REPLAY_CHUNK=4", "-DHH_REPLAY_MINW=4"],
    "c2_w4": ["-DHH_REPLAY_CHUNK=2", "-DHH_REPLAY_MINW=4"],
    "c8_w2": ["-DHH_REPLAY_CHUNK=8", "-DHH_REPLAY_MINW=2"],
    "c4_w4_nt": ["-DHH_REPLAY_CHUNK=4", "-DHH_REPLAY_MINW=4", "-DHH_REPLAY_NT=1"],
    "c4_w1_nt": ["-DHH_REPLAY_CHUNK=4", "-DHH_REPLAY_MINW=1", "-DHH_REPLAY_NT=1"],
    "c2_w4_nt": ["-DHH_REPLAY_CHUNK=2", "-DHH_REPLAY_MINW=4", "-DHH_REPLAY_NT=1"],
    "c4_w8": ["-DHH_REPLAY_CHUNK=4", "-DHH_REPLAY_MINW=8"],
    "c4_w8_nt": ["-DHH_REPLAY_CHUNK=4", "-DHH_REPLAY_MINW=8", "-DHH_REPLAY_NT=1"],
    "c3_w8": ["-DHH_REPLAY_CHUNK=3", "-DHH_REPLAY_MINW=8"],
}
if os.environ.get("HH_VARIANTS"):
    VARIANTS = json.loads(os.environ["HH_VARIANTS"])


def build():
    """Every variant = the product sources with -DHH_REPLAY_VARIANTS -Itools/variants (the experiment paths
    that did not ship live there) and the variant's -D knobs, built by the product's own recipe."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_hh_build", os.path.join(ROOT, "hedgehog.jl_amd", "_build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    os.makedirs(VDIR, exist_ok=True)
    for tag, flags in VARIANTS.items():
        out = os.path.join(VDIR, f"libhh_{tag}.so")
        mod.build_library(force=True, out=out,
                          extra_flags=["-DHH_REPLAY_VARIANTS", "-I" + os.path.join(ROOT, "tools", "variants"), *flags])
        print(tag, "->", out)


def run(rounds=7):
    import numpy as np
    import torch
    from hedgehog_jl_amd import _ffi
    n_paths, n_steps = 1_000_000, 252
    dev = torch.device("cuda", 0)
    seeds = torch.arange(1, n_paths + 1, dtype=torch.int64, device=dev)
    libs = {}
    for tag in VARIANTS:
        lib = C.CDLL(os.path.join(VDIR, f"libhh_{tag}.so"))
        for name, res, args in _ffi.SYMBOLS:
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        h = C.c_void_p()
        assert lib.hh_ctx_create(C.byref(h), 0) == 0
        lib.hh_ctx_enable_timing(h, 1)
        libs[tag] = (lib, h)
    lib0, h0 = next(iter(libs.values()))
    mode = os.environ.get("HH_TUNE_MODE", "price")   # price | anti | greeks3
    if mode == "greeks3":
        m = _ffi.make_model(seeds={"S0": [1, 0, 0], "V0": [0, 1, 0], "r_drift": [0, 0, 1],
                                   "discount": [0, 0, -float(np.exp(-0.03))]}, n_partials=3)
    else:
        m = _ffi.make_model()
    dW = torch.empty(lib0.hh_replay_elems(n_paths, n_steps, 1), dtype=torch.float64, device=dev)
    assert lib0.hh_wiener_fill(h0, 1, m.rho, m.T, n_steps, n_paths, seeds.data_ptr(), 1,
                               dW.data_ptr()) == 0
    lib0.hh_ctx_synchronize(h0)
    acc = torch.zeros(16, dtype=torch.float64, device=dev)
    c = _ffi.make_config(1, 0, n_paths, n_steps, noise_mode=1, antithetic=int(mode == "anti"),
                         n_partials=3 if mode == "greeks3" else 0)
    c.replay, c.replay_on_device = dW.data_ptr(), 1
    times = {t: [] for t in libs}
    prices = {}
    for r in range(rounds + 1):
        for tag, (lib, h) in libs.items():
            for _ in range(5):
                assert lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc.data_ptr(), None) == 0
            buf = (C.c_double * 256)()
            n = C.c_int32()
            lib.hh_ctx_read_timings(h, buf, 256, C.byref(n))
            if r > 0:
                times[tag] += [buf[i] for i in range(n.value)]
            prices[tag] = float(acc[0].item())
    # sustained regime (what bench.py sees): 40 back-to-back launches per variant, twice, alternating
    sustained = {t: [] for t in libs}
    for rep in range(2):
        for tag, (lib, h) in libs.items():
            buf = (C.c_double * 256)()
            n = C.c_int32()
            for _ in range(40):
                assert lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc.data_ptr(), None) == 0
            lib.hh_ctx_read_timings(h, buf, 256, C.byref(n))
            sustained[tag] += [buf[i] for i in range(10, n.value)]
    ref = next(iter(prices.values()))
    for tag, t in times.items():
        t = np.array(t)
        gbs = 16.0 * n_paths * n_steps / (np.median(t) * 1e-3) / 1e9
        st = np.array(sustained[tag])
        print(f"{tag:12s} median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}  "
              f"-> {gbs:7.1f} GB/s ({gbs / 80:.1f}% of 8 TB/s)  same_sum={prices[tag] == ref}  "
              f"| sustained(40 back-to-back) mean {st.mean():.4f} ms  max {st.max():.4f}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
