"""What hh_mgpu's enqueue phase costs the host, one shard after the other vs a thread per device.

A one-GPU box has one device, so the G shards all go to device 0 (G contexts, G streams): the
per-shard figure is what ONE shard's enqueue costs (argument checks, two event records, two
launches) and is the same on an 8-GPU node; the phase wall time with threads is an UPPER bound for
that node, because here G threads contend for one device's queues and one runtime lock.

usage: python tools/mgpu_enqueue.py [--gpus G] [--paths-per-shard N] [--reps R]   -> text on stdout"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from hedgehog_jl_amd import _ffi  # noqa: E402

H252 = dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1.0, strike=100.0, cp=1.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--paths-per-shard", type=int, default=125_000)
    ap.add_argument("--nsteps", type=int, default=252)
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    G, n, steps = a.gpus, a.paths_per_shard, a.nsteps
    import torch
    n_dev = torch.cuda.device_count()
    devs = [g % n_dev for g in range(G)]
    mg = _ffi.MultiGpu(devs, _ffi.HH_MGPU_HOST_SUM)
    model = _ffi.make_model(**H252)
    keep = []

    def shards(noise):
        cfgs = []
        for g in range(G):
            ctx = mg.ctx(g)
            seeds = np.arange(g * n + 1, (g + 1) * n + 1, dtype=np.uint64)
            c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, n, steps, noise_mode=noise)
            if noise == _ffi.HH_NOISE_REPLAY:
                dW = _ffi.DeviceBuffer(ctx, 8 * ctx.lib.hh_replay_elems(n, steps, _ffi.HH_HESTON))
                ctx.check(ctx.lib.hh_wiener_fill(ctx.handle, _ffi.HH_HESTON, model.rho, model.T, steps, n,
                                                 seeds.ctypes.data, 0, dW.ptr))
                ctx.synchronize()
                c.replay, c.replay_on_device = dW.ptr, 1
                keep.append(dW)
            else:
                sd = _ffi.DeviceBuffer(ctx, 8 * n)
                ctx.check(ctx.lib.hh_memcpy_h2d(ctx.handle, sd.ptr, seeds.ctypes.data, 8 * n))
                c.seeds, c.seeds_on_device = sd.ptr, 1
                keep.append(sd)
            cfgs.append(c)
        return cfgs

    print(f"# hh_mgpu enqueue cost: {G} shards on devices {devs} ({n_dev} GPU(s) on this box), "
          f"Heston Euler {n} x {steps} per shard, device-resident inputs, host ordered sum, {a.reps} solves each")
    print(f"# {'noise':8s} {'enqueue':8s} {'shard_us (median of every shard)':>34s} {'shard_us max':>13s} "
          f"{'phase_us':>9s} {'total_ms':>9s} {'kernel_ms':>10s}")
    for noise, label in ((_ffi.HH_NOISE_REPLAY, "REPLAY"), (_ffi.HH_NOISE_GENERATE, "GENERATE")):
        cfgs = shards(noise)
        prices = {}
        for mode, mlabel in ((_ffi.HH_MGPU_ENQUEUE_SERIAL, "serial"), (_ffi.HH_MGPU_ENQUEUE_THREADS, "threads")):
            mg.set_option(_ffi.HH_MGPU_OPT_ENQUEUE, mode)
            for _ in range(5):
                r = mg.solve_shards(model, cfgs)
            per_all, phase, total, kern = [], [], [], []
            for _ in range(a.reps):
                r = mg.solve_shards(model, cfgs)
                per, whole = mg.enqueue_stats()
                per_all.append(per)
                phase.append(whole)
                total.append(r.total_ms)
                kern.append(r.kernel_ms)
            prices[mlabel] = r.price
            per_all = np.array(per_all)
            print(f"  {label:8s} {mlabel:8s} {np.median(per_all):34.1f} {np.median(per_all.max(axis=1)):13.1f} "
                  f"{np.median(phase):9.1f} {np.median(total):9.3f} {np.median(kern):10.3f}")
            # a solve after the workers have parked (idle for longer than their polling window)
            time.sleep(0.05)
            r = mg.solve_shards(model, cfgs)
            _, whole = mg.enqueue_stats()
            print(f"  {label:8s} {mlabel:8s} first solve after 50 ms idle: phase_us {whole:.1f} total_ms {r.total_ms:.3f}")
        assert prices["serial"] == prices["threads"], prices
    mg.close()
    lsm(a, devs)


def lsm(a, devs):
    """The sharded LSM induction inside the library: 2 + (dates - 1) + 1 exchanges, each followed by one launch
    per device — the per-date host path (host ordered sum here; an RCCL all-reduce on a node)."""
    import ctypes as C
    import math
    G = min(len(devs), 4)
    n, steps, degree = 200_000, 100, 5
    m = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0, cp=-1.0)
    c = _ffi.make_config(_ffi.HH_LOGNORMAL, _ffi.HH_EXACT_LAW, n, steps, antithetic=1,
                         seeds=np.arange(1, n + 1, dtype=np.uint64))
    print(f"# hh_mgpu_lsm_solve: {2 * n} trajectories x {steps} dates, degree {degree}, {G} shards on devices {devs[:G]}, host ordered sum")
    mg = _ffi.MultiGpu(devs[:G], _ffi.HH_MGPU_HOST_SUM)
    prices = []
    for mode, label in ((_ffi.HH_MGPU_ENQUEUE_SERIAL, "serial"), (_ffi.HH_MGPU_ENQUEUE_THREADS, "threads")):
        mg.set_option(_ffi.HH_MGPU_OPT_ENQUEUE, mode)
        res = _ffi.hh_lsm_result()
        t = []
        for _ in range(6):
            mg.check(mg.lib.hh_mgpu_lsm_solve(mg.handle, C.byref(m), C.byref(c), degree, math.exp(-0.05 / steps),
                                              C.byref(res), None, None))
            t.append(res.total_ms)
        prices.append(res.price)
        print(f"  enqueue {label:8s} total_ms median {np.median(t[1:]):8.3f}  ({np.median(t[1:]) * 1e3 / (steps + 2):6.1f} us per exchange + phase)  price {res.price:.6f}")
    assert prices[0] == prices[1]
    one = _ffi.MultiGpu(devs[:1], _ffi.HH_MGPU_HOST_SUM)
    res = _ffi.hh_lsm_result()
    for _ in range(3):
        one.check(one.lib.hh_mgpu_lsm_solve(one.handle, C.byref(m), C.byref(c), degree, math.exp(-0.05 / steps),
                                            C.byref(res), None, None))
    print(f"  one device (the fused persistent induction): total_ms {res.total_ms:.3f}")
    mg.close()
    one.close()


if __name__ == "__main__":
    main()
