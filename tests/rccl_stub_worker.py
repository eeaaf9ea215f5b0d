"""hh_mgpu's RCCL branch with MORE THAN ONE rank on a one-GPU box, in a process of its own.

The library binds RCCL once per process (dlopen of $HEDGEHOG_MC_RCCL), so this runs as a child of
tests/test_gpu_mgpu_rccl_stub.py with HEDGEHOG_MC_RCCL = tests/c/libstub_rccl.so — a test-only
stand-in that carries the collectives out on the caller's streams, accepts a device listed several
times, and can be told to fail for one rank (leaving the earlier ranks' streams with a kernel that
waits for peers that never come, as RCCL would).  Every scenario compares with the host's ordered sum
on the same sharding, which the stand-in reproduces bit for bit.

usage: rccl_stub_worker.py <out.json>      (prints nothing but the scenarios' names)"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import hedgehog_jl_amd as hh  # noqa: E402,F401
from hedgehog_jl_amd import _ffi  # noqa: E402
from tests.shard_worker import problem  # noqa: E402

HES, GBM = _ffi.HH_HESTON, _ffi.HH_LOGNORMAL
EM, EXACT, BK = _ffi.HH_EULER_MARUYAMA, _ffi.HH_EXACT_LAW, _ffi.HH_BROADIE_KAYA
HESTON_SEEDS = {"S0": [1, 0, 0], "V0": [0, 1, 0], "r_drift": [0, 0, 1], "discount": [0, 0, -float(np.exp(-0.03))]}
DEV = [0, 0, 0]

stub = C.CDLL(os.environ["HEDGEHOG_MC_RCCL"])
stub.stub_rccl_fail_at.argtypes = [C.c_int, C.c_int]
stub.stub_rccl_fail_at.restype = None
stub.stub_rccl_counters.argtypes = [C.POINTER(C.c_int)]
stub.stub_rccl_counters.restype = None


def counters():
    a = (C.c_int * 4)()
    stub.stub_rccl_counters(a)
    return dict(zip(("calls", "complete", "orphans", "aborts"), a))


def seeds_for(n, salt=0):
    return np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(salt)


def bits(res, P=0):
    return [res.price, res.sum_payoff, res.sumsq_payoff, res.std_error, int(res.n_paths_done)] + \
        [res.dprice[k] for k in range(P)]


def european(mg, n=10_001, steps=17, want_terminal=True):
    m = _ffi.make_model(seeds=HESTON_SEEDS, n_partials=3)
    c = _ffi.make_config(HES, EM, n, steps, antithetic=1, seeds=seeds_for(n, 5), n_partials=3)
    t = np.zeros(2 * n) if want_terminal else None
    t0 = time.perf_counter()
    r = mg.solve(m, c, t)
    return bits(r, 3), (t.tolist() if want_terminal else None), time.perf_counter() - t0


def basket(mg, n=9_000, steps=8):
    m = _ffi.make_model()
    c = _ffi.make_config(HES, EM, n, steps, seeds=seeds_for(n, 6))
    strikes = np.array([80.0, 95.0, 100.0, 105.0, 130.0])
    cps = np.array([1.0, -1.0, 1.0, 1.0, -1.0])
    out = (_ffi.hh_result * 5)()
    mg.check(mg.lib.hh_mgpu_solve_basket(mg.handle, C.byref(m), C.byref(c), strikes.ctypes.data, cps.ctypes.data, 5, out))
    return [bits(out[k]) for k in range(5)]


def lsm(mg, case):
    from hedgehog_jl_amd.lsm import _lsm_structs
    from hedgehog_jl_amd.domain import df
    from hedgehog_jl_amd.dates import MILLISECONDS_IN_YEAR_365
    prob, method = problem(case)
    model, c, T = _lsm_structs(prob, method.mc_method)
    mkt = prob.market_inputs
    step_discount = float(df(mkt.rate, mkt.referenceDate + (T / case["steps"]) * MILLISECONDS_IN_YEAR_365))
    ntot = case["n"] * (2 if case["anti"] else 1)
    tau, val = np.empty(ntot, dtype=np.int32), np.empty(ntot)
    res = _ffi.hh_lsm_result()
    t0 = time.perf_counter()
    mg.check(mg.lib.hh_mgpu_lsm_solve(mg.handle, C.byref(model), C.byref(c), method.degree, step_discount,
                                      C.byref(res), tau.ctypes.data, val.ctypes.data))
    return [res.price, res.std_error, int(res.n_paths_total)], tau.tolist(), val.tolist(), time.perf_counter() - t0


LSM_CASE = dict(model="gbm", n=6000, steps=20, degree=4, anti=1, cp=-1.0, strike=100.0, seed=7)
LSM_CASE_H = dict(model="heston", n=2500, steps=6, degree=3, anti=0, cp=-1.0, strike=100.0, seed=9)


def main(out_path):
    out = {}
    host = _ffi.MultiGpu(DEV, _ffi.HH_MGPU_HOST_SUM)
    ref_eu = european(host)
    ref_bk = basket(host)
    ref_lsm = lsm(host, LSM_CASE)
    ref_lsm_h = lsm(host, LSM_CASE_H)

    # 1. G = 3 ranks through the RCCL branch (AUTO picks it because the stand-in initialises)
    mg = _ffi.MultiGpu(DEV)
    out["auto_mode_is_rccl"] = mg.reduce_mode == _ffi.HH_MGPU_REDUCE_RCCL
    c0 = counters()
    eu = european(mg)
    out["european_bit_equal"] = eu[0] == ref_eu[0] and eu[1] == ref_eu[1]
    out["basket_bit_equal"] = basket(mg) == ref_bk
    l1 = lsm(mg, LSM_CASE)
    out["lsm_bit_equal"] = l1[:3] == ref_lsm[:3]
    l2 = lsm(mg, LSM_CASE_H)
    out["lsm_heston_bit_equal"] = l2[:3] == ref_lsm_h[:3]
    c1 = counters()
    # one grouped all-reduce per European / basket solve; 2 + (steps - 1) + 1 per LSM solve
    out["groups"] = c1["complete"] - c0["complete"]
    out["groups_expected"] = 2 + (2 + 19 + 1) + (2 + 5 + 1)
    out["calls"] = c1["calls"] - c0["calls"]
    # the count a caller may print next to "reduce: rccl" comes from a collective, and the line can name the library
    out["selftest"] = list(mg.selftest())
    out["selftest_host_context"] = list(host.selftest())
    out["rccl_info"] = mg.rccl_info()
    # several models on the same draws, sharded: one all-reduce of the 3 x 16 block
    models = [_ffi.make_model(S0=100.1), _ffi.make_model(S0=99.9), _ffi.make_model(V0=0.05, strike=95.0)]
    cm = _ffi.make_config(HES, EM, 10_001, 17, antithetic=1, seeds=seeds_for(10_001, 5))
    c0m = counters()
    out["multi_bit_equal"] = [bits(r) for r in mg.solve_multi(models, cm)] == [bits(r) for r in host.solve_multi(models, cm)]
    out["multi_groups"] = counters()["complete"] - c0m["complete"]
    # serial enqueue gives the same bits
    mg.set_option(_ffi.HH_MGPU_OPT_ENQUEUE, _ffi.HH_MGPU_ENQUEUE_SERIAL)
    out["serial_enqueue_bit_equal"] = european(mg)[:2] == ref_eu[:2]
    per, whole = mg.enqueue_stats()
    out["enqueue_stats_ok"] = len(per) == 3 and all(0.0 < p < 1e6 for p in per) and whole >= max(per)
    mg.set_option(_ffi.HH_MGPU_OPT_ENQUEUE, _ffi.HH_MGPU_ENQUEUE_THREADS)

    # 2. AUTO: rank 1 refuses its all-reduce after rank 0 was enqueued -> the solve is finished on the host,
    #    without waiting for rank 0's orphan (which would take the stand-in's three seconds)
    stub.stub_rccl_fail_at(1, 1)
    c0 = counters()
    eu = european(mg)
    c1 = counters()
    out["auto_failure_result_bit_equal"] = eu[:2] == ref_eu[:2]
    out["auto_failure_seconds"] = eu[2]
    out["auto_failure_orphans"] = c1["orphans"] - c0["orphans"]
    out["auto_failure_aborts"] = c1["aborts"] - c0["aborts"]
    out["auto_failure_mode_is_host"] = mg.reduce_mode == _ffi.HH_MGPU_REDUCE_HOST
    out["auto_failure_text"] = mg.last_error()
    out["auto_after_failure_bit_equal"] = european(mg)[:2] == ref_eu[:2] and lsm(mg, LSM_CASE)[:3] == ref_lsm[:3]
    mg.close()

    # 3. HH_MGPU_RCCL: the same failure is a status, promptly; the context stays refused afterwards
    mg = _ffi.MultiGpu(DEV, _ffi.HH_MGPU_RCCL)
    out["strict_first_solve_bit_equal"] = european(mg)[:2] == ref_eu[:2]
    stub.stub_rccl_fail_at(2, 1)
    t0 = time.perf_counter()
    try:
        european(mg)
        out["strict_failure_code"] = 0
    except _ffi.HedgehogMCError as e:
        out["strict_failure_code"] = e.code
        out["strict_failure_text"] = str(e)
    out["strict_failure_seconds"] = time.perf_counter() - t0
    try:
        european(mg)
        out["strict_second_code"] = 0
    except _ffi.HedgehogMCError as e:
        out["strict_second_code"] = e.code
    try:
        lsm(mg, LSM_CASE)
        out["strict_lsm_code"] = 0
    except _ffi.HedgehogMCError as e:
        out["strict_lsm_code"] = e.code
    mg.close()

    # 4. the LSM induction: the 7th exchange fails for rank 2 (ranks 0 and 1 enqueued).  In place, so the
    #    local sums are gone: AUTO runs the induction again on the host sum, HH_MGPU_RCCL returns the status
    mg = _ffi.MultiGpu(DEV)
    stub.stub_rccl_fail_at(2, 7)
    c0 = counters()
    l3 = lsm(mg, LSM_CASE)
    c1 = counters()
    out["lsm_auto_failure_bit_equal"] = l3[:3] == ref_lsm[:3]
    out["lsm_auto_failure_seconds"] = l3[3]
    out["lsm_auto_failure_orphans"] = c1["orphans"] - c0["orphans"]
    out["lsm_auto_failure_mode_is_host"] = mg.reduce_mode == _ffi.HH_MGPU_REDUCE_HOST
    mg.close()
    mg = _ffi.MultiGpu(DEV, _ffi.HH_MGPU_RCCL)
    stub.stub_rccl_fail_at(1, 3)
    t0 = time.perf_counter()
    try:
        lsm(mg, LSM_CASE)
        out["lsm_strict_failure_code"] = 0
    except _ffi.HedgehogMCError as e:
        out["lsm_strict_failure_code"] = e.code
    out["lsm_strict_failure_seconds"] = time.perf_counter() - t0
    mg.close()

    # 4b. a shard that cannot be moved off the stream the failed collective sits on (here: a stream lent by the
    #     caller): the call returns a status once the caller's buffers are free — it does not finish on that
    #     stream — and the context refuses further work; destroying it must not wait for the orphan either
    import torch
    mg = _ffi.MultiGpu(DEV)
    lent = torch.cuda.Stream()
    mg.ctx(1).set_stream(lent.cuda_stream)
    stub.stub_rccl_fail_at(2, 1)
    t0 = time.perf_counter()
    try:
        european(mg)
        out["stuck_code"] = 0
    except _ffi.HedgehogMCError as e:
        out["stuck_code"] = e.code
    try:
        european(mg)
        out["stuck_second_code"] = 0
    except _ffi.HedgehogMCError as e:
        out["stuck_second_code"] = e.code
        out["stuck_text"] = str(e)
    mg.close()
    out["stuck_seconds"] = time.perf_counter() - t0

    # 5. eight ranks, as on the node the driver benches (seven worker threads + the caller's)
    host8 = _ffi.MultiGpu([0] * 8, _ffi.HH_MGPU_HOST_SUM)
    mg8 = _ffi.MultiGpu([0] * 8)
    out["eight_ranks_mode_is_rccl"] = mg8.reduce_mode == _ffi.HH_MGPU_REDUCE_RCCL
    ok8 = True
    for n8 in (7, 8, 2049, 100_003):  # fewer trajectories than ranks (idle shards), one each, ragged
        ok8 &= european(mg8, n=n8)[:2] == european(host8, n=n8)[:2]
    out["eight_ranks_bit_equal"] = bool(ok8)
    out["eight_ranks_lsm_bit_equal"] = lsm(mg8, LSM_CASE)[:3] == lsm(host8, LSM_CASE)[:3]
    stub.stub_rccl_fail_at(5, 1)
    eu8 = european(mg8, n=100_003)
    out["eight_ranks_failure_bit_equal"] = eu8[:2] == european(host8, n=100_003)[:2]
    out["eight_ranks_failure_seconds"] = eu8[2]
    mg8.close()
    host8.close()

    # 6. the basket through a failing collective
    mg = _ffi.MultiGpu(DEV)
    stub.stub_rccl_fail_at(1, 1)
    out["basket_auto_failure_bit_equal"] = basket(mg) == ref_bk
    mg.close()
    host.close()
    out["HH_ERR_RCCL"] = _ffi.HH_ERR_RCCL
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
