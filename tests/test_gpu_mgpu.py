"""hh_mgpu_*: ONE library call from ONE host thread shards a solve over several devices
(what montecarlo.jl:478-493 stays — one `solve` — when the ensemble is sharded).

On the one-GPU test box the sharding logic runs with device 0 listed TWICE or three times: RCCL
refuses a duplicated device at ncclCommInitAll, so those contexts take the host's ordered sum —
which is exactly the fallback path the library promises when RCCL is absent or fails.  The RCCL
path itself runs with one device (flags = HH_MGPU_RCCL forces the all-reduce even for one rank)
and, on boxes with two or more GPUs, with real ranks.
"""
import ctypes as C

import numpy as np
import pytest

from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o
from tests.test_gpu_parity import HESTON_SEEDS, gpu_solve, seeds_for

pytestmark = pytest.mark.gpu

GBM, HES = _ffi.HH_LOGNORMAL, _ffi.HH_HESTON
EM, EXACT, BK = _ffi.HH_EULER_MARUYAMA, _ffi.HH_EXACT_LAW, _ffi.HH_BROADIE_KAYA
GEN, REP = _ffi.HH_NOISE_GENERATE, _ffi.HH_NOISE_REPLAY


def n_gpus():
    import torch
    return torch.cuda.device_count()


def same_result(a, b, P=0, rel=1e-13):
    assert a.n_paths_done == b.n_paths_done
    for f in ("price", "sum_payoff", "sumsq_payoff", "std_error"):
        assert getattr(a, f) == pytest.approx(getattr(b, f), rel=rel, abs=1e-300), f
    for k in range(P):
        assert a.dprice[k] == pytest.approx(b.dprice[k], rel=rel, abs=1e-300)


def test_one_device_is_the_single_solve_bit_for_bit(hhlib):
    n, steps = 20_000, 40
    seeds = seeds_for(n, 3)
    m = o.make_model(seeds=HESTON_SEEDS, n_partials=3)
    c = o.make_config(HES, EM, n, steps, antithetic=1, seeds=seeds, n_partials=3)
    r1, t1 = gpu_solve(hhlib, m, c)
    mg = _ffi.MultiGpu([0])
    assert mg.n_devices == 1 and mg.reduce_mode == _ffi.HH_MGPU_REDUCE_HOST
    t2 = np.zeros(2 * n)
    r2 = mg.solve(m, c, t2)
    assert (r2.price, r2.sum_payoff, r2.sumsq_payoff, r2.std_error) == \
           (r1.price, r1.sum_payoff, r1.sumsq_payoff, r1.std_error)
    assert [r2.dprice[k] for k in range(3)] == [r1.dprice[k] for k in range(3)]
    np.testing.assert_array_equal(t1, t2)
    mg.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
@pytest.mark.parametrize("n", [1, 2, 257, 10_001])
def test_host_ordered_sum_reproduces_the_single_solve(hhlib, devices, n):
    """Two / three shards (empty ones included: n < devices) on the host-sum path: Σ, Σ², the three
    partials and every terminal sample of the one-device solve."""
    steps = 17
    seeds = seeds_for(n, 5)
    m = o.make_model(seeds=HESTON_SEEDS, n_partials=3)
    c = o.make_config(HES, EM, n, steps, antithetic=1, seeds=seeds, n_partials=3)
    r1, t1 = gpu_solve(hhlib, m, c)
    mg = _ffi.MultiGpu(devices)  # AUTO: RCCL refuses the duplicate device -> host sum
    assert mg.reduce_mode == _ffi.HH_MGPU_REDUCE_HOST
    assert "summed on the host" in mg.last_error()
    t2 = np.zeros(2 * n)
    r2 = mg.solve(m, c, t2)
    same_result(r2, r1, P=3)
    np.testing.assert_array_equal(t1, t2)  # a trajectory does not depend on its shard
    mg.close()


def test_exact_law_shards_by_path_offset(hhlib):
    """Exact laws draw by GLOBAL index from one key (montecarlo.jl:456): the shards continue the
    sample, they do not repeat it."""
    n = 100_003
    seeds = np.array([12345], dtype=np.uint64)
    m = o.make_model(sigma=0.2)
    c = o.make_config(GBM, EXACT, n, 1, antithetic=1, seeds=seeds, path_offset=77)
    r1, t1 = gpu_solve(hhlib, m, c)
    mg = _ffi.MultiGpu([0, 0, 0], _ffi.HH_MGPU_HOST_SUM)
    t2 = np.zeros(2 * n)
    r2 = mg.solve(m, c, t2)
    same_result(r2, r1)
    np.testing.assert_array_equal(t1, t2)
    mg.close()


@pytest.mark.parametrize("layout", [_ffi.HH_REPLAY_TILE_MAJOR, _ffi.HH_REPLAY_PATH_MAJOR])
def test_replay_buffers_are_sliced_by_the_shard_ranges(hhlib, oracle, layout):
    n, steps = 5_000, 9
    seeds = seeds_for(n, 8)
    m = o.make_model()
    dW = oracle.wiener_fill(HES, m.rho, m.T, steps, seeds)  # tile-major
    if layout == _ffi.HH_REPLAY_PATH_MAJOR:
        tiles = dW.reshape(-1, steps, 2, 256)
        dW = np.ascontiguousarray(tiles.transpose(0, 3, 1, 2).reshape(-1, steps, 2)[:n])
    c = o.make_config(HES, EM, n, steps, noise_mode=REP, replay=dW, replay_layout=layout)
    r1, t1 = gpu_solve(hhlib, m, c)
    mg = _ffi.MultiGpu([0, 0, 0], _ffi.HH_MGPU_HOST_SUM)
    for g in range(3):  # tile-major data can only be cut at tile boundaries
        a, b = mg.shard_range(n, g, tile_aligned=layout == _ffi.HH_REPLAY_TILE_MAJOR)
        assert layout == _ffi.HH_REPLAY_PATH_MAJOR or a % 256 == 0
    t2 = np.zeros(n)
    r2 = mg.solve(m, c, t2)
    same_result(r2, r1)
    np.testing.assert_array_equal(t1, t2)
    mg.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_replay_slices_that_do_not_start_on_a_16_byte_boundary(hhlib, devices):
    """One double per trajectory (the exact law; lognormal Euler with an odd row length) cut at an odd
    trajectory: the shard's slice starts 8 bytes off a 16-byte boundary, which the kernels' operand
    rules refuse — the library stages it (the single solve accepts the whole buffer)."""
    n = 10_001  # 2 devices: a = 5001; 3 devices: a = 3334, 6668
    rng = np.random.default_rng(11)
    m = o.make_model(sigma=0.2)
    z = rng.standard_normal(n)
    for layout in (_ffi.HH_REPLAY_TILE_MAJOR, _ffi.HH_REPLAY_PATH_MAJOR):
        c = o.make_config(GBM, EXACT, n, 1, antithetic=1, noise_mode=REP, replay=z, replay_layout=layout)
        r1, t1 = gpu_solve(hhlib, m, c)
        mg = _ffi.MultiGpu(devices, _ffi.HH_MGPU_HOST_SUM)
        t2 = np.zeros(2 * n)
        r2 = mg.solve(m, c, t2)
        same_result(r2, r1)
        np.testing.assert_array_equal(t1, t2)
        mg.close()
    for steps in (1, 7):  # odd rows of the reference's own layout: the repack path
        dW = rng.standard_normal((n, steps)) * np.sqrt(1.0 / steps)
        c = o.make_config(GBM, EM, n, steps, noise_mode=REP, replay=dW.ravel(),
                          replay_layout=_ffi.HH_REPLAY_PATH_MAJOR)
        r1, t1 = gpu_solve(hhlib, m, c)
        mg = _ffi.MultiGpu(devices, _ffi.HH_MGPU_HOST_SUM)
        t2 = np.zeros(n)
        r2 = mg.solve(m, c, t2)
        same_result(r2, r1)
        np.testing.assert_array_equal(t1, t2)
        mg.close()
    with pytest.raises(_ffi.HedgehogMCError, match="16-byte aligned"):  # the WHOLE buffer, as for hh_mc_solve
        buf = np.zeros(n + 1)
        c = o.make_config(GBM, EXACT, n, 1, noise_mode=REP, replay=buf[1:])
        mg = _ffi.MultiGpu(devices, _ffi.HH_MGPU_HOST_SUM)
        try:
            mg.solve(m, c)
        finally:
            mg.close()


@pytest.mark.parametrize("mode", [_ffi.HH_MGPU_ENQUEUE_SERIAL, _ffi.HH_MGPU_ENQUEUE_THREADS])
def test_enqueue_modes_and_their_statistics(hhlib, mode):
    n, steps = 30_000, 25
    m = o.make_model(seeds=HESTON_SEEDS, n_partials=3)
    c = o.make_config(HES, EM, n, steps, seeds=seeds_for(n, 3), n_partials=3)
    r1, t1 = gpu_solve(hhlib, m, c)
    mg = _ffi.MultiGpu([0, 0, 0, 0], _ffi.HH_MGPU_HOST_SUM)
    mg.set_option(_ffi.HH_MGPU_OPT_ENQUEUE, mode)
    for _ in range(3):  # workers found awake, then parked (the sleep), then woken again
        t2 = np.zeros(n)
        r2 = mg.solve(m, c, t2)
        same_result(r2, r1, P=3)
        np.testing.assert_array_equal(t1, t2)
        import time
        time.sleep(0.01)
    per, whole = mg.enqueue_stats()
    assert len(per) == 4 and all(0.0 < p < 1e6 for p in per) and whole >= max(per)
    with pytest.raises(_ffi.HedgehogMCError, match="HH_MGPU_OPT_ENQUEUE"):
        mg.set_option(_ffi.HH_MGPU_OPT_ENQUEUE, 7)
    mg.close()


def test_broadie_kaya_shards_and_its_replay_slices(hhlib):
    n = 3_001
    m = o.make_model()
    c = o.make_config(HES, BK, n, 1, seeds=np.array([99], dtype=np.uint64))
    r1, t1 = gpu_solve(hhlib, m, c)
    mg = _ffi.MultiGpu([0, 0], _ffi.HH_MGPU_HOST_SUM)
    t2 = np.zeros(n)
    r2 = mg.solve(m, c, t2)
    same_result(r2, r1)
    np.testing.assert_array_equal(t1, t2)
    assert (r2.bk_newton_fail, r2.bk_cf_terms) == (r1.bk_newton_fail, r1.bk_cf_terms)
    # the caller's draws [V_T | u | Z]: three slices per shard
    rng = np.random.default_rng(4)
    draws = np.concatenate([0.04 * rng.chisquare(2.0, n) / 2.0, rng.uniform(0.001, 0.999, n),
                            rng.standard_normal(n)])
    c2 = o.make_config(HES, BK, n, 1, noise_mode=REP, replay=draws)
    r3, t3 = gpu_solve(hhlib, m, c2)
    t4 = np.zeros(n)
    r4 = mg.solve(m, c2, t4)
    same_result(r4, r3)
    np.testing.assert_array_equal(t3, t4)
    mg.close()


def test_device_resident_shards(hhlib, oracle):
    """hh_mgpu_solve_shards: one hh_config per device with device-resident increments (what the bench
    times), filled on each device's own context."""
    n, steps = 6_000, 12
    seeds = seeds_for(n, 2)
    m = o.make_model()
    c = o.make_config(HES, EM, n, steps, seeds=seeds)
    r1, _ = gpu_solve(hhlib, m, c, want_terminal=False)
    mg = _ffi.MultiGpu([0, 0], _ffi.HH_MGPU_HOST_SUM)
    cfgs, keep = [], []
    for g in range(2):
        a, b = mg.shard_range(n, g, tile_aligned=True)
        ctx = mg.ctx(g)
        sh = np.ascontiguousarray(seeds[a:b])
        buf = _ffi.DeviceBuffer(ctx, 8 * ctx.lib.hh_replay_elems(b - a, steps, HES))
        ctx.check(ctx.lib.hh_wiener_fill(ctx.handle, HES, m.rho, m.T, steps, b - a, sh.ctypes.data, 0, buf.ptr))
        cg = o.make_config(HES, EM, b - a, steps, noise_mode=REP)
        cg.replay, cg.replay_on_device = buf.ptr, 1
        cfgs.append(cg)
        keep.append((sh, buf))
    r2 = mg.solve_shards(m, cfgs)
    same_result(r2, r1, rel=1e-12)  # REPLAY of the fill == GENERATE up to the sqrt(dt) product's rounding
    mg.close()


def test_basket_over_shards(hhlib):
    n, steps = 9_000, 8
    seeds = seeds_for(n, 6)
    m = o.make_model()
    c = o.make_config(HES, EM, n, steps, seeds=seeds)
    strikes = np.array([80.0, 95.0, 100.0, 105.0, 130.0])
    cps = np.array([1.0, -1.0, 1.0, 1.0, -1.0])
    one = (_ffi.hh_result * 5)()
    hhlib.check(hhlib.lib.hh_mc_solve_basket(hhlib.handle, C.byref(m), C.byref(c), strikes.ctypes.data,
                                             cps.ctypes.data, 5, one, None))
    mg = _ffi.MultiGpu([0, 0, 0], _ffi.HH_MGPU_HOST_SUM)
    many = (_ffi.hh_result * 5)()
    mg.check(mg.lib.hh_mgpu_solve_basket(mg.handle, C.byref(m), C.byref(c), strikes.ctypes.data,
                                         cps.ctypes.data, 5, many))
    for k in range(5):
        same_result(many[k], one[k])
    mg.close()


def test_errors_do_not_leave_work_behind(hhlib):
    mg = _ffi.MultiGpu([0, 0], _ffi.HH_MGPU_HOST_SUM)
    m = o.make_model()
    c = o.make_config(HES, EM, 1000, 5, seeds=seeds_for(1000))
    c.seeds_on_device = 1  # host buffers only
    with pytest.raises(_ffi.HedgehogMCError, match="host buffers"):
        mg.solve(m, c)
    c.seeds_on_device = 0
    c.seeds_len = 10  # fewer seeds than trajectories: the reference's ArgumentError (montecarlo.jl:65-66)
    with pytest.raises(_ffi.HedgehogMCError, match="Number of seeds"):
        mg.solve(m, c)
    bad = o.make_model(S0=-1.0)
    c.seeds_len = 1000
    with pytest.raises(_ffi.HedgehogMCError, match="shard 0"):
        mg.solve(bad, c)
    r = mg.solve(m, c)  # the context is still usable
    assert r.n_paths_done == 1000
    with pytest.raises(_ffi.HedgehogMCError):
        _ffi.MultiGpu([0, 99])
    mg.close()


def test_rccl_all_reduce_on_one_rank(hhlib):
    """flags = HH_MGPU_RCCL: the communicator, the grouped ncclAllReduce and the run-time binding of
    librccl all run, even though one rank has nothing to add."""
    n, steps = 4_000, 6
    m = o.make_model()
    c = o.make_config(HES, EM, n, steps, seeds=seeds_for(n))
    r1, _ = gpu_solve(hhlib, m, c, want_terminal=False)
    mg = _ffi.MultiGpu([0], _ffi.HH_MGPU_RCCL)
    assert mg.reduce_mode == _ffi.HH_MGPU_REDUCE_RCCL
    r2 = mg.solve(m, c)
    assert (r2.price, r2.sum_payoff, r2.sumsq_payoff) == (r1.price, r1.sum_payoff, r1.sumsq_payoff)
    with pytest.raises(_ffi.HedgehogMCError):  # RCCL required, duplicate device: refused at creation
        _ffi.MultiGpu([0, 0], _ffi.HH_MGPU_RCCL)
    mg.close()


@pytest.mark.skipif("n_gpus() < 2", reason="needs two GPUs (the driver's multi-GPU node)")
def test_rccl_all_reduce_over_real_ranks(hhlib):
    g = n_gpus()
    n, steps = 100_000, 32
    m = o.make_model(seeds=HESTON_SEEDS, n_partials=3)
    c = o.make_config(HES, EM, n, steps, seeds=seeds_for(n), n_partials=3)
    r1, t1 = gpu_solve(hhlib, m, c)
    mg = _ffi.MultiGpu(list(range(g)))
    assert mg.reduce_mode == _ffi.HH_MGPU_REDUCE_RCCL, mg.last_error()
    t2 = np.zeros(n)
    r2 = mg.solve(m, c, t2)
    same_result(r2, r1, P=3)
    np.testing.assert_array_equal(t1, t2)
    mh = _ffi.MultiGpu(list(range(g)), _ffi.HH_MGPU_HOST_SUM)
    same_result(mh.solve(m, c), r1, P=3)
    mg.close()
    mh.close()


def test_host_mirror_devices_keyword(hhlib):
    """solve(prob, MonteCarlo(…, devices=…)): the reference's one call, sharded inside the library."""
    import hedgehog_jl_amd as hh
    from datetime import date
    ref = date(2021, 1, 1)
    prob = hh.PricingProblem(hh.VanillaOption(100.0, date(2022, 1, 1), hh.European(), hh.Call(), hh.Spot()),
                             hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7))
    cfg = hh.SimulationConfig(30_000, steps=20, seeds=np.arange(1, 30_001),
                              variance_reduction=hh.Antithetic())
    one = hh.solve(prob, hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(), cfg))
    two = hh.solve(prob, hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(), cfg, devices=(0, 0)))
    assert two.price == pytest.approx(one.price, rel=1e-13)
    np.testing.assert_array_equal(one.ensemble[0], two.ensemble[0])
    np.testing.assert_array_equal(one.ensemble[1], two.ensemble[1])


def test_host_mirror_devices_keyword_reaches_baskets_and_batched_greeks(hhlib):
    """solve(::BasketPricingProblem, MonteCarlo(…, devices=…)) and the fused BatchGreekProblem pass through the
    same sharding: hh_mgpu_solve_basket / hh_mgpu_solve with dual seeds."""
    import dataclasses
    from datetime import date

    import hedgehog_jl_amd as hh
    ref, exp1, exp2 = date(2021, 1, 1), date(2022, 1, 1), date(2021, 7, 1)
    mkt = hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7)
    cfg = hh.SimulationConfig(20_000, steps=16, seeds=np.arange(1, 20_001))
    mc = hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(), cfg)
    mc3 = dataclasses.replace(mc, devices=(0, 0, 0))
    payoffs = [hh.VanillaOption(k, e, hh.European(), cp, hh.Spot())
               for e in (exp1, exp2) for k, cp in ((90.0, hh.Put()), (100.0, hh.Call()), (115.0, hh.Call()))]
    one = hh.solve(hh.BasketPricingProblem(payoffs, mkt), mc)
    many = hh.solve(hh.BasketPricingProblem(payoffs, mkt), mc3)
    for a, b in zip(one.solutions, many.solutions):
        assert b.price == pytest.approx(a.price, rel=1e-13)
        assert b.std_error == pytest.approx(a.std_error, rel=1e-12)
    prob = hh.PricingProblem(payoffs[1], mkt)
    lenses = (hh.SpotLens(), hh.optic("market_inputs.V0"))
    g1 = hh.solve(hh.BatchGreekProblem(prob, lenses), hh.ForwardAD(), mc)
    g3 = hh.solve(hh.BatchGreekProblem(prob, lenses), hh.ForwardAD(), mc3)
    for ln in lenses:
        assert g3[ln] == pytest.approx(g1[ln], rel=1e-12)


@pytest.mark.parametrize("case", [dict(model="gbm", n=6000, steps=20, degree=4, anti=1, cp=-1.0, strike=100.0, seed=7),
                                  dict(model="gbm", n=3001, steps=7, degree=2, anti=0, cp=-1.0, strike=110.0, seed=8),
                                  dict(model="heston", n=2500, steps=6, degree=3, anti=0, cp=-1.0, strike=100.0, seed=9)],
                         ids=lambda c: f"{c['model']}-{c['n']}x{c['steps']}")
@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0)])
def test_lsm_sharded_inside_the_library(hhlib, case, devices):
    """hh_mgpu_lsm_solve: the phased induction on every shard with the exchanges INSIDE the library (here
    the host ordered sum: device 0 listed several times).  One device: the fused solve bit for bit;
    several: the same stopping decisions except where a payoff equals its fitted continuation to
    rounding, as for the two-process form (tests/test_gpu_sharded.py)."""
    import dataclasses

    import hedgehog_jl_amd as hh
    from tests.shard_worker import problem
    prob, method = problem(case)
    single = hh.solve(prob, method)
    mc = dataclasses.replace(method.mc_method, devices=devices)
    many = hh.solve(prob, hh.LSM(mc, method.degree))
    n_tot = case["n"] * (2 if case["anti"] else 1)
    assert many.result.n_paths_total == n_tot
    tau1, val1 = single.stopping_info
    tau2, val2 = many.stopping_info
    same = tau1 == tau2
    if len(devices) == 1:
        assert many.price == single.price and same.all()
        np.testing.assert_array_equal(val1, val2)
        return
    assert same.mean() >= 0.998
    np.testing.assert_allclose(val2[same], val1[same], rtol=1e-12)
    assert many.price == pytest.approx(single.price, rel=2e-4 if not same.all() else 1e-11)
    assert many.std_error == pytest.approx(single.std_error, rel=1e-3)
