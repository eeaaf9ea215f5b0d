"""solve(::BasketPricingProblem, ::MonteCarlo): one simulation per expiry group, all strikes
reduced together — must equal the reference's definition, independent per-payoff solves
(src/calibration/basket.jl:35-38), here checked against the CPU oracle run payoff by payoff."""
import ctypes as C

import numpy as np
import pytest

import hedgehog_jl_amd as hh
from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dyn,strategy", [(1, 0), (0, 0), (0, 1)])
@pytest.mark.parametrize("anti", [0, 1])
@pytest.mark.parametrize("P", [0, 2])
def test_basket_equals_independent_oracle_solves(hhlib, oracle, dyn, strategy, anti, P):
    n, steps = 5000 + 37, 25
    seeds = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(7919)
    strikes = np.array([80.0, 95.0, 100.0, 100.0, 105.0, 130.0, 400.0])
    cps = np.array([1.0, 1.0, 1.0, -1.0, -1.0, 1.0, 1.0])
    sd = {"S0": [1, 0], "sigma": [0, 1]} if P else {}
    base = dict(sigma=0.25 if dyn == 0 else 0.3, seeds=sd, n_partials=P)
    m = o.make_model(**base)
    c = o.make_config(dyn, strategy, n, steps, antithetic=anti, seeds=seeds, n_partials=P)
    res = (_ffi.hh_result * len(strikes))()
    term = np.zeros(n * (2 if anti else 1))
    hhlib.check(hhlib.lib.hh_mc_solve_basket(hhlib.handle, C.byref(m), C.byref(c),
                                             strikes.ctypes.data, cps.ctypes.data, len(strikes), res,
                                             term.ctypes.data))
    for k, (K, cp) in enumerate(zip(strikes, cps)):
        mk = o.make_model(strike=float(K), cp=float(cp), **base)
        ro, to, _ = oracle.mc_solve(mk, c)
        np.testing.assert_allclose(term, to, rtol=1e-11)
        assert res[k].price == pytest.approx(ro.price, rel=1e-11, abs=1e-13)
        assert res[k].std_error == pytest.approx(ro.std_error, rel=1e-8, abs=1e-13)
        for q in range(P):
            assert res[k].dprice[q] == pytest.approx(ro.dprice[q], rel=1e-9, abs=1e-11)
    assert res[-1].price == 0.0  # far out of the money: empty payoff, still a valid record


def test_basket_host_api_groups_by_expiry():
    ref = hh.Date(2021, 1, 1)
    mkt = hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7)
    e1, e2 = hh.Date(2022, 1, 1), hh.Date(2021, 7, 1)
    payoffs = [hh.VanillaOption(K, e, hh.European(), cp, hh.Spot())
               for K, e, cp in ((90.0, e1, hh.Call()), (100.0, e2, hh.Call()), (100.0, e1, hh.Put()),
                                (110.0, e2, hh.Put()), (100.0, e1, hh.Call()))]
    n = 20_000
    mc = hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(),
                       hh.SimulationConfig(n, steps=40, seeds=np.arange(1, n + 1)))
    basket = hh.BasketPricingProblem(payoffs, mkt)
    sol = hh.solve(basket, mc)
    assert isinstance(sol, hh.BasketPricingSolution) and len(sol.solutions) == 5
    for p, s in zip(payoffs, sol.solutions):
        single = hh.solve(hh.PricingProblem(p, mkt), mc, ensemble=False)
        assert s.price == pytest.approx(single.price, rel=1e-12)
        assert s.problem.payoff is p
    # calibration-style use: a Dual model parameter flows to every payoff (calibration.jl:75-88)
    mkt_d = hh.set(hh.PricingProblem(payoffs[0], mkt), hh.optic("market_inputs.V0"),
                   hh.Dual(0.04, (1.0,))).market_inputs
    sol_d = hh.solve(hh.BasketPricingProblem(payoffs, mkt_d), mc)
    for p, s in zip(payoffs, sol_d.solutions):
        single = hh.solve(hh.PricingProblem(p, mkt_d), mc, ensemble=False)
        assert s.price.partials[0] == pytest.approx(single.price.partials[0], rel=1e-10)


def test_basket_full_size_strike_ladder():
    """10^6 x 252 Heston, 41 strikes in one pass: prices decrease in strike for calls, satisfy
    put-call parity on the shared paths, and cost about one simulation."""
    import torch
    ctx = hh.get_context(0)
    n, steps = 1_000_000, 252
    seeds = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
    strikes = np.linspace(60.0, 140.0, 41)
    allK = np.concatenate([strikes, strikes])
    cps = np.concatenate([np.ones(41), -np.ones(41)])
    m = o.make_model()
    c = o.make_config(1, 0, n, steps)
    c.seeds, c.seeds_on_device = seeds.data_ptr(), 1
    res = (_ffi.hh_result * 82)()
    ctx.check(ctx.lib.hh_mc_solve_basket(ctx.handle, C.byref(m), C.byref(c), allK.ctypes.data,
                                         cps.ctypes.data, 82, res, None))
    calls = np.array([res[k].price for k in range(41)])
    puts = np.array([res[41 + k].price for k in range(41)])
    assert np.all(np.diff(calls) < 0) and np.all(np.diff(puts) > 0)
    D = np.exp(-0.03)
    fwd = (calls - puts) / D + strikes  # = E[S_T] for every strike, same paths
    assert np.ptp(fwd) < 1e-9 * 100 and fwd[0] == pytest.approx(100 * np.exp(0.03), rel=1e-3)
    single = _ffi.hh_result()
    ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(single), None))
    assert res[20].price == pytest.approx(single.price, rel=1e-12)  # K = 100 call
    assert res[0].kernel_ms < 3 * single.kernel_ms + 1.0


def test_basket_on_broadie_kaya_samples(hhlib):
    """A strike ladder on exact Heston samples: one Broadie–Kaya simulation, all strikes reduced on
    it; each equals the single-payoff solve on the same key."""
    n = 20_000
    strikes = np.array([90.0, 100.0, 110.0, 100.0])
    cps = np.array([1.0, 1.0, 1.0, -1.0])
    m = o.make_model()
    c = o.make_config(1, 2, n, seeds=[2718])
    res = (_ffi.hh_result * 4)()
    hhlib.check(hhlib.lib.hh_mc_solve_basket(hhlib.handle, C.byref(m), C.byref(c),
                                             strikes.ctypes.data, cps.ctypes.data, 4, res, None))
    for k in range(4):
        mk = o.make_model(strike=float(strikes[k]), cp=float(cps[k]))
        single = _ffi.hh_result()
        hhlib.check(hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(mk), C.byref(c), C.byref(single),
                                          None))
        assert res[k].price == pytest.approx(single.price, rel=1e-12)
        # the simulation's diagnostics (fall-backs, series terms) come with every payoff
        assert res[k].bk_newton_fail == single.bk_newton_fail > 0
        assert res[k].bk_bisect_fallback == single.bk_bisect_fallback
        assert res[k].bk_maxguess_fallback == single.bk_maxguess_fallback
        assert res[k].bk_cf_terms == single.bk_cf_terms > 10 * n


@pytest.mark.parametrize("n", [1, 4096, 4097, 8 * 4096 + 1, 9 * 4096, 17 * 4096 - 3])
@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 8, 9])
@pytest.mark.parametrize("anti", [0, 1])
def test_basket_grouping_and_padding_leave_each_payoff_bit_identical(hhlib, n, K, anti):
    """The payoff kernel evaluates 4 payoffs per loaded sample and pads its grid to 8 chunks per XCD
    round: a short last group, a lone payoff, chunk counts on both sides of a multiple of 8 and a
    ragged last chunk must all give every payoff the sums of a one-payoff launch, bit for bit."""
    m = o.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, seeds={"S0": [1.0], "sigma": [0.0]}, n_partials=1)
    c = o.make_config(0, 1, n, antithetic=anti, seeds=[12345], n_partials=1)
    strikes = np.linspace(70.0, 140.0, K)
    cps = np.where(np.arange(K) % 2 == 0, 1.0, -1.0)
    res = (_ffi.hh_result * K)()
    hhlib.check(hhlib.lib.hh_mc_solve_basket(hhlib.handle, C.byref(m), C.byref(c), strikes.ctypes.data,
                                             cps.ctypes.data, K, res, None))
    for k in range(K):
        one = (_ffi.hh_result * 1)()
        hhlib.check(hhlib.lib.hh_mc_solve_basket(hhlib.handle, C.byref(m), C.byref(c),
                                                 strikes[k:k + 1].ctypes.data, cps[k:k + 1].ctypes.data, 1,
                                                 one, None))
        assert (res[k].price, res[k].std_error, res[k].dprice[0], res[k].n_paths_done) == \
               (one[0].price, one[0].std_error, one[0].dprice[0], one[0].n_paths_done), (k, K, n)
