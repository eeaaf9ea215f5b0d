"""Device Carr–Madan against the scipy restatement (oracle/analytic.py) and the
values the reference's MC tests use as their targets."""
import numpy as np
import pytest

import hedgehog_jl_amd as hh
from oracle import analytic

pytestmark = pytest.mark.gpu


def heston_prob(params, r, ref, expiry, K=100.0, cp=None):
    payoff = hh.VanillaOption(K, expiry, hh.European(), cp or hh.Call(), hh.Spot())
    return hh.PricingProblem(payoff, hh.HestonInputs(ref, r, 100.0, *params))


@pytest.mark.parametrize("params,r,ref,expiry,bound,target", [
    ((0.04, 2.0, 0.04, 0.3, -0.7), 0.03, hh.Date(2020, 1, 1), hh.Date(2021, 1, 1), 32.0,
     9.257069529912402),                                   # montecarlo_heston.jl:13-49
    ((0.04, 2.0, 0.04, 0.3, -0.7), 0.03, hh.Date(2021, 1, 1), hh.Date(2022, 1, 1), 400.0,
     9.242521073959068),                                   # H252
    ((1.5, 0.04, 0.3, -0.6, 0.04), 0.05, hh.Date(2025, 1, 1), hh.Date(2025, 12, 31), 32.0,
     46.31412390823014),                                   # montecarlo_heston.jl:161-170 as run (Q2)
])
def test_carr_madan_heston_targets(params, r, ref, expiry, bound, target):
    prob = heston_prob(params, r, ref, expiry)
    price = hh.solve(prob, hh.CarrMadan(1.0, bound, hh.HestonDynamics())).price
    assert price == pytest.approx(target, rel=1e-9)
    T = hh.yearfrac(ref, expiry)
    V0, k, th, s, rho = params
    assert price == pytest.approx(
        analytic.carr_madan_heston(100.0, 100.0, r, V0, k, th, s, rho, T, bound=bound), rel=1e-9)
    put = hh.solve(heston_prob(params, r, ref, expiry, cp=hh.Put()),
                   hh.CarrMadan(1.0, bound, hh.HestonDynamics())).price
    assert put == pytest.approx(price - 100.0 + 100.0 * np.exp(-r * T), rel=1e-12)


def test_carr_madan_lognormal_equals_black_scholes():
    """test/agreement/price_agreement.jl: Carr–Madan with the lognormal law vs BlackScholesAnalytic,
    atol 1e-6 (365-day expiry, so the √α quirk of montecarlo.jl:302 is invisible)."""
    ref = hh.Date(2021, 1, 1)
    for K in (80.0, 100.0, 125.0):
        for cp in (hh.Call(), hh.Put()):
            payoff = hh.VanillaOption(K, hh.Date(2022, 1, 1), hh.European(), cp, hh.Spot())
            prob = hh.PricingProblem(payoff, hh.BlackScholesInputs(ref, 0.05, 100.0, 0.2))
            cm = hh.solve(prob, hh.CarrMadan(1.0, 32.0, hh.LognormalDynamics())).price
            bs = hh.solve(prob, hh.BlackScholesAnalytic()).price
            assert cm == pytest.approx(bs, abs=1e-6)
            assert bs == pytest.approx(analytic.bs_price(100.0, K, 0.05, 0.2, 1.0, cp()), rel=1e-13)


def test_monte_carlo_against_the_device_fourier_price():
    """The reference's Heston agreement test, entirely on the GPU box: Euler MC vs device Carr–Madan."""
    prob = heston_prob((0.04, 2.0, 0.04, 0.3, -0.7), 0.03, hh.Date(2020, 1, 1), hh.Date(2021, 1, 1))
    cm = hh.solve(prob, hh.CarrMadan(1.0, 32.0, hh.HestonDynamics())).price
    n = 400_000
    mc = hh.solve(prob, hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(),
                                      hh.SimulationConfig(n, steps=200, seeds=np.arange(1, n + 1),
                                                          variance_reduction=hh.Antithetic())),
                  ensemble=False)
    assert abs(mc.price - cm) < 4 * mc.std_error + 0.02


try:
    from hypothesis import HealthCheck, Phase, given, settings
    from hypothesis import strategies as st
except ImportError:  # pragma: no cover
    given = None

if given is not None:
    @settings(max_examples=60, deadline=None, derandomize=True, database=None,
              phases=[Phase.explicit, Phase.generate], suppress_health_check=[HealthCheck.too_slow])
    @given(V0=st.floats(0.005, 0.5), kappa=st.floats(0.1, 5.0), theta=st.floats(0.005, 0.3),
           sigma=st.floats(0.05, 1.2), rho=st.floats(-0.95, 0.95), r=st.floats(-0.01, 0.1),
           days=st.integers(20, 1500), moneyness=st.floats(0.6, 1.6), alpha=st.sampled_from([0.75, 1.0, 1.5]),
           bound=st.sampled_from([32.0, 100.0, 400.0]), put=st.booleans())
    def test_carr_madan_random_heston(V0, kappa, theta, sigma, rho, r, days, moneyness, alpha, bound, put):
        """Random Heston parameters, damping and integration bounds: the device quadrature against the
        scipy restatement of carr_madan.jl:47-92 (same integrand, adaptive quadrature).  Absolute bar
        1e-7 of the spot: the reference itself compares the method to others with atol 1e-6 … 1e-2."""
        import datetime as dt
        ref = hh.Date(2021, 1, 1)
        expiry = ref + dt.timedelta(days=days)
        K = 100.0 * moneyness
        prob = heston_prob((V0, kappa, theta, sigma, rho), r, ref, expiry, K=K, cp=hh.Put() if put else hh.Call())
        price = hh.solve(prob, hh.CarrMadan(alpha, bound, hh.HestonDynamics())).price
        T = hh.yearfrac(ref, expiry)
        want = analytic.carr_madan_heston(100.0, K, r, V0, kappa, theta, sigma, rho, T, alpha=alpha, bound=bound)
        if put:
            want = want - 100.0 + K * np.exp(-r * T)
        assert np.isfinite(price)
        assert price == pytest.approx(want, abs=1e-5)


def test_carr_madan_basket_equals_single_solves():
    """solve(::BasketPricingProblem, ::CarrMadan): every Fourier integral of the basket in one launch
    (the calibration objective's inner loop, calibration.jl:75-88) equals the per-payoff solves bit for
    bit — same integrand, same panels, same reduction — over strikes, expiries, calls and puts."""
    ref = hh.Date(2021, 1, 1)
    mkt = hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7)
    payoffs = [hh.VanillaOption(K, e, hh.European(), cp, hh.Spot())
               for e in (hh.Date(2021, 4, 1), hh.Date(2022, 1, 1), hh.Date(2024, 1, 1))
               for K in (70.0, 95.0, 100.0, 130.0) for cp in (hh.Call(), hh.Put())]
    method = hh.CarrMadan(1.0, 32.0, hh.HestonDynamics())
    sol = hh.solve(hh.BasketPricingProblem(payoffs, mkt), method)
    assert isinstance(sol, hh.BasketPricingSolution) and len(sol.solutions) == len(payoffs)
    for p, s in zip(payoffs, sol.solutions):
        assert s.price == hh.solve(hh.PricingProblem(p, mkt), method).price
        assert s.problem.payoff is p
    # the lognormal law, with the reference's √T quirk on and off (366-day expiry: they differ)
    bs = hh.BlackScholesInputs(ref, 0.05, 100.0, 0.2)
    pays = [hh.VanillaOption(K, hh.Date(2022, 1, 2), hh.European(), hh.Call(), hh.Spot()) for K in (90.0, 110.0)]
    for compat in (False, True):
        meth = hh.CarrMadan(1.0, 32.0, hh.LognormalDynamics(), compat_sqrt_alpha=compat)
        got = [s.price for s in hh.solve(hh.BasketPricingProblem(pays, bs), meth).solutions]
        assert got == [hh.solve(hh.PricingProblem(p, bs), meth).price for p in pays]
