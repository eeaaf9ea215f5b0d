"""Per-date exact Heston paths (HestonNoise, heston.jl:82-91, driven by the
NoiseProblem of montecarlo.jl:209-231) on the HIP path: every transition against the numpy/scipy
oracle on identical draws, the laws of the grid rows against closed forms, and LSM on those paths
against the oracle's backward induction and the European Fourier price."""
import ctypes as C
import math

import numpy as np
import pytest

import hedgehog_jl_amd as hh
from hedgehog_jl_amd import _ffi
from oracle import analytic, bk_oracle, lsm_oracle
from tests import oracle_ffi as o

pytestmark = pytest.mark.gpu

H252 = dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1.0)
Q2 = dict(S0=100.0, V0=1.5, kappa=0.04, theta=0.3, sigma=-0.6, rho=0.04, r=0.05, T=364 / 365)
NU_ONE = dict(S0=100.0, V0=0.06, kappa=1.0, theta=0.09, sigma=0.3, rho=-0.4, r=0.02, T=1.5)


def gpu_grid(ctx, prm, seeds, steps, want_var=True, **bk):
    n = len(seeds)
    m = o.make_model(**prm, strike=100.0, cp=1.0)
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, steps, seeds=seeds)
    for k, v in bk.items():
        setattr(c, k, v)
    spot = np.zeros((steps + 1, n))
    var = np.zeros((steps + 1, n)) if want_var else None
    res = _ffi.hh_result()
    ctx.check(ctx.lib.hh_heston_exact_grid(ctx.handle, C.byref(m), C.byref(c), spot.ctypes.data,
                                           var.ctypes.data if want_var else None, 0, C.byref(res)))
    return spot, var, res


@pytest.mark.parametrize("name,prm,steps", [("h252", H252, 6), ("q2", Q2, 4), ("nu_one", NU_ONE, 5),
                                            ("h252_monthly", H252, 12)])
def test_every_transition_matches_oracle(hhlib, name, prm, steps):
    """Each (date, trajectory) transition restarted in the oracle from the state the device held:
    no error amplification along the chain.  V' is an exact NCχ² draw (1e-12); S' goes through the
    CDF inversion whose stopping decisions can flip on rounding (tests/test_gpu_bk.py header)."""
    n = 70
    seeds = np.random.default_rng(len(name)).integers(1, 2**63, n).astype(np.uint64)
    spot, var, res = gpu_grid(hhlib, prm, seeds, steps)
    assert np.all(np.isfinite(spot)) and np.all(spot > 0) and np.all(var > 0)
    np.testing.assert_array_equal(spot[0], prm["S0"])
    np.testing.assert_array_equal(var[0], prm["V0"])
    dt = prm["T"] / steps
    stats = {"newton_fail": 0, "bisect": 0, "maxguess": 0}
    rel_s, rel_v = [], []
    for k in range(steps):
        for i in range(n):
            dist = bk_oracle.LogHestonDistribution(spot[k, i], var[k, i], prm["kappa"], prm["theta"],
                                                   prm["sigma"], prm["rho"], prm["r"], dt)
            logS, VT, _ = bk_oracle.rand_path(dist, int(seeds[i]), k, stats)
            rel_s.append(abs(spot[k + 1, i] - math.exp(logS)) / math.exp(logS))
            rel_v.append(abs(var[k + 1, i] - VT) / VT)
    rel_s, rel_v = np.array(rel_s), np.array(rel_v)
    assert rel_v.max() < 1e-11
    assert np.mean(rel_s > 1e-6) <= 0.02, (np.sort(rel_s)[-8:], np.median(rel_s))
    assert abs(int(res.bk_newton_fail) - stats["newton_fail"]) <= max(2, 0.02 * n * steps)
    assert res.n_paths_done == n


def test_chain_matches_oracle_chain(hhlib):
    """The whole chain against oracle/bk_oracle.exact_grid (the oracle carries its OWN states)."""
    n, steps = 48, 5
    seeds = np.arange(11, 11 + n, dtype=np.uint64)
    spot, var, _ = gpu_grid(hhlib, H252, seeds, steps)
    rs, rv = bk_oracle.exact_grid(**H252, n_steps=steps, seeds=seeds)
    np.testing.assert_allclose(var, rv, rtol=1e-9)
    rel = np.abs(spot - rs) / rs
    assert np.mean(rel > 1e-5) <= 0.03, np.sort(rel.ravel())[-8:]


@pytest.mark.parametrize("name,prm,steps,n,bk", [
    ("h252", H252, 12, 3001, {}), ("q2", Q2, 5, 700, {}), ("nu_one", NU_ONE, 7, 513, {}),
    ("h252_tight_tol", H252, 4, 300, {"bk_cf_tol": 1e-6}),   # series beyond the cache: the fall-back kernel
    ("one_date", H252, 1, 400, {})])
def test_batched_dates_and_a_chain_per_date_agree_bit_for_bit(name, prm, steps, n, bk):
    """HH_OPT_GRID_FORM: all (date, trajectory) pairs in one kernel chain vs one chain per date — same
    draws, same arithmetic per pair, so the same bits in every row; the counters are sums over pairs."""
    seeds = np.random.default_rng(steps).integers(1, 2**63, n).astype(np.uint64)
    out = []
    for form in (_ffi.HH_GRID_FORM_PER_DATE, _ffi.HH_GRID_FORM_BATCHED):
        ctx = hh.Context(0)
        ctx.check(ctx.lib.hh_ctx_set_option(ctx.handle, _ffi.HH_OPT_GRID_FORM, form))
        if "bk_cf_tol" in bk:
            ctx.check(ctx.lib.hh_ctx_set_option(ctx.handle, _ffi.HH_OPT_BK_TERM_CACHE, 16))
        out.append(gpu_grid(ctx, prm, seeds, steps, **bk))
    (s0, v0, r0), (s1, v1, r1) = out
    np.testing.assert_array_equal(s0, s1)
    np.testing.assert_array_equal(v0, v1)
    for f in ("bk_newton_fail", "bk_bisect_fallback", "bk_maxguess_fallback", "bk_cf_terms"):
        assert getattr(r0, f) == getattr(r1, f), f
    assert r0.bk_cf_terms > 0 and np.all(np.isfinite(s1)) and np.all(s1 > 0)


def test_batched_grid_splits_into_several_chains_when_the_pairs_pass_the_cache_budget():
    """1024 cached terms per pair: 8 GiB hold 2^20 pairs, so 300 000 trajectories x 7 dates run as
    3 + 3 + 1 dates; rows identical with a smaller ensemble's (trajectories depend on their seed only)."""
    n, steps = 300_000, 7
    seeds = np.arange(1, n + 1, dtype=np.uint64)
    ctx = hh.Context(0)
    ctx.check(ctx.lib.hh_ctx_set_option(ctx.handle, _ffi.HH_OPT_BK_TERM_CACHE, 1024))
    big, bv, res = gpu_grid(ctx, H252, seeds, steps)
    small, sv, _ = gpu_grid(hh.Context(0), H252, seeds[:2000], steps)
    np.testing.assert_array_equal(big[:, :2000], small)
    np.testing.assert_array_equal(bv[:, :2000], sv)
    assert np.all(np.isfinite(big)) and res.bk_cf_terms > 0


def test_trajectories_depend_on_their_seed_only(hhlib):
    """montecarlo.jl:331: one seed per trajectory — position in the ensemble, ensemble size and
    tile placement are invisible."""
    seeds = np.arange(1, 701, dtype=np.uint64)
    full, fv, _ = gpu_grid(hhlib, H252, seeds, 3)
    part, pv, _ = gpu_grid(hhlib, H252, seeds[300:555], 3)
    np.testing.assert_array_equal(full[:, 300:555], part)
    np.testing.assert_array_equal(fv[:, 300:555], pv)
    rev, _, _ = gpu_grid(hhlib, H252, seeds[::-1].copy(), 3, want_var=False)
    np.testing.assert_array_equal(rev[:, ::-1], full)


def test_one_step_grid_is_the_terminal_law_in_distribution(hhlib):
    """n_steps = 1 is ONE transition over [0, T]: same law as the one-shot HestonBroadieKaya solve
    (different draws: per-trajectory seeds vs the single stream), so both match Carr–Madan."""
    n = 200_000
    spot, var, res = gpu_grid(hhlib, H252, np.arange(1, n + 1, dtype=np.uint64), 1)
    ST = spot[1]
    D = math.exp(-H252["r"] * H252["T"])
    pay = D * np.maximum(ST - 100.0, 0.0)
    cm = analytic.carr_madan_heston(100.0, 100.0, H252["r"], H252["V0"], H252["kappa"], H252["theta"],
                                    H252["sigma"], H252["rho"], H252["T"], bound=400.0)
    assert abs(pay.mean() - cm) < 4 * pay.std() / math.sqrt(n) + 2e-3 * cm


@pytest.mark.parametrize("prm,steps,n", [(H252, 12, 200_000), (H252, 50, 100_000), (NU_ONE, 6, 100_000)])
def test_grid_rows_have_the_exact_marginal_laws(hhlib, prm, steps, n):
    """At every date: E[S_t] = S0 e^{rt} (martingale), E[V_t] = θ + (V0−θ)e^{−κt},
    Var[V_t] in closed form (CIR); at expiry the call price matches Carr–Madan — exactness in the
    time step (an Euler scheme on 12 dates would miss these by far more than the MC error)."""
    spot, var, res = gpu_grid(hhlib, prm, np.arange(7, 7 + n, dtype=np.uint64), steps)
    S0, V0, k, th, sg, r, T = (prm[x] for x in ("S0", "V0", "kappa", "theta", "sigma", "r", "T"))
    for j in range(1, steps + 1):
        t = j * T / steps
        se = spot[j].std() / math.sqrt(n)
        assert abs(spot[j].mean() - S0 * math.exp(r * t)) < 4.5 * se + 1e-3 * S0, (j, "S")
        ev = th + (V0 - th) * math.exp(-k * t)
        vv = V0 * sg**2 / k * (math.exp(-k * t) - math.exp(-2 * k * t)) + \
            th * sg**2 / (2 * k) * (1 - math.exp(-k * t))**2
        assert abs(var[j].mean() - ev) < 4.5 * var[j].std() / math.sqrt(n) + 1e-9, (j, "EV")
        assert var[j].var() == pytest.approx(vv, rel=0.05), (j, "VarV")
    D = math.exp(-r * T)
    for K, cp in ((100.0, 1.0), (90.0, -1.0)):
        pay = D * np.maximum(cp * (spot[-1] - K), 0.0)
        cm = analytic.carr_madan_heston(S0, K, r, V0, k, th, sg, prm["rho"], T, bound=400.0)
        if cp < 0:
            cm = cm - S0 + K * D  # put-call parity
        assert abs(pay.mean() - cm) < 4.5 * pay.std() / math.sqrt(n) + 3e-3 * max(cm, 1.0), (K, cp)
    assert res.bk_maxguess_fallback < 0.01 * n * steps
    print(f"exact Heston grid {n} x {steps}: {res.kernel_ms:.2f} ms "
          f"({n * steps / res.kernel_ms / 1e3:.1f} M transitions/s), "
          f"{res.bk_cf_terms / (n * steps):.1f} CF terms per transition")


def gpu_lsm_heston(ctx, prm, K, cp, seeds, steps, degree):
    n = len(seeds)
    m = o.make_model(**prm, strike=K, cp=cp)
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, steps, seeds=seeds)
    tau, val = np.zeros(n, dtype=np.int32), np.zeros(n)
    grid = np.zeros((steps + 1, n))
    res = _ffi.hh_lsm_result()
    D = math.exp(-prm["r"] * prm["T"] / steps)
    ctx.check(ctx.lib.hh_lsm_solve(ctx.handle, C.byref(m), C.byref(c), degree, D, C.byref(res),
                                   tau.ctypes.data, val.ctypes.data, grid.ctypes.data))
    return res, tau, val, grid, D


@pytest.mark.parametrize("cp,K,degree,n,steps", [(-1.0, 100.0, 4, 4000, 10), (-1.0, 110.0, 3, 1500, 6),
                                                 (1.0, 95.0, 2, 900, 3)])
def test_lsm_on_exact_heston_paths_matches_oracle(hhlib, cp, K, degree, n, steps):
    """Same paths as hh_heston_exact_grid, and on them the same backward induction as
    oracle/lsm_oracle.py (least_squares_montecarlo.jl:107-134)."""
    seeds = np.random.default_rng(n).integers(1, 2**63, n).astype(np.uint64)
    res, tau, val, grid, D = gpu_lsm_heston(hhlib, H252, K, cp, seeds, steps, degree)
    spot, _, _ = gpu_grid(hhlib, H252, seeds, steps, want_var=False)
    np.testing.assert_array_equal(grid, spot)
    ref = lsm_oracle.lsm_solve(grid, K, cp, D, degree)
    same = tau == ref["stop_time"]
    assert same.mean() >= 0.998
    np.testing.assert_allclose(val[same], ref["stop_value"][same], rtol=1e-12)
    assert res.price == pytest.approx(ref["price"], rel=2e-4 if not same.all() else 1e-11)
    assert res.rows_regressed == ref["steps_regressed"]


def test_lsm_solve_grid_on_a_caller_grid(hhlib):
    """hh_lsm_solve_grid on a device grid the caller holds == hh_lsm_solve that made the same grid."""
    import torch
    n, steps, degree = 5000, 8, 3
    seeds = np.arange(1, n + 1, dtype=np.uint64)
    res, tau, val, grid, D = gpu_lsm_heston(hhlib, H252, 100.0, -1.0, seeds, steps, degree)
    dev = torch.from_numpy(grid).cuda()
    torch.cuda.synchronize()
    m = o.make_model(**H252, strike=100.0, cp=-1.0)
    res2 = _ffi.hh_lsm_result()
    tau2, val2 = np.zeros(n, dtype=np.int32), np.zeros(n)
    hhlib.check(hhlib.lib.hh_lsm_solve_grid(hhlib.handle, C.byref(m), dev.data_ptr(), n, steps, degree,
                                            D, C.byref(res2), tau2.ctypes.data, val2.ctypes.data))
    assert res2.price == res.price and res2.std_error == res.std_error
    np.testing.assert_array_equal(tau2, tau)
    np.testing.assert_array_equal(val2, val)
    rc = hhlib.lib.hh_lsm_solve_grid(hhlib.handle, C.byref(m), None, n, steps, degree, D,
                                     C.byref(res2), None, None)
    assert rc == _ffi.HH_ERR_INVALID


def test_american_heston_through_the_host_mirror():
    """solve(PricingProblem(American …, HestonInputs), LSM(HestonDynamics(), HestonBroadieKaya(), cfg,
    degree)): the American call on a non-dividend asset with r > 0 is the European one
    (Carr–Madan), the American put carries a positive early-exercise premium."""
    ref = hh.Date(2021, 1, 1)
    expiry = hh.Date(2022, 1, 1)
    mkt = hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7)
    n, steps = 100_000, 25
    cfg = hh.SimulationConfig(n, steps=steps, seeds=np.arange(1, n + 1, dtype=np.uint64))
    method = hh.LSM(hh.HestonDynamics(), hh.HestonBroadieKaya(), cfg, 4)
    cm_call = analytic.carr_madan_heston(100.0, 100.0, 0.03, 0.04, 2.0, 0.04, 0.3, -0.7, 1.0, bound=400.0)
    cm_put = cm_call - 100.0 + 100.0 * math.exp(-0.03)
    call = hh.solve(hh.PricingProblem(hh.VanillaOption(100.0, expiry, hh.American(), hh.Call(), hh.Spot()), mkt), method)
    assert abs(call.price - cm_call) < 4 * call.std_error + 0.01 * cm_call
    put = hh.solve(hh.PricingProblem(hh.VanillaOption(100.0, expiry, hh.American(), hh.Put(), hh.Spot()), mkt),
                   method, spot_paths=True)
    assert cm_put + 0.05 < put.price < cm_put + 1.0
    assert put.spot_paths.shape == (steps + 1, n) and put.stopping_info[0].max() == steps
    paths = hh.simulate_heston_exact_paths(
        hh.PricingProblem(hh.VanillaOption(100.0, expiry, hh.European(), hh.Put(), hh.Spot()), mkt),
        method.mc_method)
    np.testing.assert_array_equal(paths.spot, put.spot_paths)
    assert paths.variance.shape == (steps + 1, n) and paths.times[-1] == 1.0
    np.testing.assert_allclose(paths.log_spot[0], math.log(100.0))
    with pytest.raises(hh.MethodError):  # Euler states are log-prices: not an LSM path source
        hh.solve(hh.PricingProblem(hh.VanillaOption(100.0, expiry, hh.American(), hh.Put(), hh.Spot()), mkt),
                 hh.LSM(hh.HestonDynamics(), hh.EulerMaruyama(), cfg, 4))


def test_grid_argument_errors(hhlib):
    m = o.make_model(**H252, strike=100.0, cp=1.0)
    seeds = np.arange(1, 11, dtype=np.uint64)
    spot = np.zeros((3, 10))

    def call(c, mm=m):
        return hhlib.lib.hh_heston_exact_grid(hhlib.handle, C.byref(mm), C.byref(c), spot.ctypes.data,
                                              None, 0, None)

    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, 10, 2, seeds=seeds)
    assert call(c) == _ffi.HH_ERR_UNSUPPORTED
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, 10, 2, seeds=seeds, antithetic=1)
    assert call(c) == _ffi.HH_ERR_UNSUPPORTED
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, 10, 0, seeds=seeds)
    assert call(c) == _ffi.HH_ERR_INVALID
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, 10, 2, seeds=seeds[:4])
    assert call(c) == _ffi.HH_ERR_INVALID  # one seed per trajectory (montecarlo.jl:65-66)
    assert b"seeds" in hhlib.lib.hh_last_error(hhlib.handle)
    bad = o.make_model(**{**H252, "sigma": 0.0}, strike=100.0, cp=1.0)
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, 10, 2, seeds=seeds)
    assert call(c, bad) == _ffi.HH_ERR_INVALID
    assert call(c) == _ffi.HH_OK


try:
    from hypothesis import HealthCheck, Phase, given, settings
    from hypothesis import strategies as st
except ImportError:  # pragma: no cover
    given = None

if given is not None:
    @settings(max_examples=30, deadline=None, derandomize=True, database=None,
              phases=[Phase.explicit, Phase.generate],
              suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
    @given(kappa=st.floats(0.1, 5.0), theta=st.floats(0.005, 0.3), sigma=st.floats(0.05, 1.5),
           rho=st.sampled_from([-0.9, -0.3, 0.0, 0.6]), V0=st.floats(0.001, 0.5), T=st.floats(0.1, 3.0),
           steps=st.integers(1, 24), seed=st.integers(1, 2**31))
    def test_exact_grid_random_models_stay_finite(hhlib, kappa, theta, sigma, rho, V0, T, steps, seed):
        """Chains of transitions over random models, including the ones where the variance is absorbed at
        zero (d = 4κθ/σ² ≪ 1: rows of the variance grid sit at the 2^-1000 floor and the next transition
        starts there) and large Bessel orders / arguments (short steps, small vol-of-vol): every spot finite
        and positive, every variance positive, the martingale E[S_t] = S0 e^{rt} within 5 standard errors
        at the last date."""
        prm = dict(S0=100.0, V0=V0, kappa=kappa, theta=theta, sigma=sigma, rho=rho, r=0.02, T=T)
        n = 1500
        seeds = np.random.default_rng(seed).integers(1, 2**63, n).astype(np.uint64)
        spot, var, res = gpu_grid(hhlib, prm, seeds, steps)
        assert np.all(np.isfinite(spot)) and np.all(spot > 0), (np.isnan(spot).sum(), (spot <= 0).sum())
        assert np.all(np.isfinite(var)) and np.all(var > 0)
        ST = spot[-1]
        se = ST.std(ddof=1) / math.sqrt(n)
        assert abs(ST.mean() - 100.0 * math.exp(0.02 * T)) < 5 * se + 0.05
        # … and the same bits from a kernel chain per date (hhlib's default is the batched form)
        per_date = _per_date_ctx()
        spot1, var1, res1 = gpu_grid(per_date, prm, seeds, steps)
        np.testing.assert_array_equal(spot, spot1)
        np.testing.assert_array_equal(var, var1)
        assert res.bk_cf_terms == res1.bk_cf_terms and res.bk_bisect_fallback == res1.bk_bisect_fallback

    _PER_DATE = []

    def _per_date_ctx():
        if not _PER_DATE:
            ctx = hh.Context(0)
            ctx.check(ctx.lib.hh_ctx_set_option(ctx.handle, _ffi.HH_OPT_GRID_FORM, _ffi.HH_GRID_FORM_PER_DATE))
            _PER_DATE.append(ctx)
        return _PER_DATE[0]


@pytest.mark.parametrize("name,prm,steps,n", [("h252", H252, 12, 100_001), ("q2", Q2, 5, 230_000), ("h252_small", H252, 3, 700),
                                             ("h252_last_workgroup_of_the_sort_half_empty", H252, 3, 350_000)])
def test_sorted_and_natural_order_of_a_chain_give_the_same_grid(name, prm, steps, n):
    """HH_OPT_GRID_ORDER: a batched chain runs its (date, trajectory) pairs sorted by a coarse key of V0·V_T (lanes
    of a wave then share a regime of the Bessel function) or in their natural order — nothing a pair computes
    depends on its neighbours and the counters are whole numbers: the same rows, the same counters, bit for bit.
    (The order is applied from 2^20 pairs on: the first two cases and the last — 1026 runs of the counting sort, two of
    the four waves of its last workgroup without one; the third stays in its natural order either way.)"""
    seeds = np.random.default_rng(steps + 100).integers(1, 2**63, n).astype(np.uint64)
    out = []
    for order in (0, 1):
        ctx = hh.Context(0)
        ctx.check(ctx.lib.hh_ctx_set_option(ctx.handle, _ffi.HH_OPT_GRID_ORDER, order))
        out.append(gpu_grid(ctx, prm, seeds, steps))
        ctx.close()
    (s0, v0, r0), (s1, v1, r1) = out
    np.testing.assert_array_equal(s0, s1)
    np.testing.assert_array_equal(v0, v1)
    for f in ("bk_newton_fail", "bk_bisect_fallback", "bk_maxguess_fallback", "bk_cf_terms"):
        assert getattr(r0, f) == getattr(r1, f), f
