"""hh_mc_solve_multi: several models stepped on the SAME draws in one pass — the solves a FiniteDifference or
second-order Greek is made of (greeks_problem.jl:279-303, 318-329, 396-422: 2, 3 or 4 full solves on the seeds
of one SimulationConfig).

Bars: every output of the K-model pass is BIT-IDENTICAL to K independent hh_mc_solve calls (same arithmetic per
model, same summation order) and matches K oracle solves to the tolerances of test_gpu_parity (REPLAY 1e-12,
GENERATE 1e-11)."""
import ctypes as C

import numpy as np
import pytest

import hedgehog_jl_amd as hh
from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o

pytestmark = pytest.mark.gpu

GBM, HES = _ffi.HH_LOGNORMAL, _ffi.HH_HESTON
EM, EXACT, BK = _ffi.HH_EULER_MARUYAMA, _ffi.HH_EXACT_LAW, _ffi.HH_BROADIE_KAYA
GEN, REP = _ffi.HH_NOISE_GENERATE, _ffi.HH_NOISE_REPLAY


def seeds_for(n, salt=0):
    return np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(salt)


def bumped_models(dyn, K, same_noise_law):
    """K models around the test model, each differing in something else; same_noise_law keeps ρ and T (what a
    REPLAY run's increments were drawn for)"""
    base = dict(sigma=0.2) if dyn == GBM else {}
    bumps = [dict(), dict(S0=101.0), dict(sigma=0.21 if dyn == GBM else 0.33, strike=95.0), dict(r=0.035, cp=-1.0),
             dict(V0=0.05, kappa=1.5), dict(theta=0.05, S0=99.0), dict(strike=110.0)]
    if not same_noise_law:
        bumps[2] = dict(rho=-0.5, sigma=0.33)
        bumps[3] = dict(T=1.25, r=0.035, cp=-1.0)
    return [o.make_model(**{**base, **bumps[k % len(bumps)]}) for k in range(K)]


def solve_each(ctx, models, c, want_terminal):
    out = []
    for m in models:
        r = _ffi.hh_result()
        t = np.zeros(c.n_paths * (2 if c.antithetic else 1)) if want_terminal else None
        ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(r), t.ctypes.data if want_terminal else None))
        out.append((r, t))
    return out


def solve_multi(ctx, models, c, want_terminal):
    K = len(models)
    arr = (_ffi.hh_model * K)(*models)
    res = (_ffi.hh_result * K)()
    terms = [np.zeros(c.n_paths * (2 if c.antithetic else 1)) for _ in range(K)] if want_terminal else None
    tp = (C.c_void_p * K)(*[t.ctypes.data for t in terms]) if want_terminal else None
    ctx.check(ctx.lib.hh_mc_solve_multi(ctx.handle, arr, K, C.byref(c), res, tp))
    return [(res[k], terms[k] if want_terminal else None) for k in range(K)]


FIELDS = ("price", "std_error", "sum_payoff", "sumsq_payoff", "n_paths_done")


def same_bits(a, b):
    return all(np.float64(getattr(a, f)).tobytes() == np.float64(getattr(b, f)).tobytes() for f in FIELDS)


@pytest.mark.parametrize("dyn,strat,split", [(HES, EM, 1), (HES, EM, 0), (GBM, EM, 1), (GBM, EXACT, 1)])
@pytest.mark.parametrize("noise", [GEN, REP])
@pytest.mark.parametrize("anti", [0, 1])
@pytest.mark.parametrize("K", [2, 3, 4])
def test_multi_is_k_independent_solves_bit_for_bit(hhlib, oracle, dyn, strat, split, noise, anti, K):
    n_paths, n_steps = 256 * 9 + 77, (1 if strat == EXACT else 23)
    seeds = seeds_for(n_paths, 11)
    models = bumped_models(dyn, K, same_noise_law=(noise == REP))
    rep = None
    if noise == REP:
        rep = oracle.wiener_fill(dyn, models[0].rho, models[0].T, n_steps, seeds) if strat == EM else \
            np.random.default_rng(3).standard_normal(n_paths)
    c = o.make_config(dyn, strat, n_paths, n_steps, antithetic=anti, em_split=split, noise_mode=noise, seeds=seeds, replay=rep)
    each = solve_each(hhlib, models, c, True)
    multi = solve_multi(hhlib, models, c, True)
    for k, ((r1, t1), (rk, tk)) in enumerate(zip(each, multi)):
        assert same_bits(r1, rk), k
        assert t1.tobytes() == tk.tobytes(), k
        ro, to, _ = oracle.mc_solve(models[k], c)
        tol = 1e-12 if noise == REP and strat == EM else 1e-11
        np.testing.assert_allclose(tk, to, rtol=tol)
        assert rk.price == pytest.approx(ro.price, rel=tol, abs=1e-300)
    # models that differ must have given different prices (nothing was computed once and copied)
    assert len({np.float64(r.price).tobytes() for r, _ in multi}) == K


@pytest.mark.parametrize("K", [5, 7, 9, 16])
def test_more_models_than_one_pass_holds(hhlib, K):
    n_paths, n_steps = 1500, 10
    models = bumped_models(HES, K, same_noise_law=False)
    for k, m in enumerate(models):
        m.strike = 90.0 + k  # all different
    c = o.make_config(HES, EM, n_paths, n_steps, seeds=seeds_for(n_paths, 2))
    each = solve_each(hhlib, models, c, False)
    multi = solve_multi(hhlib, models, c, False)
    assert all(same_bits(a[0], b[0]) for a, b in zip(each, multi))


def test_multi_above_the_small_grid_and_with_a_ragged_tile(hhlib, oracle):
    n_paths, n_steps = 256 * 600 + 1, 9
    seeds = seeds_for(n_paths, 5)
    models = bumped_models(HES, 3, same_noise_law=True)
    dW = oracle.wiener_fill(HES, models[0].rho, models[0].T, n_steps, seeds)
    for c in (o.make_config(HES, EM, n_paths, n_steps, noise_mode=REP, replay=dW),
              o.make_config(HES, EM, n_paths, n_steps, seeds=seeds, antithetic=1)):
        assert all(same_bits(a[0], b[0]) for a, b in zip(solve_each(hhlib, models, c, False), solve_multi(hhlib, models, c, False)))


def test_multi_path_major_replay_and_device_buffers(hhlib, oracle):
    ctx = hhlib
    n_paths, n_steps = 256 * 12 + 3, 8
    seeds = seeds_for(n_paths, 4)
    models = bumped_models(HES, 2, same_noise_law=True)
    dW = oracle.wiener_fill(HES, models[0].rho, models[0].T, n_steps, seeds)
    pm = dW.reshape(-1, n_steps, 2, 256).transpose(0, 3, 1, 2).reshape(-1, n_steps, 2)[:n_paths].copy()
    ct = o.make_config(HES, EM, n_paths, n_steps, noise_mode=REP, replay=dW)
    cp = o.make_config(HES, EM, n_paths, n_steps, noise_mode=REP, replay=pm, replay_layout=_ffi.HH_REPLAY_PATH_MAJOR)
    want = solve_multi(ctx, models, ct, True)
    got = solve_multi(ctx, models, cp, True)
    assert all(same_bits(a[0], b[0]) and a[1].tobytes() == b[1].tobytes() for a, b in zip(want, got))
    # device-resident increments and device terminal buffers
    dev = _ffi.DeviceBuffer(ctx, dW.nbytes).upload(dW)
    cd = o.make_config(HES, EM, n_paths, n_steps, noise_mode=REP)
    cd.replay, cd.replay_on_device, cd.replay_len, cd.terminal_on_device = dev.ptr, 1, dW.size, 1
    tb = [_ffi.DeviceBuffer(ctx, 8 * n_paths) for _ in models]
    arr = (_ffi.hh_model * 2)(*models)
    res = (_ffi.hh_result * 2)()
    ctx.check(ctx.lib.hh_mc_solve_multi(ctx.handle, arr, 2, C.byref(cd), res, (C.c_void_p * 2)(*[b.ptr for b in tb])))
    for k in range(2):
        assert same_bits(res[k], want[k][0])
        assert tb[k].download(np.empty(n_paths)).tobytes() == want[k][1].tobytes()


def test_multi_with_either_form_of_the_record_reduction(hhlib):
    n_paths = 256 * 40 + 9
    models = bumped_models(HES, 4, same_noise_law=False)
    c = o.make_config(HES, EM, n_paths, 12, seeds=seeds_for(n_paths, 8), antithetic=1)
    want = solve_multi(hhlib, models, c, False)
    for mode in (0, 1):
        hhlib.set_option(_ffi.HH_OPT_FUSE_REDUCE, mode)
        try:
            got = solve_multi(hhlib, models, c, False)
        finally:
            hhlib.set_option(_ffi.HH_OPT_FUSE_REDUCE, 2)
        assert all(same_bits(a[0], b[0]) for a, b in zip(want, got)), mode


def bk_models():
    """the bumps of finite-difference Greeks on a Broadie–Kaya problem: models 1-4 differ from model 0 in nothing the
    variance process sees (they share its chain), 5 has a chain of its own (σ), 6 shares THAT one, 7 another (V0, κ)"""
    base = {}
    bumps = [dict(), dict(S0=100.1), dict(r=0.031, cp=-1.0), dict(rho=-0.65), dict(strike=101.0, S0=99.9),
             dict(sigma=0.33), dict(sigma=0.33, S0=100.1, cp=-1.0), dict(V0=0.05, kappa=1.5)]
    return [o.make_model(**{**base, **b}) for b in bumps]


@pytest.mark.parametrize("controls, cache, what", [
    (dict(), 0, "the reference's controls: a few dozen trajectories in the ladder"),
    (dict(bk_newton_maxiter=2), 0, "most trajectories in the ladder: many chunks"),
    (dict(bk_cf_tol=1e-6, bk_atol=1e-6, bk_newton_maxiter=20), 32, "series beyond a 32-term cache: the fall-back kernel's records too"),
])
def test_multi_broadie_kaya_shares_the_chain_where_the_variance_process_is_the_same(hhlib, controls, cache, what):
    """hh_mc_solve_multi on HestonBroadieKaya: a bumped spot / rate / ρ / strike is finished from the ∫V the first
    model's chain left (bk_refinish_kernel) — every model's price, sums, counters and samples as hh_mc_solve gives
    them, bit for bit."""
    n_paths = 256 * 37 + 45
    models = bk_models()
    c = o.make_config(HES, BK, n_paths, 1, seeds=seeds_for(n_paths, 1))
    for k, v in controls.items():
        setattr(c, k, v)
    if cache:
        hhlib.set_option(_ffi.HH_OPT_BK_TERM_CACHE, cache)
    try:
        each = solve_each(hhlib, models, c, True)
        multi = solve_multi(hhlib, models, c, True)
        again = solve_each(hhlib, models[::-1], c, False)[::-1]  # and the context is as good as before
    finally:
        if cache:
            hhlib.set_option(_ffi.HH_OPT_BK_TERM_CACHE, 256)
    for k, ((r1, t1), (rk, tk), (r2, _)) in enumerate(zip(each, multi, again)):
        assert same_bits(r1, rk) and same_bits(r1, r2), (what, k)
        assert t1.tobytes() == tk.tobytes(), (what, k)
        for f in ("bk_cf_terms", "bk_newton_fail", "bk_bisect_fallback", "bk_maxguess_fallback"):
            assert getattr(rk, f) == getattr(r1, f), (what, k, f)
    if controls.get("bk_newton_maxiter") == 2:
        assert each[0][0].bk_newton_fail > 0.5 * n_paths
    if cache:
        dec, ln = np.zeros(n_paths, dtype=np.uint32), np.zeros(n_paths, dtype=np.uint32)
        hhlib.check(hhlib.lib.hh_bk_decisions(hhlib.handle, n_paths, dec.ctypes.data, ln.ctypes.data))
        assert (ln > cache).any()


def test_multi_broadie_kaya_on_the_callers_draws_and_device_samples(hhlib):
    import torch
    n_paths = 5000
    rng = np.random.default_rng(3)
    draws = np.ascontiguousarray(np.stack([0.02 + 0.05 * rng.uniform(size=n_paths), rng.uniform(size=n_paths),
                                           rng.standard_normal(n_paths)]))
    models = bk_models()[:4]
    c = o.make_config(HES, BK, n_paths, 1, noise_mode=REP, replay=draws.ravel())
    each = solve_each(hhlib, models, c, True)
    K = len(models)
    dev = [torch.zeros(n_paths, dtype=torch.float64, device="cuda:0") for _ in range(K)]
    c.terminal_on_device = 1
    res = (_ffi.hh_result * K)()
    tp = (C.c_void_p * K)(*[t.data_ptr() for t in dev])
    hhlib.check(hhlib.lib.hh_mc_solve_multi(hhlib.handle, (_ffi.hh_model * K)(*models), K, C.byref(c), res, tp))
    for k in range(K):
        assert same_bits(each[k][0], res[k]) and each[k][1].tobytes() == dev[k].cpu().numpy().tobytes(), k


def test_multi_argument_errors(hhlib):
    ctx = hhlib
    models = bumped_models(HES, 2, same_noise_law=True)
    arr = (_ffi.hh_model * 2)(*models)
    res = (_ffi.hh_result * 17)()
    c = o.make_config(HES, EM, 100, 4, seeds=seeds_for(100))
    assert ctx.lib.hh_mc_solve_multi(ctx.handle, arr, 0, C.byref(c), res, None) == _ffi.HH_ERR_INVALID
    assert ctx.lib.hh_mc_solve_multi(ctx.handle, arr, 17, C.byref(c), res, None) == _ffi.HH_ERR_INVALID
    assert ctx.lib.hh_mc_solve_multi(ctx.handle, None, 2, C.byref(c), res, None) == _ffi.HH_ERR_INVALID
    cp = o.make_config(HES, EM, 100, 4, seeds=seeds_for(100), n_partials=1)
    assert ctx.lib.hh_mc_solve_multi(ctx.handle, arr, 2, C.byref(cp), res, None) == _ffi.HH_ERR_UNSUPPORTED
    models[1].S0 = -1.0
    bad = (_ffi.hh_model * 2)(*models)
    assert ctx.lib.hh_mc_solve_multi(ctx.handle, bad, 2, C.byref(c), res, None) == _ffi.HH_ERR_INVALID
    assert b"S0" in ctx.lib.hh_last_error(ctx.handle)
    # one model: hh_mc_solve itself
    r1 = _ffi.hh_result()
    ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(arr[0]), C.byref(c), C.byref(r1), None))
    ctx.check(ctx.lib.hh_mc_solve_multi(ctx.handle, arr, 1, C.byref(c), res, None))
    assert same_bits(r1, res[0])


# ---- through the host API: the reference's own call forms ---------------------------------------------------

def heston_problem():
    ref = hh.Date(2021, 1, 1)
    mkt = hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7)
    return hh.PricingProblem(hh.VanillaOption(100.0, hh.Date(2022, 1, 1), hh.European(), hh.Call(), hh.Spot()), mkt)


def heston_method(n=20_000, steps=50, anti=False):
    vr = hh.Antithetic() if anti else hh.NoVarianceReduction()
    return hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(),
                         hh.SimulationConfig(n, steps=steps, seeds=seeds_for(n, 21), variance_reduction=vr))


@pytest.mark.parametrize("scheme", [hh.FDCentral(), hh.FDForward(), hh.FDBackward()])
@pytest.mark.parametrize("path", ["market_inputs.spot", "market_inputs.V0", "market_inputs.σ", "market_inputs.ρ", "payoff.strike"])
def test_finite_difference_greek_is_the_references_two_solves(hhlib, scheme, path):
    """solve(GreekProblem, FiniteDifference, MonteCarlo) = compute_fd_derivative's formula on two plain solves
    (greeks_problem.jl:279-303): the shared pass must give that number exactly."""
    prob, m, lens, eps = heston_problem(), heston_method(), hh.optic(path), 1e-3
    x0 = lens(prob)
    price = lambda x: hh.solve(hh.set(prob, lens, x), m, ensemble=False).price
    if isinstance(scheme, hh.FDForward):
        want = (price(x0 * (1 + eps)) - price(x0)) / (x0 * eps)
    elif isinstance(scheme, hh.FDBackward):
        want = (price(x0) - price(x0 * (1 - eps))) / (x0 * eps)
    else:
        want = (price(x0 * (1 + eps)) - price(x0 * (1 - eps))) / (2 * eps * x0)
    got = hh.solve(hh.GreekProblem(prob, lens), hh.FiniteDifference(eps, scheme), m).greek
    assert np.float64(got).tobytes() == np.float64(want).tobytes()


@pytest.mark.parametrize("path", ["market_inputs.spot", "market_inputs.rate.rate", "market_inputs.ρ", "payoff.strike",
                                  "market_inputs.V0"])
def test_finite_difference_greek_on_the_exact_heston_law(hhlib, path):
    """The same through HestonBroadieKaya — the reference's only way to a Greek there (no dual numbers through rand!,
    heston.jl:261-276): a bumped spot, rate, ρ or strike shares ONE chain with its partner (bk_refinish_kernel), a
    bumped V0 runs two; either way the number of compute_fd_derivative on two plain solves, exactly."""
    prob, lens, eps = heston_problem(), hh.optic(path), 1e-3
    m = hh.MonteCarlo(hh.HestonDynamics(), hh.HestonBroadieKaya(), hh.SimulationConfig(20_000, steps=1, seeds=seeds_for(20_000, 23)))
    x0 = lens(prob)
    price = lambda x: hh.solve(hh.set(prob, lens, x), m, ensemble=False).price
    want = (price(x0 * (1 + eps)) - price(x0 * (1 - eps))) / (2 * eps * x0)
    got = hh.solve(hh.GreekProblem(prob, lens), hh.FiniteDifference(eps), m).greek
    assert np.float64(got).tobytes() == np.float64(want).tobytes()
    spot = hh.optic("market_inputs.spot")
    if path == "market_inputs.spot":  # … and the gamma: three models, one chain
        e2, s0 = 0.5, spot(prob)
        f = lambda x: hh.solve(hh.set(prob, spot, x), m, ensemble=False).price
        gamma = (f(s0 + e2) - 2 * f(s0) + f(s0 - e2)) / e2**2
        got = hh.solve(hh.SecondOrderGreekProblem(prob, spot, spot), hh.FiniteDifference(e2), m).greek
        assert np.float64(got).tobytes() == np.float64(gamma).tobytes()


def test_second_order_greeks_are_the_references_stencils(hhlib):
    prob, m, eps = heston_problem(), heston_method(anti=True), 0.5
    spot, v0 = hh.optic("market_inputs.spot"), hh.optic("market_inputs.V0")
    f = lambda x, y, l1, l2: hh.solve(hh.set(hh.set(prob, l1, x), l2, y), m, ensemble=False).price
    x0 = spot(prob)
    gamma = (f(x0 + eps, x0 + eps, spot, spot) - 2 * f(x0, x0, spot, spot) + f(x0 - eps, x0 - eps, spot, spot)) / eps**2
    got = hh.solve(hh.SecondOrderGreekProblem(prob, spot, spot), hh.FiniteDifference(eps), m).greek
    assert np.float64(got).tobytes() == np.float64(gamma).tobytes()
    e2, y0 = 0.004, v0(prob)
    cross = (f(x0 + e2, y0 + e2, spot, v0) - f(x0 + e2, y0 - e2, spot, v0) - f(x0 - e2, y0 + e2, spot, v0)
             + f(x0 - e2, y0 - e2, spot, v0)) / (4 * e2**2)
    got = hh.solve(hh.SecondOrderGreekProblem(prob, spot, v0), hh.FiniteDifference(e2), m).greek
    assert np.float64(got).tobytes() == np.float64(cross).tobytes()


def test_batch_of_finite_differences_shares_its_passes(hhlib):
    prob, m, eps = heston_problem(), heston_method(), 1e-3
    lenses = (hh.optic("market_inputs.spot"), hh.optic("market_inputs.V0"), hh.optic("market_inputs.κ"))
    got = hh.solve(hh.BatchGreekProblem(prob, lenses), hh.FiniteDifference(eps), m)
    for lens in lenses:
        one = hh.solve(hh.GreekProblem(prob, lens), hh.FiniteDifference(eps), m).greek
        assert np.float64(got[lens]).tobytes() == np.float64(one).tobytes()
    # and the bumped delta is the AD delta to the bump's order
    ad = hh.solve(hh.GreekProblem(prob, lenses[0]), hh.ForwardAD(), m).greek
    assert got[lenses[0]] == pytest.approx(ad, rel=2e-2)


def test_multi_at_the_headline_size_both_modes(hhlib):
    """10^6 x 252: the central bump of the spot, GENERATE and REPLAY, against two plain solves."""
    ctx = hhlib
    n_paths, n_steps = 1_000_000, 252
    seeds = _ffi.DeviceBuffer(ctx, 8 * n_paths).upload(seeds_for(n_paths, 1))
    dW = _ffi.DeviceBuffer(ctx, 8 * ctx.lib.hh_replay_elems(n_paths, n_steps, HES))
    models = [o.make_model(S0=100.1), o.make_model(S0=99.9)]
    ctx.check(ctx.lib.hh_wiener_fill(ctx.handle, HES, models[0].rho, models[0].T, n_steps, n_paths, seeds.ptr, 1, dW.ptr))
    for noise in (GEN, REP):
        c = o.make_config(HES, EM, n_paths, n_steps, noise_mode=noise)
        c.seeds, c.seeds_on_device, c.seeds_len = seeds.ptr, 1, n_paths
        c.replay, c.replay_on_device = dW.ptr, 1
        each = solve_each(ctx, models, c, False)
        multi = solve_multi(ctx, models, c, False)
        assert all(same_bits(a[0], b[0]) for a, b in zip(each, multi))
        delta = (multi[0][0].price - multi[1][0].price) / 0.2
        assert delta == pytest.approx(0.6557, abs=5e-3)  # Carr–Madan ∂/∂S0 of H252, Euler bias within


# ---- sharded over the devices of a hh_mgpu (on a one-GPU box: several contexts on device 0) ----------------

@pytest.mark.parametrize("G", [2, 3])
@pytest.mark.parametrize("strat,noise", [(EM, GEN), (EM, REP), (EXACT, GEN)])
def test_multi_over_shards(hhlib, oracle, G, strat, noise):
    dyn = GBM if strat == EXACT else HES
    n_paths, n_steps = 256 * 14 + 5, (1 if strat == EXACT else 9)
    seeds = seeds_for(n_paths, 9)
    models = bumped_models(dyn, 3, same_noise_law=True)
    rep = oracle.wiener_fill(dyn, models[0].rho, models[0].T, n_steps, seeds) if noise == REP else None
    c = o.make_config(dyn, strat, n_paths, n_steps, noise_mode=noise, seeds=seeds, replay=rep, antithetic=1)
    one = solve_multi(hhlib, models, c, False)
    mg = _ffi.MultiGpu([0] * G, _ffi.HH_MGPU_HOST_SUM)
    try:
        many = mg.solve_multi(models, c)
    finally:
        mg.close()
    for k in range(3):
        assert many[k].price == pytest.approx(one[k][0].price, rel=1e-13)
        assert many[k].std_error == pytest.approx(one[k][0].std_error, rel=1e-10)
        assert many[k].n_paths_done == n_paths


def test_host_mirror_devices_keyword_reaches_bumped_greeks(hhlib):
    import dataclasses
    prob, m, eps = heston_problem(), heston_method(), 1e-3
    m2 = dataclasses.replace(m, devices=(0, 0))
    lens = hh.optic("market_inputs.spot")
    g1 = hh.solve(hh.GreekProblem(prob, lens), hh.FiniteDifference(eps), m).greek
    g2 = hh.solve(hh.GreekProblem(prob, lens), hh.FiniteDifference(eps), m2).greek
    assert g2 == pytest.approx(g1, rel=1e-9)  # a difference of two prices that agree to 1e-13


def test_random_model_sets_share_a_pass_bit_for_bit(hhlib):
    """60 random sets of 2-6 models (every scalar drawn afresh, also ρ, T, call/put), random shapes and modes: the
    shared pass against one solve per model, bitwise."""
    rng = np.random.default_rng(20261004)
    for case in range(60):
        K = int(rng.integers(2, 7))
        dyn = HES if rng.random() < 0.7 else GBM
        strat = EM if dyn == HES or rng.random() < 0.5 else EXACT
        n_paths = int(rng.integers(1, 3000))
        n_steps = 1 if strat == EXACT else int(rng.integers(1, 40))
        models = [o.make_model(S0=float(rng.uniform(50, 150)), V0=float(rng.uniform(0.005, 0.3)), kappa=float(rng.uniform(0.1, 5)),
                               theta=float(rng.uniform(0.01, 0.3)), sigma=float(rng.uniform(0.05, 1.0)),
                               rho=float(rng.uniform(-0.95, 0.95)), r=float(rng.uniform(-0.02, 0.1)), T=float(rng.uniform(0.05, 3.0)),
                               strike=float(rng.uniform(50, 150)), cp=float(rng.choice([-1.0, 1.0]))) for _ in range(K)]
        c = o.make_config(dyn, strat, n_paths, n_steps, antithetic=int(rng.integers(0, 2)), em_split=int(rng.integers(0, 2)),
                          seeds=seeds_for(n_paths, case))
        each = solve_each(hhlib, models, c, True)
        multi = solve_multi(hhlib, models, c, True)
        for k in range(K):
            assert same_bits(each[k][0], multi[k][0]), (case, k)
            assert each[k][1].tobytes() == multi[k][1].tobytes(), (case, k)


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif("_n_gpus() < 2", reason="needs two GPUs: a context on a device that is not the calling thread's current one")
def test_accumulate_multi_broadie_kaya_from_a_thread_on_another_device(hhlib):
    """hh_mc_accumulate_multi grows the context's record buffer before it launches anything: it must do so on the
    CONTEXT's device whatever the calling thread's current device is (the Broadie–Kaya branch returns before the
    common hipSetDevice).  A context on device 1 driven from a thread that sits on device 0 gives device 0's bits."""
    import threading

    import torch
    n_paths = 256 * 37 + 45  # above one tile row, ragged: the doubled record buffer is (re)allocated by this call
    models = bk_models()[:3]
    K = len(models)
    c = o.make_config(HES, BK, n_paths, 1, seeds=seeds_for(n_paths, 1))
    want = solve_multi(hhlib, models, c, False)
    other = _ffi.Context(1)
    acc = torch.zeros(K * _ffi.HH_ACC_LEN, dtype=torch.float64, device="cuda:1")
    got, err = [], []

    def run():
        try:
            torch.cuda.set_device(0)  # this thread's current device is NOT the context's
            other.check(other.lib.hh_mc_accumulate_multi(other.handle, (_ffi.hh_model * K)(*models), K, C.byref(c),
                                                          acc.data_ptr(), None))
            other.check(other.lib.hh_ctx_synchronize(other.handle))
            host = acc.cpu().numpy()
            for k in range(K):
                r = _ffi.hh_result()
                a = np.ascontiguousarray(host[k * _ffi.HH_ACC_LEN:(k + 1) * _ffi.HH_ACC_LEN])
                other.check(other.lib.hh_mc_finalize(C.byref(models[k]), C.byref(c), a.ctypes.data, C.byref(r)))
                got.append(r)
        except Exception as e:  # noqa: BLE001 — handed to the asserting thread
            err.append(e)

    t = threading.Thread(target=run)
    t.start()
    t.join()
    other.close()
    assert not err, err
    for k in range(K):
        assert same_bits(want[k][0], got[k]), k
