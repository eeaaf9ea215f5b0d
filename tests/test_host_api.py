"""Host mirror of the reference's operator interface — everything that needs no GPU."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import hedgehog_jl_amd as hh
from hedgehog_jl_amd import _ffi
from hedgehog_jl_amd.montecarlo import _model_and_config
from tests.conftest import HAS_GPU, ROOT


def heston_problem(cp=hh.Call()):
    ref = hh.Date(2020, 1, 1)
    payoff = hh.VanillaOption(100.0, hh.add_years(ref, 1), hh.European(), cp, hh.Spot())
    return hh.PricingProblem(payoff, hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7))


def bs_problem():
    ref = hh.Date(2020, 1, 1)
    payoff = hh.VanillaOption(1.0, hh.Date(2021, 1, 1), hh.European(), hh.Call(), hh.Spot())
    return hh.PricingProblem(payoff, hh.BlackScholesInputs(ref, 0.03, 1.0, 1.0))


def test_yearfrac_act365():
    assert hh.yearfrac(hh.Date(2020, 1, 1), hh.Date(2021, 1, 1)) == 366 / 365
    assert hh.yearfrac(hh.Date(2021, 1, 1), hh.Date(2022, 1, 1)) == 1.0
    assert hh.to_ticks(hh.Date(1, 1, 1)) == 366 * 86400000  # Julia: date2epochdays(Date(1)) == 366
    assert hh.to_ticks(12345) == 12345


def test_simulation_config_contract():
    cfg = hh.SimulationConfig(10)
    assert cfg.steps == 1 and isinstance(cfg.variance_reduction, hh.NoVarianceReduction)
    assert cfg.seeds.dtype == np.uint64 and len(cfg.seeds) == 10
    with pytest.raises(ValueError, match="must be ≥ number of trajectories"):  # montecarlo.jl:65-66
        hh.SimulationConfig(10, seeds=[1, 2, 3])
    c2 = cfg.replace(seeds=np.arange(20), variance_reduction=hh.Antithetic())
    assert c2.trajectories == 10 and isinstance(c2.variance_reduction, hh.Antithetic)
    # the config owns a frozen copy of its seeds (its device copies are cached with it): the caller's
    # array stays the caller's, and a config made from a config shares the frozen vector
    mine = np.arange(1, 11, dtype=np.uint64)
    c3 = hh.SimulationConfig(10, seeds=mine)
    mine[0] = 99
    assert c3.seeds[0] == 1 and mine.flags.writeable and not c3.seeds.flags.writeable
    with pytest.raises(ValueError):
        c3.seeds[0] = 5
    assert c3.replace(steps=7).seeds is c3.seeds


def test_model_packing_heston_euler():
    prob = heston_problem()
    method = hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(),
                           hh.SimulationConfig(100, steps=50, variance_reduction=hh.Antithetic()))
    m, c, _, P, D = _model_and_config(prob, method)
    assert (c.dynamics, c.strategy, c.antithetic, c.n_steps, c.n_paths) == (1, 0, 1, 50, 100)
    assert m.T == 366 / 365 and m.r_drift == 0.03 and m.cp == 1.0 and P == 0
    assert m.discount == pytest.approx(np.exp(-0.03 * 366 / 365), rel=1e-15)
    assert (m.V0, m.kappa, m.theta, m.sigma, m.rho) == (0.04, 2.0, 0.04, 0.3, -0.7)


def test_model_packing_duals():
    prob = bs_problem()
    method = hh.MonteCarlo(hh.LognormalDynamics(), hh.BlackScholesExact(), hh.SimulationConfig(10))
    p2 = hh.set(prob, hh.ZeroRateSpineLens(1), hh.Dual(0.03, (1.0,)))
    m, c, keep, P, D = _model_and_config(p2, method)
    assert P == 1 and c.n_partials == 1 and c.strategy == _ffi.HH_EXACT_LAW
    assert m.dr_drift[0] == 1.0
    T = 366 / 365
    assert m.ddiscount[0] == pytest.approx(-T * np.exp(-0.03 * T), rel=1e-14)  # rate_curve.jl:149
    assert not m.dS0 and not m.dsigma
    p3 = hh.set(prob, hh.VolLens(1, 1), hh.Dual(1.0, (1.0,)))
    m, *_ = _model_and_config(p3, method)
    assert m.dsigma[0] == 1.0 and not m.dr_drift


def test_unsupported_combinations_raise_method_error():
    prob = heston_problem()
    cfg = hh.SimulationConfig(10)
    with pytest.raises(hh.MethodError):  # Heston inputs + lognormal dynamics: no sde_problem method
        _model_and_config(prob, hh.MonteCarlo(hh.LognormalDynamics(), hh.EulerMaruyama(), cfg))
    with pytest.raises(hh.MethodError):
        _model_and_config(prob, hh.MonteCarlo(hh.HestonDynamics(), hh.BlackScholesExact(), cfg))
    amer = hh.PricingProblem(hh.VanillaOption(100.0, hh.Date(2021, 1, 1), hh.American(), hh.Call(),
                                              hh.Spot()), prob.market_inputs)
    with pytest.raises(hh.MethodError):  # montecarlo.jl:479: European + Spot only
        _model_and_config(amer, hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(), cfg))
    with pytest.raises(hh.MethodError):
        hh.solve(prob, "not a method")
    with pytest.raises(TypeError):  # pricing_methods.jl:30-36: no ZeroRateSpineLens getter for Heston
        hh.ZeroRateSpineLens(1)(prob)
    with pytest.raises(TypeError):  # greeks_problem.jl:66-69: HestonInputs has no sigma surface
        hh.VolLens(1, 1)(prob)


def test_lenses_get_set():
    prob = heston_problem()
    for lens, val in ((hh.SpotLens(), 100.0), (hh.optic("market_inputs.spot"), 100.0),
                      (hh.optic("_.market_inputs.V0"), 0.04),
                      (hh.optic("market_inputs.rate.rate"), 0.03)):
        assert lens(prob) == val
        p2 = hh.set(prob, lens, val * 2)
        assert lens(p2) == val * 2 and lens(prob) == val  # functional update
    bs = bs_problem()
    assert hh.VolLens(1, 1)(bs) == 1.0 and hh.ZeroRateSpineLens(1)(bs) == 0.03
    assert hh.ZeroRateSpineLens(1)(hh.set(bs, hh.ZeroRateSpineLens(1), 0.05)) == 0.05
    assert hh.VolLens(1, 1)(hh.set(bs, hh.VolLens(1, 1), 0.5)) == 0.5
    assert len({hh.SpotLens(): 1, hh.optic("market_inputs.spot"): 2, hh.VolLens(1, 1): 3}) == 3


def test_dual_arithmetic():
    from hedgehog_jl_amd.dual import dexp, dlog
    x = hh.Dual(2.0, (1.0, 0.0))
    y = hh.Dual(3.0, (0.0, 1.0))
    z = dexp(-x * y) + dlog(x) / y - 2 * x ** 2
    assert z.value == pytest.approx(np.exp(-6) + np.log(2) / 3 - 8)
    assert z.partials[0] == pytest.approx(-3 * np.exp(-6) + 1 / 6 - 8)
    assert z.partials[1] == pytest.approx(-2 * np.exp(-6) - np.log(2) / 9)
    assert (1 - x).partials == (-1.0, 0.0) and (1 / x).partials[0] == -0.25
    assert x > 1 and x < y


def test_monte_carlo_solution_is_a_plain_frozen_dataclass():
    """pricing_solutions.jl:22-27: a value type.  `dataclasses.replace` / `asdict` / `fields` work, equality
    covers the samples, and samples that were left on the device are downloaded ONCE on the first read."""
    import dataclasses

    from hedgehog_jl_amd.domain import MonteCarloSolution, _DeviceSamples
    calls = []

    def fetch():
        calls.append(1)
        return np.arange(4.0)

    sol = MonteCarloSolution("prob", "method", 1.5, _DeviceSamples(fetch), std_error=0.1)
    assert [f.name for f in dataclasses.fields(sol)][:4] == ["problem", "method", "price", "ensemble"]
    assert not calls                                  # nothing moved yet
    np.testing.assert_array_equal(sol.ensemble, np.arange(4.0))
    assert sol.ensemble is sol.ensemble and calls == [1]
    other = dataclasses.replace(sol, price=2.0)
    assert other.price == 2.0 and other.ensemble is sol.ensemble and other != sol
    assert dataclasses.replace(sol) == sol
    assert dataclasses.asdict(sol)["price"] == 1.5
    with pytest.raises(dataclasses.FrozenInstanceError):
        sol.price = 3.0
    pair = MonteCarloSolution("p", "m", 1.0, (np.ones(2), np.zeros(2)))
    assert pair == MonteCarloSolution("p", "m", 1.0, (np.ones(2), np.zeros(2)))
    assert pair != MonteCarloSolution("p", "m", 1.0, (np.ones(2), np.ones(2)))

    class Closed:
        handle = None
    gone = MonteCarloSolution("p", "m", 1.0, _DeviceSamples(fetch, Closed()))
    with pytest.raises(RuntimeError, match="Context has been closed"):
        gone.ensemble


def test_cabi_header_and_library_agree():
    """Every function include/hedgehog_mc.h declares is bound in _ffi.SYMBOLS and exported by the
    built library (no compute call is made here)."""
    hdr = open(os.path.join(ROOT, "include", "hedgehog_mc.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|size_t|uint64_t|const char\*|hh_ctx\*)\s+(hh_\w+)\s*\(", hdr, re.M))
    bound = {s[0] for s in _ffi.SYMBOLS}
    assert declared == bound, declared ^ bound
    lib = hh.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.hh_abi_version() == 6
    assert lib.hh_replay_elems(257, 3, 1) == 2 * 3 * 2 * 256
    assert lib.hh_replay_elems(256, 5, 0) == 5 * 256
    # struct layouts seen by ctypes == what the compiler laid out (sizes are part of the ABI)
    assert C.sizeof(_ffi.hh_model) == 11 * 8 + 8 * 8
    assert C.sizeof(_ffi.hh_result) == 4 * 8 + 8 * 8 + 5 * 8 + 2 * 8
    assert C.sizeof(_ffi.hh_config) == 10 * 4 + 2 * 4 + 2 * 8 + 2 * 8 + 4 * 8 + 2 * 4 + 4 * 4 + 2 * 8


def test_marginal_law_is_the_reference_formula_quirk_included():
    """montecarlo.jl:293-303: Normal(log S0 + (r − σ²/2)·√α, σ·√α) — √α in the MEAN as written there (Q1)."""
    import math
    ref = hh.Date(2020, 1, 1)
    for expiry, alpha in ((hh.Date(2021, 1, 1), 366 / 365), (hh.Date(2025, 1, 1), 1827 / 365)):
        prob = hh.PricingProblem(hh.VanillaOption(1.0, expiry, hh.European(), hh.Put(), hh.Spot()),
                                 hh.BlackScholesInputs(ref, 0.03, 1.0, 0.04))
        law = hh.marginal_law(prob, hh.LognormalDynamics(), expiry)
        assert law.std() == pytest.approx(0.04 * math.sqrt(alpha), rel=1e-15)
        assert law.mean() == pytest.approx((0.03 - 0.04**2 / 2) * math.sqrt(alpha), rel=1e-14)
        fixed = hh.marginal_law(prob, hh.LognormalDynamics(), expiry, compat_sqrt_alpha=False)
        assert fixed.mean() == pytest.approx((0.03 - 0.04**2 / 2) * alpha, rel=1e-14) and fixed.var() == law.var()
    d = hh.marginal_law(hh.PricingProblem(prob.payoff, hh.BlackScholesInputs(ref, 0.03, hh.Dual(2.0, (1.0,)), 0.04)),
                        hh.LognormalDynamics(), expiry)
    assert d.mean().partials == (0.5,)  # d log S0 / dS0
    with pytest.raises(hh.MethodError):
        hh.marginal_law(heston_problem(), hh.HestonDynamics(), expiry)


def test_library_shard_ranges_partition_the_ensemble():
    """hh_mgpu_shard_range (pure host arithmetic, callable without a GPU): the ranges hh_mgpu_solve cuts —
    [g·per, min(N, (g+1)·per)), per = ⌈N/G⌉, whole tiles for tile-major REPLAY data — cover
    the ensemble exactly once, in order, for every G the context accepts."""
    lib = hh.load_library()
    a, b = C.c_uint64(), C.c_uint64()
    for n in (1, 2, 255, 256, 257, 1000, 10_001, 1_000_000, 2**32 - 256):
        for G in (1, 2, 3, 5, 8, 64):
            for tile in (0, 1):
                at, prev_len = 0, None
                for g in range(G):
                    lib.hh_mgpu_shard_range(n, G, g, tile, C.byref(a), C.byref(b))
                    assert a.value == at and a.value <= b.value <= n
                    if tile and b.value < n:
                        assert b.value % 256 == 0
                    if prev_len is not None:
                        assert b.value - a.value <= prev_len  # the remainder shard is the last non-empty one
                    prev_len, at = b.value - a.value, b.value
                assert at == n
    lib.hh_mgpu_shard_range(10, 0, 0, 0, C.byref(a), C.byref(b))  # bad arguments give the empty range
    assert (a.value, b.value) == (0, 0)
    lib.hh_mgpu_shard_range(10, 2, 5, 0, C.byref(a), C.byref(b))
    assert (a.value, b.value) == (0, 0)


def test_rccl_stand_in_exports_what_the_library_binds():
    """tests/c/stub_rccl.hip must keep up with the RCCL entry points hh_mgpu.hip resolves with dlsym —
    otherwise the multi-rank tests would silently fall back to the host sum."""
    import subprocess
    from tests.c.build_stub import STUB, build_stub
    build_stub()
    src = open(os.path.join(ROOT, "hedgehog.jl_amd", "csrc", "hh_mgpu.hip")).read()
    bound = set(re.findall(r'dlsym\(api\.lib, "(nccl\w+)"\)', src))
    assert {"ncclCommInitAll", "ncclCommDestroy", "ncclCommAbort", "ncclAllReduce", "ncclGroupStart", "ncclGroupEnd",
            "ncclGetErrorString", "ncclGetVersion"} == bound
    exported = subprocess.run(["nm", "-D", "--defined-only", STUB], capture_output=True, text=True, check=True).stdout
    for name in bound | {"stub_rccl_fail_at", "stub_rccl_counters"}:
        assert re.search(r"\bT " + name + r"\b", exported), name


def test_finalize_is_pure_host_arithmetic():
    lib = hh.load_library()
    from tests import oracle_ffi as o
    T = 1.0
    m = o.make_model(seeds={"discount": [-T * np.exp(-0.03)]}, n_partials=1)
    c = o.make_config(1, 0, 4, 1, n_partials=1)
    acc = np.zeros(16)
    acc[0], acc[1], acc[2], acc[10] = 10.0, 30.0, 2.0, 4.0
    r = _ffi.hh_result()
    assert lib.hh_mc_finalize(C.byref(m), C.byref(c), acc.ctypes.data, C.byref(r)) == 0
    D = np.exp(-0.03)
    assert r.price == pytest.approx(D * 2.5)
    assert r.std_error == pytest.approx(D * np.sqrt(((30 - 4 * 2.5**2) / 3) / 4))
    assert r.dprice[0] == pytest.approx(-T * D * 2.5 + D * 0.5)


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device the product path raises, it does not compute."""
    with pytest.raises(hh.HedgehogMCError):
        hh.Context(0)
    with pytest.raises(hh.HedgehogMCError):  # the several-GPU entry point likewise (HH_ERR_HIP from hh_mgpu_create)
        _ffi.MultiGpu([0, 1])
    prob = heston_problem()
    with pytest.raises(hh.HedgehogMCError):
        hh.solve(prob, hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(),
                                     hh.SimulationConfig(16, steps=4)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hedgehog.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower(), (dirpath, f)


def test_interpolated_rate_curve_scalars_and_spine_lens():
    """rate_curve.jl:60-91,182-186 (linear in the zero rate, constant extrapolation) and
    pricing_methods.jl:34-50.  The kernels only ever see the scalars the host resolves — including
    the reference's quirk Q4: the Euler drift uses zero_rate(curve, 0.0), i.e. the FIRST spine zero,
    while the exact law and the discount use the rate at expiry."""
    import math
    ref = hh.Date(2020, 1, 1)
    tenors = [0.25, 0.5, 1.0, 2.0, 5.0]
    zs = [0.02, 0.025, 0.03, 0.035, 0.04]
    curve = hh.RateCurve(ref, tenors, [math.exp(-z * t) for z, t in zip(zs, tenors)])
    assert curve.zeros == pytest.approx(zs, rel=1e-12) and hh.spine_zeros(curve) == list(curve.zeros)
    assert hh.zero_rate(curve, 0.0) == curve.zeros[0]                       # constant extrapolation
    assert hh.zero_rate(curve, hh.Date(2030, 1, 1)) == curve.zeros[-1]
    T = 366 / 365
    zT = zs[2] + (zs[3] - zs[2]) * (T - 1.0)
    assert hh.zero_rate(curve, hh.Date(2021, 1, 1)) == pytest.approx(zT, rel=1e-12)
    assert hh.df(curve, hh.Date(2021, 1, 1)) == pytest.approx(math.exp(-zT * T), rel=1e-13)
    with pytest.raises(ValueError):
        hh.RateCurve(ref, [1.0, 0.5], [0.9, 0.95])
    with pytest.raises(ValueError):
        hh.RateCurve(ref, [], [])

    payoff = hh.VanillaOption(100.0, hh.Date(2021, 1, 1), hh.European(), hh.Call(), hh.Spot())
    prob = hh.PricingProblem(payoff, hh.BlackScholesInputs(ref, curve, 100.0, 0.2))
    lens = hh.ZeroRateSpineLens(3)
    assert lens(prob) == curve.zeros[2]
    p2 = hh.set(prob, lens, hh.Dual(lens(prob), (1.0,)))
    cfg = hh.SimulationConfig(10, steps=5)
    m, c, _, P, _ = _model_and_config(p2, hh.MonteCarlo(hh.LognormalDynamics(),
                                                        hh.BlackScholesExact(), cfg))
    w3 = 1.0 - (T - 1.0)  # weight of spine node 3 at t = T
    assert m.r_drift == pytest.approx(zT) and m.dr_drift[0] == pytest.approx(w3)
    assert m.ddiscount[0] == pytest.approx(-T * w3 * math.exp(-zT * T))
    m, c, _, P, _ = _model_and_config(p2, hh.MonteCarlo(hh.LognormalDynamics(), hh.EulerMaruyama(),
                                                        cfg))
    assert m.r_drift == curve.zeros[0] and not m.dr_drift          # Q4: drift from the first node
    assert m.ddiscount[0] == pytest.approx(-T * w3 * math.exp(-zT * T))


def test_rate_curve_against_the_reference_test_vectors():
    """test/unit/rate_curve.jl: the spine (tenors, discount factors) and the flat-curve checks of
    the reference's own test, held as data in tests/golden/reference_known_answers.json — the host
    mirror resolves `discount` and `r_drift` from these curves before every solve."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                    "reference_known_answers.json")))["rate_curve"]
    ref = hh.Date(*g["reference_date"])
    curve = hh.RateCurve(ref, g["tenors"], g["dfs"])
    for t, d in zip(g["tenors"], g["dfs"]):
        assert hh.df_yf(curve, t) == pytest.approx(d, abs=g["atol"])                       # :26-28
        assert hh.zero_rate_yf(curve, t) == pytest.approx(-math.log(d) / t, abs=g["atol"])  # :31-33
    # constant extrapolation and linear interpolation in the zero rate (the interp of the test)
    zs = [-math.log(d) / t for d, t in zip(g["dfs"], g["tenors"])]
    assert hh.zero_rate_yf(curve, 0.01) == zs[0] and hh.zero_rate_yf(curve, 30.0) == zs[-1]
    assert hh.zero_rate_yf(curve, 0.75) == pytest.approx(0.5 * (zs[1] + zs[2]), abs=1e-15)
    r = g["flat_rate"]
    flat = hh.FlatRateCurve(r)
    for t in g["flat_yearfracs"] + g["flat_vector_yearfracs"]:
        assert hh.zero_rate_yf(flat, t) == r                                               # :41-42, 51
        assert hh.df_yf(flat, t) == pytest.approx(math.exp(-r * t), abs=g["atol"])
    t0, t1 = hh.Date(*g["flat_date_from"]), hh.Date(*g["flat_date_to"])
    flat = hh.FlatRateCurve(r, reference_date=t0)
    tau = 365 / 365
    assert hh.zero_rate(flat, t1) == pytest.approx(r, abs=g["atol"])                       # :61
    assert hh.df(flat, t1) == pytest.approx(math.exp(-r * tau), abs=g["atol"])              # :62


@pytest.mark.skipif(__import__("shutil").which("hipcc") is None, reason="needs hipcc")
def test_optional_ring_form_of_the_replay_kernel_still_compiles(tmp_path):
    """The LDS-DMA ring form of the price-only REPLAY kernel is kept behind compile-time knobs for
    A/B runs (tools/tune_replay.py): it must keep compiling for gfx950."""
    import subprocess
    src = os.path.join(ROOT, "hedgehog.jl_amd", "csrc", "hh_kernels.hip")
    out = tmp_path / "ring.o"
    proc = subprocess.run(["hipcc", "-c", "-O1", "-std=c++17", "--offload-arch=gfx950",
                           "-ffp-contract=off", "-DHH_REPLAY_LDS=4", "-DHH_REPLAY_PPT=2",
                           "-DHH_REPLAY_PAD_KIB=0", src, "-o", str(out)],
                          capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]


def test_full_path_entry_points_refuse_duals_instead_of_dropping_them():
    """float(Dual) is defined (value part), so a host layer that packs plain doubles would silently return a
    price without partials: LSM and the exact Heston paths carry none, and say so."""
    from hedgehog_jl_amd.dual import Dual
    from hedgehog_jl_amd.lsm import _lsm_structs
    ref, exp_ = hh.Date(2021, 1, 1), hh.Date(2022, 1, 1)
    put = hh.VanillaOption(100.0, exp_, hh.American(), hh.Put(), hh.Spot())
    mc = hh.MonteCarlo(hh.LognormalDynamics(), hh.BlackScholesExact(), hh.SimulationConfig(100, steps=10))
    for mkt in (hh.BlackScholesInputs(ref, 0.05, Dual(100.0, (1.0,)), 0.2),
                hh.BlackScholesInputs(ref, 0.05, 100.0, Dual(0.2, (1.0,))),
                hh.BlackScholesInputs(ref, Dual(0.05, (1.0,)), 100.0, 0.2)):
        with pytest.raises(hh.MethodError, match="FiniteDifference"):
            _lsm_structs(hh.PricingProblem(put, mkt), mc)
    _lsm_structs(hh.PricingProblem(put, hh.BlackScholesInputs(ref, 0.05, 100.0, 0.2)), mc)


def test_finite_difference_formulas_on_a_closed_form_method():
    """greeks_problem.jl:279-303, 396-422 on a method that is NOT MonteCarlo: the host layer's FiniteDifference
    solvers must then run the reference's plain sequence of solves (the shared pass is MonteCarlo's alone) —
    checked against the Black–Scholes closed forms the reference's own tests use (test/unit/black_scholes.jl,
    greeks_agreement.jl:32-120: AD = FD = analytic to 1e-5)."""
    import math
    ref, exp_ = hh.Date(2021, 1, 1), hh.Date(2022, 1, 1)
    S, K, r, sig, T = 100.0, 100.0, 0.05, 0.2, 1.0
    prob = hh.PricingProblem(hh.VanillaOption(K, exp_, hh.European(), hh.Call(), hh.Spot()),
                             hh.BlackScholesInputs(ref, r, S, sig))
    bs = hh.BlackScholesAnalytic()
    d1 = (math.log(S / K) + (r + 0.5 * sig * sig) * T) / (sig * math.sqrt(T))
    pdf = math.exp(-0.5 * d1 * d1) / math.sqrt(2 * math.pi)
    delta, gamma, vega = 0.5 * (1 + math.erf(d1 / math.sqrt(2))), pdf / (S * sig * math.sqrt(T)), S * pdf * math.sqrt(T)
    spot, vol = hh.SpotLens(), hh.VolLens(K, exp_)
    for scheme, tol in ((hh.FDCentral(), 1e-6), (hh.FDForward(), 2e-3), (hh.FDBackward(), 2e-3)):
        g = hh.solve(hh.GreekProblem(prob, spot), hh.FiniteDifference(1e-4, scheme), bs).greek
        assert g == pytest.approx(delta, rel=tol)
    assert hh.solve(hh.GreekProblem(prob, vol), hh.FiniteDifference(1e-4), bs).greek == pytest.approx(vega, rel=1e-6)
    assert hh.solve(hh.SecondOrderGreekProblem(prob, spot, spot), hh.FiniteDifference(1e-2), bs).greek == \
        pytest.approx(gamma, rel=1e-4)
    vanna = -pdf * (d1 - sig * math.sqrt(T)) / sig  # ∂²C/∂S∂σ
    assert hh.solve(hh.SecondOrderGreekProblem(prob, spot, vol), hh.FiniteDifference(1e-3), bs).greek == \
        pytest.approx(vanna, rel=1e-3)
    batch = hh.solve(hh.BatchGreekProblem(prob, (spot, vol)), hh.FiniteDifference(1e-4), bs)
    assert batch[spot] == pytest.approx(delta, rel=1e-6) and batch[vol] == pytest.approx(vega, rel=1e-6)


def test_problems_that_cannot_share_a_pass_are_left_to_the_plain_loop():
    """solve_montecarlo_many says None (and touches no GPU) for what hh_mc_solve_multi does not take: a dual number in
    a problem, one problem alone, more than HH_MAX_MODELS problems."""
    from hedgehog_jl_amd.dual import Dual
    ref, exp_ = hh.Date(2021, 1, 1), hh.Date(2022, 1, 1)
    call = hh.VanillaOption(100.0, exp_, hh.European(), hh.Call(), hh.Spot())
    mc = hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(), hh.SimulationConfig(50, steps=3, seeds=range(1, 51)))
    plain = hh.PricingProblem(call, hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7))
    dual = hh.PricingProblem(call, hh.HestonInputs(ref, 0.03, Dual(100.0, (1.0,)), 0.04, 2.0, 0.04, 0.3, -0.7))
    assert hh.solve_montecarlo_many([plain], mc) is None
    assert hh.solve_montecarlo_many([plain] * (_ffi.HH_MAX_MODELS + 1), mc) is None
    assert hh.solve_montecarlo_many([plain, dual], mc) is None
