"""Two Broadie–Kaya code paths of the shipped library against their plain forms, bit for bit (a second build of the
library, tests/c/build_bk_check.py):

* log I_ν(ν_κ) and the characteristic function at 0 (moments_from_cf, sample_from_cf.jl:50-61; heston.jl:184-212) are
  evaluated on the real axis — besseli_logmul_re, chf_at_zero — where the plain form goes through the complex code:
  the real code is the complex code's real parts operation by operation, so NO sample may move;
* the bisection ladder of a trajectory whose secant failed (sample_from_cf.jl:123-133) is run by its WAVE — the lanes
  evaluate the next levels of the bisection tree at once, then walk them with the loop's own tests (wave_ladder) —
  where the plain form is that loop in the failed lane: the same abscissae, the same decisions, the same ∫V, the same
  decision words and counters.

Every regime of tests/test_gpu_bk.py, plus controls that send nearly every trajectory to the ladder (batches of eight
per wave), a lone one per wave, iteration caps that end a ladder in the middle of a tree, tolerances that end it at
its first level or after fifty."""
import ctypes as C
import os

import numpy as np
import pytest

from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o
from tests.test_gpu_bk import PARAMS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def check_lib():
    from tests.c.build_bk_check import build_bk_check
    try:
        path = build_bk_check()
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"the check build of the library could not be made: {e}")
    lib = C.CDLL(path)
    for name, res, args in _ffi.SYMBOLS:
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    h = C.c_void_p()
    assert lib.hh_ctx_create(C.byref(h), 0) == 0
    yield lib, h
    lib.hh_ctx_destroy(h)


def solve(lib, h, prm, n, seed, **controls):
    m = o.make_model(**prm)
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, seeds=[seed])
    for k, v in controls.items():
        setattr(c, k, v)
    res = _ffi.hh_result()
    term = np.zeros(n)
    assert lib.hh_mc_solve(h, C.byref(m), C.byref(c), C.byref(res), term.ctypes.data) == 0
    return res, term


@pytest.mark.parametrize("name", sorted(PARAMS))
def test_real_axis_setup_and_wave_ladder_move_no_sample(hhlib, check_lib, name):
    lib, h = check_lib
    n = 30_000
    r1, t1 = solve(hhlib.lib, hhlib.handle, PARAMS[name], n, 777)
    r0, t0 = solve(lib, h, PARAMS[name], n, 777)
    assert t1.tobytes() == t0.tobytes()
    for f in ("price", "std_error", "bk_newton_fail", "bk_bisect_fallback", "bk_maxguess_fallback", "bk_cf_terms"):
        assert getattr(r1, f) == getattr(r0, f), f


def solve_replay(lib, h, prm, draws):
    n = draws.shape[1]
    m = o.make_model(**prm)
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, noise_mode=_ffi.HH_NOISE_REPLAY, replay=draws.ravel())
    res = _ffi.hh_result()
    term = np.zeros(n)
    assert lib.hh_mc_solve(h, C.byref(m), C.byref(c), C.byref(res), term.ctypes.data) == 0
    return res, term


def same(r1, t1, r0, t0):
    assert t1.tobytes() == t0.tobytes()
    for f in ("price", "std_error", "bk_newton_fail", "bk_bisect_fallback", "bk_maxguess_fallback", "bk_cf_terms"):
        assert getattr(r1, f) == getattr(r0, f), f


def decisions(lib, h, n):
    dec, ln = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    assert lib.hh_bk_decisions(h, n, dec.ctypes.data, ln.ctypes.data) == 0
    return dec, ln


@pytest.mark.parametrize("controls", [
    dict(bk_newton_maxiter=2),                                   # every secant gives up at once: 64 failed lanes per wave
    dict(bk_newton_maxiter=2, bk_bisect_maxiter=1),              # … and the ladder's cap ends it inside the first tree
    dict(bk_newton_maxiter=2, bk_bisect_maxiter=3),
    dict(bk_newton_maxiter=2, bk_bisect_maxiter=7),
    dict(bk_newton_maxiter=3, bk_atol=1e-13),                    # ~45 midpoints: many turns of the wave
    dict(bk_newton_maxiter=2, bk_atol=0.5),                      # the width test ends it at the first midpoints
    dict(bk_newton_maxiter=4),                                   # a mixed wave: some lanes in the ladder, some not
], ids=lambda c: "-".join(f"{k[3:]}={v}" for k, v in c.items()))
def test_wave_ladder_against_the_lane_by_lane_loop(hhlib, check_lib, controls):
    lib, h = check_lib
    n = 50_000
    for name in ("h252", "q2"):
        r1, t1 = solve(hhlib.lib, hhlib.handle, PARAMS[name], n, 31, **controls)
        d1 = decisions(hhlib.lib, hhlib.handle, n)
        r0, t0 = solve(lib, h, PARAMS[name], n, 31, **controls)
        d0 = decisions(lib, h, n)
        same(r1, t1, r0, t0)
        assert (d1[0] == d0[0]).all() and (d1[1] == d0[1]).all()
        if controls.get("bk_newton_maxiter") == 2 and "bk_atol" not in controls:
            assert r1.bk_newton_fail > 0.5 * n


def test_wave_ladder_lone_failures(hhlib, check_lib):
    """The caller's draws (REPLAY): the same V_T and the median u for every trajectory — the secant converges — but for
    a few dozen far in the tails, spread over 1172 tiles: a wave has at most one or two of them, so all 64 (or 32)
    lanes walk one trajectory's tree."""
    lib, h = check_lib
    n = 300_000
    rng = np.random.default_rng(3)
    draws = np.stack([np.full(n, 0.045), np.full(n, 0.5), rng.standard_normal(n)])
    special = rng.choice(n, 96, replace=False)
    draws[1, special] = np.where(np.arange(96) % 2 == 0, 1e-9, 1.0 - 1e-9)
    draws = np.ascontiguousarray(draws)
    r1, t1 = solve_replay(hhlib.lib, hhlib.handle, PARAMS["h252"], draws)
    r0, t0 = solve_replay(lib, h, PARAMS["h252"], draws)
    same(r1, t1, r0, t0)
    assert 0 < r1.bk_newton_fail <= 96  # sparse: fewer than one failure per four tiles, yet some


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 9, 17, 64])
def test_wave_ladder_group_sizes(hhlib, check_lib, k):
    """k lanes of ONE tile with uniforms far in a tail (the caller's draws; which tail fails the secant is the
    model's business, so both are tried): the group widths 64 / 32 / 16 / 8 and the batches behind the first eight,
    in one wave and spread over the four"""
    lib, h = check_lib
    n = 1024
    fails = 0
    for variant in range(4):
        rng = np.random.default_rng(100 * k + variant)
        draws = np.stack([np.full(n, 0.045), np.full(n, 0.5), rng.standard_normal(n)])
        where = 256 + (64 * 2 + rng.choice(64, k, replace=False) if variant < 2 else rng.choice(256, k, replace=False))
        draws[1, where] = (1e-9 if variant % 2 == 0 else 1.0 - 1e-9)
        draws = np.ascontiguousarray(draws)
        r1, t1 = solve_replay(hhlib.lib, hhlib.handle, PARAMS["h252"], draws)
        r0, t0 = solve_replay(lib, h, PARAMS["h252"], draws)
        same(r1, t1, r0, t0)
        assert r1.bk_newton_fail <= k
        fails += r1.bk_newton_fail
    assert fails >= k  # one of the two tails sends its lanes to the ladder


def test_model_constants_beside_the_bessel_tables_follow_the_model(hhlib):
    """The tables a chain finds in place are keyed by ν — and, since ϕ(0) takes four model constants from there (CfZero:
    functions of κ, σ², T), by those too.  Three models with the SAME ν = 2κθ/σ² − 1 solved back to back in one context,
    each against a context of its own."""
    base = dict(PARAMS["h252"])
    models = [base,
              dict(base, kappa=4.0, theta=0.02),            # κθ unchanged: the same ν, another κ
              dict(base, T=0.5),                            # the same ν, κ, σ: another T
              dict(base, kappa=8.0, theta=0.04, sigma=0.6)]  # κθ/σ² unchanged: another σ² as well
    nus = {round(2 * m["kappa"] * m["theta"] / m["sigma"] ** 2 - 1, 12) for m in models}
    assert len(nus) == 1
    n = 20_000
    shared = [solve(hhlib.lib, hhlib.handle, m, n, 5)[1] for m in models + models[::-1]]
    for m, got in zip(models + models[::-1], shared):
        ctx = _ffi.Context(0)
        try:
            want = solve(ctx.lib, ctx.handle, m, n, 5)[1]
        finally:
            ctx.close()
        assert got.tobytes() == want.tobytes()
