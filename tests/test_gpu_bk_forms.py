"""Two Broadie–Kaya code paths of round 5 against the forms they replaced, bit for bit (a second build of the library,
tests/c/build_bk_check.py):

* log I_ν(ν_κ) and the characteristic function at 0 (moments_from_cf, sample_from_cf.jl:50-61; heston.jl:184-212) are
  evaluated on the real axis — besseli_logmul_re, chf_at_zero — where they went through the complex code: the real
  code is the complex code's real parts operation by operation, so NO sample may move;
* the ladder kernel finds its trajectories by a search in LDS over the tiles of its chunk, where it searched the
  prefix sums of all tiles in place: the same trajectories, the same records.

Every regime of tests/test_gpu_bk.py, plus controls that send most trajectories to the ladder (many chunks) and almost
none (one chunk spanning every tile)."""
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
def test_real_axis_setup_and_staged_ladder_move_no_sample(hhlib, check_lib, name):
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


def test_ladder_chunks_dense(hhlib, check_lib):
    """every secant gives up after its two starting points: nearly all trajectories in the ladder, a chunk per tile"""
    lib, h = check_lib
    n = 100_000
    r1, t1 = solve(hhlib.lib, hhlib.handle, PARAMS["h252"], n, 31, bk_newton_maxiter=2)
    r0, t0 = solve(lib, h, PARAMS["h252"], n, 31, bk_newton_maxiter=2)
    assert t1.tobytes() == t0.tobytes()
    assert r1.price == r0.price and r1.bk_newton_fail == r0.bk_newton_fail and r1.bk_bisect_fallback == r0.bk_bisect_fallback
    assert r1.bk_newton_fail > 0.5 * n


def test_ladder_one_chunk_spanning_every_tile(hhlib, check_lib):
    """The caller's draws (REPLAY): the same V_T and the median u for every trajectory — the secant converges — but for
    a few dozen, far in the tails, spread over 1172 tiles: ONE ladder chunk whose trajectories lie more tiles apart
    than the kernel stages in LDS, so the shipped build searches in place there too (and must find the same ones)."""
    lib, h = check_lib
    n = 300_000
    rng = np.random.default_rng(3)
    draws = np.stack([np.full(n, 0.045), np.full(n, 0.5), rng.standard_normal(n)])
    special = rng.choice(n, 96, replace=False)
    draws[1, special] = np.where(np.arange(96) % 2 == 0, 1e-9, 1.0 - 1e-9)
    draws = np.ascontiguousarray(draws)
    r1, t1 = solve_replay(hhlib.lib, hhlib.handle, PARAMS["h252"], draws)
    r0, t0 = solve_replay(lib, h, PARAMS["h252"], draws)
    assert t1.tobytes() == t0.tobytes()
    assert r1.price == r0.price and r1.bk_newton_fail == r0.bk_newton_fail and r1.bk_bisect_fallback == r0.bk_bisect_fallback
    assert 0 < r1.bk_newton_fail <= 96  # sparse: fewer than one failure per four tiles, yet some


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
