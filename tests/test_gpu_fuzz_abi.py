"""Hostile scalars through the C-ABI: NaN / infinite / negative / huge model parameters, unknown enum
values, zero sizes.  The contract of include/hedgehog_mc.h: every entry point returns a status
(0 = OK, negative = invalid argument / unsupported combination) and hh_last_error() explains it; no
abort crosses the boundary, and no kernel loop is unbounded in its data (Broadie–Kaya leaves every
series / root-search loop on NaN; the samplers' rejection loops are capped).  A call that is accepted
may return NaN prices for NaN inputs — it must return."""
import ctypes as C
import math
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, Phase, given, settings
from hypothesis import strategies as st

from hedgehog_jl_amd import _ffi

pytestmark = pytest.mark.gpu


def _trace(*what):
    """HH_TEST_TRACE=<file>: the arguments of the call about to be made, for finding one that does not return."""
    path = os.environ.get("HH_TEST_TRACE")
    if path:
        with open(path, "a") as f:
            f.write(repr(what) + "\n")

weird = st.one_of(st.floats(allow_nan=True, allow_infinity=True, width=64),
                  st.sampled_from([0.0, -0.0, 1e-320, 1e-300, 1e300, -1.0, 1.0, 0.04, 2.0, 100.0, float("nan"),
                                   float("inf"), -float("inf")]))
sane = dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1.0, strike=100.0, cp=1.0)


@settings(max_examples=int(os.environ.get("HH_FUZZ_EXAMPLES", "800")), deadline=None, derandomize=True, database=None, phases=[Phase.explicit, Phase.generate],
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(which=st.lists(st.sampled_from(sorted(sane)), min_size=1, max_size=3, unique=True), vals=st.lists(weird, min_size=3, max_size=3),
       dynamics=st.sampled_from([0, 1, 1, 7]), strategy=st.sampled_from([0, 1, 2, 2, 9]), anti=st.sampled_from([0, 1]),
       n_paths=st.sampled_from([0, 1, 100, 300]), n_steps=st.sampled_from([0, 1, 7, 20]), P=st.sampled_from([0, 0, 2]),
       noise=st.sampled_from([0, 0, 0, 3]))
def test_hostile_scalars_return_a_status(hhlib, which, vals, dynamics, strategy, anti, n_paths, n_steps, P, noise):
    prm = dict(sane)
    for k, v in zip(which, vals):
        prm[k] = v
    sd = {"S0": [1.0, 0.0], "sigma": [0.0, 1.0]} if P else {}
    m = _ffi.make_model(seeds=sd, n_partials=P, discount=1.0, **prm)
    c = _ffi.make_config(dynamics, strategy, n_paths, n_steps, antithetic=anti, noise_mode=noise,
                         seeds=np.arange(1, max(n_paths, 1) + 1, dtype=np.uint64), n_partials=P)
    res = _ffi.hh_result()
    term = np.zeros(max(n_paths, 1) * 2)
    _trace("mc", prm, dynamics, strategy, anti, n_paths, n_steps, P, noise)
    rc = hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(m), C.byref(c), C.byref(res), term.ctypes.data)
    assert rc in (_ffi.HH_OK, _ffi.HH_ERR_INVALID, _ffi.HH_ERR_UNSUPPORTED), (rc, prm)
    if rc != _ffi.HH_OK:
        assert len(hhlib.lib.hh_last_error(hhlib.handle)) > 0
    elif all(math.isfinite(v) for v in prm.values()) and prm["S0"] < 1e100 and abs(prm["sigma"]) < 1e3:
        assert res.n_paths_done == n_paths


@settings(max_examples=int(os.environ.get("HH_FUZZ_EXAMPLES", "500")), deadline=None, derandomize=True, database=None, phases=[Phase.explicit, Phase.generate],
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(which=st.lists(st.sampled_from(["S0", "sigma", "r", "T", "strike"]), min_size=1, max_size=2, unique=True),
       vals=st.lists(weird, min_size=2, max_size=2), degree=st.sampled_from([0, 1, 5, 8, 9, -1]),
       n_paths=st.sampled_from([0, 1, 64, 500]), n_steps=st.sampled_from([0, 1, 2, 12]),
       disc=st.sampled_from([0.0, 1.0, 0.999, float("nan"), -1.0, 2.0]))
def test_hostile_scalars_lsm(hhlib, which, vals, degree, n_paths, n_steps, disc):
    prm = dict(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0, cp=-1.0)
    for k, v in zip(which, vals):
        prm[k] = v
    m = _ffi.make_model(discount=1.0, **prm)
    c = _ffi.make_config(0, 1, n_paths, n_steps, seeds=np.arange(1, max(n_paths, 1) + 1, dtype=np.uint64))
    res = _ffi.hh_lsm_result()
    _trace("lsm", prm, degree, n_paths, n_steps, disc)
    rc = hhlib.lib.hh_lsm_solve(hhlib.handle, C.byref(m), C.byref(c), degree, disc, C.byref(res), None, None, None)
    assert rc in (_ffi.HH_OK, _ffi.HH_ERR_INVALID, _ffi.HH_ERR_UNSUPPORTED), (rc, prm)


@settings(max_examples=int(os.environ.get("HH_FUZZ_EXAMPLES", "500")), deadline=None, derandomize=True, database=None, phases=[Phase.explicit, Phase.generate],
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(which=st.lists(st.sampled_from(["S0", "V0", "kappa", "theta", "sigma", "rho"]), min_size=1, max_size=2, unique=True),
       vals=st.lists(weird, min_size=2, max_size=2), dyn=st.sampled_from([0, 1, 5]),
       strike=weird, T=weird, r=weird, cp=st.sampled_from([1.0, -1.0, 0.0]), alpha=st.sampled_from([1.0, 0.0, -1.0]),
       bound=st.sampled_from([32.0, 0.0, float("inf")]), grad=st.booleans(), K=st.sampled_from([0, 1, 3]))
def test_hostile_scalars_carr_madan_basket(hhlib, which, vals, dyn, strike, T, r, cp, alpha, bound, grad, K):
    prm = dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7)
    for k, v in zip(which, vals):
        prm[k] = v
    m = _ffi.make_model(**prm)
    n = max(K, 1)
    strikes, cps = np.full(n, strike), np.full(n, cp)
    Ts, rs, Ds = np.full(n, T), np.full(n, r), np.full(n, 0.97)
    out, g = np.zeros(n), np.zeros((n, _ffi.HH_CM_GRAD_LEN))
    args = (hhlib.handle, C.byref(m), dyn, 0, alpha, bound, strikes.ctypes.data, cps.ctypes.data, Ts.ctypes.data,
            rs.ctypes.data, Ds.ctypes.data, K, out.ctypes.data)
    _trace("cm", prm, dyn, strike, T, r, cp, alpha, bound, grad, K)
    rc = hhlib.lib.hh_carr_madan_basket_grad(*args, g.ctypes.data) if grad else hhlib.lib.hh_carr_madan_basket(*args)
    assert rc in (_ffi.HH_OK, _ffi.HH_ERR_INVALID, _ffi.HH_ERR_UNSUPPORTED), rc


@settings(max_examples=int(os.environ.get("HH_FUZZ_EXAMPLES", "500")), deadline=None, derandomize=True, database=None, phases=[Phase.explicit, Phase.generate],
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(which=st.lists(st.sampled_from(sorted(sane)), min_size=1, max_size=3, unique=True), vals=st.lists(weird, min_size=3, max_size=3),
       n_paths=st.sampled_from([0, 1, 200]), n_steps=st.sampled_from([0, 1, 5]))
def test_hostile_scalars_exact_grid(hhlib, which, vals, n_paths, n_steps):
    prm = dict(sane)
    for k, v in zip(which, vals):
        prm[k] = v
    m = _ffi.make_model(discount=1.0, **prm)
    c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n_paths, n_steps,
                         seeds=np.arange(1, max(n_paths, 1) + 1, dtype=np.uint64))
    res = _ffi.hh_result()
    spot = np.zeros((max(n_steps, 1) + 1) * max(n_paths, 1))
    _trace("grid", prm, n_paths, n_steps)
    rc = hhlib.lib.hh_heston_exact_grid(hhlib.handle, C.byref(m), C.byref(c), spot.ctypes.data, None, 0, C.byref(res))
    assert rc in (_ffi.HH_OK, _ffi.HH_ERR_INVALID, _ffi.HH_ERR_UNSUPPORTED), (rc, prm)
    if rc == _ffi.HH_OK and all(1e-3 <= abs(prm[k]) <= 10.0 for k in ("V0", "kappa", "theta", "sigma", "T")) \
            and abs(prm["r"]) <= 1.0 and 1e-3 <= prm["S0"] <= 1e6:
        assert np.all(np.isfinite(spot)) and np.all(spot > 0)  # moderate inputs: moderate outputs


@settings(max_examples=int(os.environ.get("HH_FUZZ_EXAMPLES", "500")), deadline=None, derandomize=True, database=None,
          phases=[Phase.explicit, Phase.generate],
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(strikes=st.lists(weird, min_size=1, max_size=6), cps=st.lists(st.sampled_from([1.0, -1.0, 0.0, 2.0]), min_size=6, max_size=6),
       dynamics=st.sampled_from([0, 1]), strategy=st.sampled_from([0, 1, 2]), anti=st.sampled_from([0, 1]),
       n_paths=st.sampled_from([1, 100, 4097]), K=st.sampled_from([0, 1, 5, 6]))
def test_hostile_scalars_basket(hhlib, strikes, cps, dynamics, strategy, anti, n_paths, K):
    strikes = np.array((strikes * 6)[:6], dtype=np.float64)
    cps = np.array(cps, dtype=np.float64)
    m = _ffi.make_model()
    c = _ffi.make_config(dynamics, strategy, n_paths, 5, antithetic=anti,
                         seeds=np.arange(1, n_paths + 1, dtype=np.uint64))
    res = (_ffi.hh_result * 6)()
    _trace("basket", list(strikes), list(cps), dynamics, strategy, anti, n_paths, K)
    rc = hhlib.lib.hh_mc_solve_basket(hhlib.handle, C.byref(m), C.byref(c), strikes.ctypes.data, cps.ctypes.data,
                                      K, res, None)
    assert rc in (_ffi.HH_OK, _ffi.HH_ERR_INVALID, _ffi.HH_ERR_UNSUPPORTED), rc


@settings(max_examples=int(os.environ.get("HH_FUZZ_EXAMPLES", "500")), deadline=None, derandomize=True, database=None,
          phases=[Phase.explicit, Phase.generate],
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(dynamics=st.sampled_from([0, 1]), strategy=st.sampled_from([0, 1, 2]), layout=st.sampled_from([0, 1, 4]),
       n_paths=st.sampled_from([1, 255, 257, 700]), n_steps=st.sampled_from([1, 3, 9]), anti=st.sampled_from([0, 1]),
       short=st.sampled_from([0, 1, 17]), fill=st.sampled_from([0.01, float("nan"), 1e300, -1e300]))
def test_replay_buffers_of_every_shape(hhlib, dynamics, strategy, layout, n_paths, n_steps, anti, short, fill):
    """REPLAY with buffers of the right size, and `short` elements too small (declared through replay_len):
    accepted or refused on the host, never read out of bounds on the device."""
    nc = 2 if dynamics == 1 else 1
    if strategy == 0:
        need = n_paths * n_steps * nc if layout == 1 else ((n_paths + 255) // 256) * n_steps * nc * 256
    elif strategy == 2:
        need = 3 * n_paths
    else:
        need = n_paths
    buf = np.full(max(need - short, 1), fill)
    m = _ffi.make_model()
    c = _ffi.make_config(dynamics, strategy, n_paths, n_steps, antithetic=anti, noise_mode=_ffi.HH_NOISE_REPLAY,
                         replay=buf, replay_layout=layout)
    res = _ffi.hh_result()
    _trace("replay", dynamics, strategy, layout, n_paths, n_steps, anti, short, fill)
    rc = hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(m), C.byref(c), C.byref(res), None)
    assert rc in (_ffi.HH_OK, _ffi.HH_ERR_INVALID, _ffi.HH_ERR_UNSUPPORTED), rc
    if short and need - short >= 1 and rc == _ffi.HH_OK:
        raise AssertionError("a replay buffer shorter than the kernels index was accepted")
