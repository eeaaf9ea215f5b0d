"""Randomised sweep of the C-ABI: any configuration either is rejected with a documented status
code or agrees with the CPU oracle.  Small sizes, many shapes (ragged tiles, odd step counts,
random parameters incl. negative vol-of-vol and clipped variance, random dual seeds)."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o

pytestmark = pytest.mark.gpu

NAMES = ["S0", "V0", "kappa", "theta", "sigma", "r_drift", "discount", "strike"]


@st.composite
def cases(draw):
    dyn = draw(st.sampled_from([0, 1]))
    strategy = draw(st.sampled_from([0, 0, 1, 2]))
    n = draw(st.integers(min_value=1, max_value=700))
    steps = draw(st.integers(min_value=0, max_value=13))
    anti = draw(st.sampled_from([0, 1]))
    split = draw(st.sampled_from([0, 1]))
    noise = draw(st.sampled_from([0, 0, 1]))
    P = draw(st.sampled_from([0, 0, 1, 2, 3, 4, 8]))
    f = lambda lo, hi: draw(st.floats(min_value=lo, max_value=hi, allow_nan=False))
    prm = dict(S0=f(50, 150), V0=f(0.005, 0.5), kappa=f(0.05, 5.0), theta=f(0.005, 0.3),
               sigma=f(0.05, 1.2) * draw(st.sampled_from([1.0, 1.0, -1.0])), rho=f(-0.95, 0.95),
               r=f(-0.02, 0.1), T=f(0.05, 2.0), strike=f(50, 150),
               cp=draw(st.sampled_from([1.0, -1.0])))
    sd = {}
    for nm in NAMES:
        if P and draw(st.booleans()):
            sd[nm] = [draw(st.sampled_from([0.0, 1.0, -0.5])) for _ in range(P)]
    seed0 = draw(st.integers(min_value=0, max_value=2**62))
    return dyn, strategy, n, steps, anti, split, noise, P, prm, sd, seed0


@given(cases())
@settings(max_examples=120, deadline=None, derandomize=True,
          suppress_health_check=list(HealthCheck))
def test_random_configurations(hhlib, oracle, case):
    dyn, strategy, n, steps, anti, split, noise, P, prm, sd, seed0 = case
    m = o.make_model(**prm, seeds=sd, n_partials=P)
    seeds = (np.arange(n, dtype=np.uint64) * np.uint64(6364136223846793005) + np.uint64(seed0))
    rep = None
    if noise == 1:
        if strategy == 0 and steps > 0:
            rep = oracle.wiener_fill(dyn, prm["rho"], prm["T"], steps, seeds)
        else:
            rep = np.random.default_rng(seed0 % 2**32).standard_normal(max(n, 1))
    c = o.make_config(dyn, strategy, n, steps, antithetic=anti, em_split=split, noise_mode=noise,
                      seeds=seeds, replay=rep, n_partials=P)
    res = _ffi.hh_result()
    term = np.zeros(n * (2 if anti else 1))
    rc = hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(m), C.byref(c), C.byref(res), term.ctypes.data)

    valid_pair = (dyn == 0 and strategy in (0, 1)) or (dyn == 1 and strategy in (0, 2))
    if not valid_pair:
        assert rc == _ffi.HH_ERR_UNSUPPORTED
        return
    if strategy == 0 and steps == 0:
        assert rc == _ffi.HH_ERR_INVALID
        return
    if strategy == 2:
        if anti or P or noise == 1:
            assert rc == _ffi.HH_ERR_UNSUPPORTED
        else:
            assert rc == _ffi.HH_OK and np.all(np.isfinite(term)) and res.n_paths_done == n
        return
    assert rc == _ffi.HH_OK, hhlib.lib.hh_last_error(hhlib.handle)
    ro, to, _ = oracle.mc_solve(m, c)
    np.testing.assert_allclose(term, to, rtol=2e-10, atol=0)
    assert res.price == pytest.approx(ro.price, rel=1e-10, abs=1e-12)
    scale = max(abs(ro.price), 1.0)
    for k in range(P):
        assert res.dprice[k] == pytest.approx(ro.dprice[k], rel=1e-8, abs=1e-9 * scale * 100)
