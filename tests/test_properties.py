"""Property-based checks of the host-side logic (no GPU)."""
import datetime as dt
import math

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

import hedgehog_jl_amd as hh
from hedgehog_jl_amd.dual import dexp, dlog

finite = st.floats(min_value=0.1, max_value=10.0, allow_nan=False)


@given(finite, finite, finite)
@settings(max_examples=50, deadline=None, derandomize=True)
def test_dual_matches_finite_differences(a, b, c):
    def f(x, y, z):
        return dexp(-x * y) * dlog(z + x) / (y + 1.0) - (x - z) ** 3 + 2.0 / z

    duals = [hh.Dual(v, tuple(1.0 if i == k else 0.0 for i in range(3)))
             for k, v in enumerate((a, b, c))]
    out = f(*duals)
    base = [a, b, c]
    for k in range(3):
        h = 1e-6 * max(1.0, abs(base[k]))
        up, dn = list(base), list(base)
        up[k] += h
        dn[k] -= h
        fd = (f(*up) - f(*dn)) / (2 * h)
        assert math.isclose(out.partials[k], fd, rel_tol=1e-5, abs_tol=1e-6)
    assert math.isclose(out.value, f(a, b, c), rel_tol=1e-14)


@given(st.dates(min_value=dt.date(1900, 1, 1), max_value=dt.date(2200, 1, 1)),
       st.integers(min_value=0, max_value=20000))
@settings(derandomize=True)
def test_yearfrac_is_act_365(d, days):
    e = d + dt.timedelta(days=days)
    assert hh.yearfrac(d, e) == days / 365
    assert hh.to_ticks(e) - hh.to_ticks(d) == days * 86400000


@given(st.integers(min_value=1, max_value=10**7), st.integers(min_value=1, max_value=16))
@settings(derandomize=True)
def test_shards_partition_the_trajectories(n, world):
    r = [hh.shard_range(n, k, world) for k in range(world)]
    assert r[0][0] == 0 and r[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert sum(b - a for a, b in r) == n


@given(st.integers(min_value=1, max_value=700), st.integers(min_value=1, max_value=9),
       st.sampled_from([0, 1]))
@settings(max_examples=25, deadline=None, derandomize=True)
def test_replay_pack_is_a_bijection_onto_the_tile_layout(n_paths, n_steps, dyn):
    from tests import oracle_ffi as o
    orc = o.load()
    nc = 2 if dyn == 1 else 1
    src = np.arange(1, n_paths * n_steps * nc + 1, dtype=np.float64).reshape(n_paths, n_steps, nc)
    out = orc.replay_pack(dyn, n_paths, n_steps, src)
    assert out.size == orc.replay_elems(n_paths, n_steps, dyn) == \
        hh.load_library().hh_replay_elems(n_paths, n_steps, dyn)
    t = out.reshape(-1, n_steps, nc, 256)
    back = t.transpose(0, 3, 1, 2).reshape(-1, n_steps, nc)[:n_paths]
    assert np.array_equal(back, src)
    assert np.count_nonzero(out) == src.size  # padding lanes stay zero
