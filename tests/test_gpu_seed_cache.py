"""hh_seeds_cache: a SimulationConfig's seed vector kept in device memory by the context.  The reference reads
seeds[i] at EVERY solve (montecarlo.jl:331), so a cached copy must never outlive a change of the vector: the
cache is content-addressed (length + a fingerprint of every element), bounded, and freed with the context."""
import ctypes as C

import numpy as np
import pytest

import hedgehog_jl_amd as hh
from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o

pytestmark = pytest.mark.gpu

HES, EM = _ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA


def price_with(ctx, seeds_dev, n, steps=6):
    m = o.make_model()
    c = o.make_config(HES, EM, n, steps)
    c.seeds, c.seeds_on_device, c.seeds_len = seeds_dev, 1, n
    r = _ffi.hh_result()
    ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(r), None))
    return r.price


def price_host(ctx, seeds, steps=6):
    m = o.make_model()
    c = o.make_config(HES, EM, seeds.size, steps, seeds=seeds)
    r = _ffi.hh_result()
    ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(r), None))
    return r.price


def test_hit_miss_and_a_vector_changed_in_place():
    ctx = hh.Context(0)
    n = 5000
    v = np.arange(1, n + 1, dtype=np.uint64)
    s0 = ctx.seeds_cache_stats()
    d1 = ctx.seeds_on_device(v)
    assert ctx.seeds_on_device(v) == d1
    assert ctx.seeds_on_device(v.copy()) == d1  # the same numbers at another address: the same entry
    s1 = ctx.seeds_cache_stats()
    assert s1["uploads"] - s0["uploads"] == 1 and s1["hits"] - s0["hits"] == 2
    p1 = price_with(ctx, d1, n)
    assert p1 == price_host(ctx, v)
    # ONE element in the middle changes, in place: neither the ends nor any strided sample of it moved
    v[n // 2 + 1] ^= np.uint64(1)
    d2 = ctx.seeds_on_device(v)
    assert d2 != d1 and ctx.seeds_cache_stats()["uploads"] - s1["uploads"] == 1
    p2 = price_with(ctx, d2, n)
    assert p2 == price_host(ctx, v) and p2 != p1
    # a given fingerprint is trusted; 0 means "compute it"
    f = ctx.lib.hh_seeds_fingerprint(v.ctypes.data, n)
    assert f != 0 and ctx.seeds_on_device(v, f) == d2
    assert ctx.lib.hh_seeds_fingerprint(v.ctypes.data, n - 1) != f
    ctx.close()


def test_eviction_is_least_recently_used_and_bounded():
    ctx = hh.Context(0)
    vs = [np.arange(1, 1001, dtype=np.uint64) + np.uint64(1000 * k) for k in range(12)]
    first = ctx.seeds_on_device(vs[0])
    for v in vs[1:8]:
        ctx.seeds_on_device(v)
    assert ctx.seeds_on_device(vs[0]) == first       # eight entries: still there, and now the most recent
    assert ctx.seeds_cache_stats()["evictions"] == 0
    ctx.seeds_on_device(vs[8])                       # the ninth vector: vs[1] (least recently used) goes
    st = ctx.seeds_cache_stats()
    assert st["evictions"] == 1 and st["uploads"] == 9
    assert ctx.seeds_on_device(vs[0]) == first and ctx.seeds_cache_stats()["uploads"] == 9
    ctx.seeds_on_device(vs[1])
    assert ctx.seeds_cache_stats()["uploads"] == 10  # it had to come back
    for v in vs:                                      # every copy, old or re-uploaded, holds its vector
        assert price_with(ctx, ctx.seeds_on_device(v), v.size) == price_host(ctx, v)
    ctx.close()                                       # frees every entry with the context (no leak check here, no crash)


def test_bad_arguments_and_contexts_do_not_share():
    a, b = hh.Context(0), hh.Context(0)
    v = np.arange(1, 300, dtype=np.uint64)
    p = C.c_void_p()
    assert a.lib.hh_seeds_cache(a.handle, None, 5, 0, C.byref(p)) == _ffi.HH_ERR_INVALID
    assert a.lib.hh_seeds_cache(a.handle, C.c_void_p(v.ctypes.data), 0, 0, C.byref(p)) == _ffi.HH_ERR_INVALID
    assert a.lib.hh_seeds_cache(a.handle, C.c_void_p(v.ctypes.data), v.size, 0, None) == _ffi.HH_ERR_INVALID
    da, db = a.seeds_on_device(v), b.seeds_on_device(v)
    assert da != db and b.seeds_cache_stats()["uploads"] == 1
    a.close()
    assert price_with(b, db, v.size) == price_host(b, v)  # b's copy is b's
    b.close()


def test_host_mirror_goes_through_the_cache():
    prob = hh.PricingProblem(hh.VanillaOption(100.0, hh.Date(2022, 1, 1), hh.European(), hh.Call(), hh.Spot()),
                             hh.HestonInputs(hh.Date(2021, 1, 1), 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7))
    raw = np.arange(1, 4001, dtype=np.uint64)
    cfg = hh.SimulationConfig(4000, steps=8, seeds=raw)
    m = hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(), cfg)
    ctx = hh.get_context(0)
    s0 = ctx.seeds_cache_stats()
    p = [hh.solve(prob, m, ensemble=False).price for _ in range(5)]
    s1 = ctx.seeds_cache_stats()
    assert len(set(p)) == 1 and s1["uploads"] - s0["uploads"] <= 1 and s1["hits"] - s0["hits"] >= 4
    raw[:200] += np.uint64(10**6)  # the caller's array is not the config's: SimulationConfig took a frozen copy
    assert hh.solve(prob, m, ensemble=False).price == p[0]
    cfg2 = cfg.replace(seeds=raw)
    assert hh.solve(prob, hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(), cfg2), ensemble=False).price != p[0]
