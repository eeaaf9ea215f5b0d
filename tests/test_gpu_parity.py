"""Parity of the HIP path (through the C-ABI) with the CPU oracle on identical inputs.

Tolerances (fp64; the BASELINE bar is 1e-4 relative on the price):
  * REPLAY (identical Wiener increments): per-path samples 1e-12 rel, price 1e-12 rel
  * GENERATE (same Philox stream, different libm for log/sincospi): per-path 1e-11 rel
  * dual partials: 1e-10 rel on the accumulated Greeks
"""
import ctypes as C

import numpy as np
import pytest

from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o

pytestmark = pytest.mark.gpu

GBM, HES = _ffi.HH_LOGNORMAL, _ffi.HH_HESTON
EM, EXACT = _ffi.HH_EULER_MARUYAMA, _ffi.HH_EXACT_LAW
GEN, REP = _ffi.HH_NOISE_GENERATE, _ffi.HH_NOISE_REPLAY


def gpu_solve(ctx, model, cfg, want_terminal=True):
    res = _ffi.hh_result()
    n = cfg.n_paths * (2 if cfg.antithetic else 1)
    term = np.zeros(n) if want_terminal else None
    ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(model), C.byref(cfg), C.byref(res),
                                  term.ctypes.data if term is not None else None))
    return res, term


def seeds_for(n, salt=0):
    return (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
            + np.uint64(salt))


def check(res_g, term_g, res_o, term_o, P, rtol_path, rtol_price, rtol_d=1e-10):
    np.testing.assert_allclose(term_g, term_o, rtol=rtol_path, atol=0)
    assert res_g.n_paths_done == res_o.n_paths_done
    assert res_g.price == pytest.approx(res_o.price, rel=rtol_price, abs=1e-300)
    assert res_g.std_error == pytest.approx(res_o.std_error, rel=1e-8, abs=1e-14)
    for k in range(P):
        assert res_g.dprice[k] == pytest.approx(res_o.dprice[k], rel=rtol_d, abs=1e-12)


HESTON_SEEDS = {"S0": [1, 0, 0], "V0": [0, 1, 0], "r_drift": [0, 0, 1],
                "discount": [0, 0, -float(np.exp(-0.03))]}


@pytest.mark.parametrize("n_paths", [1, 255, 256, 257, 1000, 4099])
@pytest.mark.parametrize("n_steps", [1, 3, 4, 5, 8, 9, 100])
def test_heston_euler_replay_ragged(hhlib, oracle, n_paths, n_steps):
    seeds = seeds_for(n_paths)
    dW = oracle.wiener_fill(HES, -0.7, 1.0, n_steps, seeds)
    m = o.make_model()
    c = o.make_config(HES, EM, n_paths, n_steps, noise_mode=REP, replay=dW)
    rg, tg = gpu_solve(hhlib, m, c)
    ro, to, _ = oracle.mc_solve(m, c)
    check(rg, tg, ro, to, 0, 1e-12, 1e-12)


@pytest.mark.parametrize("dyn", [GBM, HES])
@pytest.mark.parametrize("n_steps", [1, 2, 5, 9])
def test_replay_ragged_above_the_small_grid_threshold(hhlib, oracle, dyn, n_steps):
    """A grid of more than 512 workgroups (beyond the chip's resident set: workgroups start and leave
    while others run) with a ragged last tile, odd step counts and runs shorter than one pipeline
    chunk: the register pipeline's guarded head and tail (test_heston_euler_replay_ragged covers the
    small grids)."""
    n_paths = 512 * 256 + 4099
    seeds = seeds_for(n_paths, 9)
    m = o.make_model(sigma=0.2 if dyn == GBM else 0.3)
    dW = oracle.wiener_fill(dyn, m.rho, m.T, n_steps, seeds)
    c = o.make_config(dyn, EM, n_paths, n_steps, noise_mode=REP, replay=dW)
    rg, tg = gpu_solve(hhlib, m, c)
    ro, to, _ = oracle.mc_solve(m, c)
    check(rg, tg, ro, to, 0, 1e-12, 1e-12)


# the step form only matters where the diffusion depends on the state: (GBM, classic) is not a case
@pytest.mark.parametrize("dyn,split", [(GBM, 1), (HES, 1), (HES, 0)])
@pytest.mark.parametrize("anti", [0, 1])
@pytest.mark.parametrize("noise", [GEN, REP])
@pytest.mark.parametrize("P", [0, 1, 3, 5])
def test_euler_matrix(hhlib, oracle, dyn, anti, noise, P, split):
    n_paths, n_steps = 3000, 50
    seeds = seeds_for(n_paths, 7)
    names = ["S0", "sigma", "r_drift", "strike", "V0"] if dyn == GBM else \
        ["S0", "V0", "r_drift", "kappa", "sigma"]
    sd = {}
    for k in range(P):
        v = [0.0] * P
        v[k] = 1.0
        sd[names[k]] = v
    if P >= 3:  # the rate also moves the discount factor
        sd["discount"] = [0.0, 0.0, -float(np.exp(-0.03))] + [0.0] * (P - 3)
    m = o.make_model(sigma=0.2 if dyn == GBM else 0.3, seeds=sd, n_partials=P)
    rep = oracle.wiener_fill(dyn, m.rho, m.T, n_steps, seeds) if noise == REP else None
    c = o.make_config(dyn, EM, n_paths, n_steps, antithetic=anti, em_split=split, noise_mode=noise,
                      seeds=seeds, replay=rep, n_partials=P)
    rg, tg = gpu_solve(hhlib, m, c)
    ro, to, _ = oracle.mc_solve(m, c)
    check(rg, tg, ro, to, P, 1e-12 if noise == REP else 1e-11, 1e-12 if noise == REP else 1e-11)


@pytest.mark.parametrize("cp", [1.0, -1.0])
def test_put_and_call_q2_parameters(hhlib, oracle, cp):
    """The parameter set the reference's own test actually runs: V0=1.5, κ=0.04, θ=0.3,
    σ=-0.6 (negative vol-of-vol), ρ=0.04, r=0.05, T=364/365."""
    n_paths, n_steps = 2000, 200
    seeds = seeds_for(n_paths, 3)
    m = o.make_model(S0=100, V0=1.5, kappa=0.04, theta=0.3, sigma=-0.6, rho=0.04, r=0.05,
                     T=364 / 365, strike=100, cp=cp)
    c = o.make_config(HES, EM, n_paths, n_steps, antithetic=1, seeds=seeds)
    rg, tg = gpu_solve(hhlib, m, c)
    ro, to, _ = oracle.mc_solve(m, c)
    check(rg, tg, ro, to, 0, 1e-10, 1e-11)


@pytest.mark.parametrize("anti", [0, 1])
@pytest.mark.parametrize("P", [0, 3])
@pytest.mark.parametrize("offset", [0, 1, 12345])
@pytest.mark.parametrize("compat", [0, 1])
def test_exact_lognormal(hhlib, oracle, anti, P, offset, compat):
    n_paths = 5001
    sd = {"S0": [1, 0, 0], "sigma": [0, 1, 0], "r_drift": [0, 0, 1],
          "discount": [0, 0, -float(366 / 365 * np.exp(-0.03 * 366 / 365))]} if P else {}
    m = o.make_model(S0=1.0, sigma=1.0, r=0.03, T=366 / 365, strike=1.0, seeds=sd, n_partials=P)
    c = o.make_config(GBM, EXACT, n_paths, antithetic=anti, seeds=[42], n_partials=P,
                      path_offset=offset, compat_sqrt_alpha=compat)
    rg, tg = gpu_solve(hhlib, m, c)
    ro, to, _ = oracle.mc_solve(m, c)
    check(rg, tg, ro, to, P, 1e-12, 1e-12)


def test_exact_lognormal_replay_normals(hhlib, oracle):
    n = 777
    z = np.random.default_rng(5).standard_normal(n)
    m = o.make_model(S0=100, sigma=0.2, r=0.05, T=1.0, strike=90)
    c = o.make_config(GBM, EXACT, n, antithetic=1, noise_mode=REP, replay=z)
    rg, tg = gpu_solve(hhlib, m, c)
    ro, to, _ = oracle.mc_solve(m, c)
    check(rg, tg, ro, to, 0, 1e-13, 1e-13)


def test_exact_sharding_is_invisible(hhlib):
    """Exact laws draw by GLOBAL trajectory index from one key (montecarlo.jl:456), so two shards
    reproduce the single-shard samples exactly."""
    n = 6000
    m = o.make_model(S0=100, sigma=0.2, r=0.05, T=1.0, strike=100)
    full = gpu_solve(hhlib, m, o.make_config(GBM, EXACT, n, seeds=[9]))[1]
    a = gpu_solve(hhlib, m, o.make_config(GBM, EXACT, 2501, seeds=[9]))[1]
    b = gpu_solve(hhlib, m, o.make_config(GBM, EXACT, n - 2501, seeds=[9], path_offset=2501))[1]
    np.testing.assert_array_equal(full, np.concatenate([a, b]))


@pytest.mark.parametrize("dyn", [GBM, HES])
@pytest.mark.parametrize("n_paths,n_steps", [(1, 1), (300, 7), (1025, 33)])
def test_wiener_fill_matches_oracle(hhlib, oracle, dyn, n_paths, n_steps):
    import torch
    seeds = seeds_for(n_paths, 11)
    want = oracle.wiener_fill(dyn, -0.7, 1.5, n_steps, seeds)
    n = hhlib.lib.hh_replay_elems(n_paths, n_steps, dyn)
    assert n == want.size
    buf = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    hhlib.check(hhlib.lib.hh_wiener_fill(hhlib.handle, dyn, -0.7, 1.5, n_steps, n_paths,
                                         seeds.ctypes.data, 0, buf.data_ptr()))
    hhlib.synchronize()
    np.testing.assert_allclose(buf.cpu().numpy(), want, rtol=0, atol=2e-15)


@pytest.mark.parametrize("dyn", [GBM, HES])
@pytest.mark.parametrize("n_paths,n_steps", [(1, 1), (300, 7), (1025, 33)])
def test_replay_pack_bit_exact(hhlib, oracle, dyn, n_paths, n_steps):
    import torch
    nc = 2 if dyn == HES else 1
    src = np.random.default_rng(1).standard_normal((n_paths, n_steps, nc))
    want = oracle.replay_pack(dyn, n_paths, n_steps, src)
    buf = torch.full((want.size,), float("nan"), dtype=torch.float64, device="cuda")
    hhlib.check(hhlib.lib.hh_replay_pack(hhlib.handle, dyn, n_paths, n_steps, src.ctypes.data, 0,
                                         buf.data_ptr()))
    hhlib.synchronize()
    np.testing.assert_array_equal(buf.cpu().numpy(), want)


def test_path_major_replay_equals_tile_major(hhlib, oracle):
    n_paths, n_steps = 700, 12
    src = np.random.default_rng(2).standard_normal((n_paths, n_steps, 2)) * 0.05
    m = o.make_model()
    c1 = o.make_config(HES, EM, n_paths, n_steps, noise_mode=REP, replay=src,
                       replay_layout=_ffi.HH_REPLAY_PATH_MAJOR)
    c2 = o.make_config(HES, EM, n_paths, n_steps, noise_mode=REP,
                       replay=oracle.replay_pack(HES, n_paths, n_steps, src))
    r1, t1 = gpu_solve(hhlib, m, c1)
    r2, t2 = gpu_solve(hhlib, m, c2)
    np.testing.assert_array_equal(t1, t2)
    assert r1.price == r2.price
    ro, to, _ = oracle.mc_solve(m, c1)
    np.testing.assert_allclose(t1, to, rtol=1e-12)


def _path_major(oracle, dyn, n_paths, n_steps, seeds, rho=-0.7, T=1.0):
    """the oracle's increments in the reference's own layout dW[path][step][comp] + the tile-major form"""
    nc = 2 if dyn == HES else 1
    tiled = oracle.wiener_fill(dyn, rho, T, n_steps, seeds)
    pm = tiled.reshape(-1, n_steps, nc, 256).transpose(0, 3, 1, 2).reshape(-1, n_steps, nc)[:n_paths]
    return np.ascontiguousarray(pm), tiled


@pytest.mark.parametrize("dyn", [GBM, HES])
@pytest.mark.parametrize("n_paths", [1, 63, 64, 65, 255, 256, 257, 1000, 4099])
@pytest.mark.parametrize("n_steps", [1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 100, 252])
def test_path_major_replay_ragged(hhlib, oracle, dyn, n_paths, n_steps):
    """The reference's noise layout streamed as it stands (euler_pm_kernel): rows of every length —
    shorter than one 128-byte line, not a multiple of it, starting at every phase of a line — and the
    lognormal rows of an odd number of steps, which still take the repack route.  Same samples and
    sums as the tile-major kernel, bit for bit; oracle parity at the REPLAY bar."""
    seeds = seeds_for(n_paths, 21)
    m = o.make_model(sigma=0.2 if dyn == GBM else 0.3)
    pm, tiled = _path_major(oracle, dyn, n_paths, n_steps, seeds, m.rho, m.T)
    c1 = o.make_config(dyn, EM, n_paths, n_steps, noise_mode=REP, replay=pm,
                       replay_layout=_ffi.HH_REPLAY_PATH_MAJOR)
    c2 = o.make_config(dyn, EM, n_paths, n_steps, noise_mode=REP, replay=tiled)
    r1, t1 = gpu_solve(hhlib, m, c1)
    r2, t2 = gpu_solve(hhlib, m, c2)
    np.testing.assert_array_equal(t1, t2)
    assert (r1.price, r1.sum_payoff, r1.sumsq_payoff) == (r2.price, r2.sum_payoff, r2.sumsq_payoff)
    ro, to, _ = oracle.mc_solve(m, c2)
    check(r1, t1, ro, to, 0, 1e-12, 1e-12)


@pytest.mark.parametrize("dyn,split", [(GBM, 1), (HES, 1), (HES, 0)])
@pytest.mark.parametrize("anti", [0, 1])
@pytest.mark.parametrize("duals", [0, 1, 3])
def test_path_major_replay_variants(hhlib, oracle, dyn, split, anti, duals):
    n_paths, n_steps = 1500, 38
    seeds = seeds_for(n_paths, 22)
    sd = {0: None, 1: {"V0": [1.0]} if dyn == HES else {"sigma": [1.0]},
          3: HESTON_SEEDS if dyn == HES else {"S0": [1, 0, 0], "sigma": [0, 1, 0], "r_drift": [0, 0, 1],
                                              "discount": [0, 0, -float(np.exp(-0.03))]}}[duals]
    m = o.make_model(sigma=0.2 if dyn == GBM else 0.3, seeds=sd, n_partials=duals)
    pm, tiled = _path_major(oracle, dyn, n_paths, n_steps, seeds, m.rho, m.T)
    kw = dict(antithetic=anti, em_split=split, noise_mode=REP, n_partials=duals)
    c1 = o.make_config(dyn, EM, n_paths, n_steps, replay=pm, replay_layout=_ffi.HH_REPLAY_PATH_MAJOR, **kw)
    c2 = o.make_config(dyn, EM, n_paths, n_steps, replay=tiled, **kw)
    r1, t1 = gpu_solve(hhlib, m, c1)
    r2, t2 = gpu_solve(hhlib, m, c2)
    np.testing.assert_array_equal(t1, t2)
    assert r1.price == r2.price and [r1.dprice[k] for k in range(duals)] == [r2.dprice[k] for k in range(duals)]
    ro, to, _ = oracle.mc_solve(m, c2)
    check(r1, t1, ro, to, duals, 1e-12, 1e-12)


@pytest.mark.parametrize("shift", [1, 2, 3, 5, 7])
def test_path_major_replay_from_any_16_byte_boundary(hhlib, oracle, shift):
    """Device-resident rows that start `shift` pieces into a 128-byte line (a slice of a larger
    buffer): the loader clamps what lies before the first and behind the last row."""
    import torch
    n_paths, n_steps = 777, 45
    seeds = seeds_for(n_paths, 23)
    m = o.make_model()
    pm, tiled = _path_major(oracle, HES, n_paths, n_steps, seeds, m.rho, m.T)
    buf = torch.full((pm.size + 2 * shift + 64,), float("nan"), dtype=torch.float64, device="cuda")
    buf[2 * shift:2 * shift + pm.size] = torch.from_numpy(pm.ravel()).cuda()
    c1 = o.make_config(HES, EM, n_paths, n_steps, noise_mode=REP, replay_layout=_ffi.HH_REPLAY_PATH_MAJOR)
    c1.replay, c1.replay_on_device, c1.replay_len = buf.data_ptr() + 16 * shift, 1, pm.size
    c2 = o.make_config(HES, EM, n_paths, n_steps, noise_mode=REP, replay=tiled)
    r1, t1 = gpu_solve(hhlib, m, c1)
    r2, t2 = gpu_solve(hhlib, m, c2)
    np.testing.assert_array_equal(t1, t2)
    assert r1.price == r2.price


def test_device_resident_inputs(hhlib, oracle):
    """seeds / replay / terminal living in HBM (the bench configuration)."""
    import torch
    n_paths, n_steps = 2048 + 17, 20
    seeds = seeds_for(n_paths, 5)
    m = o.make_model()
    c = o.make_config(HES, EM, n_paths, n_steps, seeds=seeds)
    ro, to, _ = oracle.mc_solve(m, c)
    d_seeds = torch.from_numpy(seeds.view(np.int64)).cuda()
    d_term = torch.zeros(n_paths, dtype=torch.float64, device="cuda")
    c.seeds = d_seeds.data_ptr()
    c.seeds_on_device = 1
    c.terminal_on_device = 1
    res = _ffi.hh_result()
    hhlib.check(hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(m), C.byref(c), C.byref(res),
                                      d_term.data_ptr()))
    np.testing.assert_allclose(d_term.cpu().numpy(), to, rtol=1e-11)
    # same draws through a device-resident REPLAY buffer
    n = hhlib.lib.hh_replay_elems(n_paths, n_steps, HES)
    d_rep = torch.empty(n, dtype=torch.float64, device="cuda")
    hhlib.check(hhlib.lib.hh_wiener_fill(hhlib.handle, HES, m.rho, m.T, n_steps, n_paths,
                                         d_seeds.data_ptr(), 1, d_rep.data_ptr()))
    c.noise_mode, c.replay, c.replay_on_device = REP, d_rep.data_ptr(), 1
    res2 = _ffi.hh_result()
    hhlib.check(hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(m), C.byref(c), C.byref(res2),
                                      d_term.data_ptr()))
    assert res2.price == pytest.approx(res.price, rel=1e-13)
    assert res.price == pytest.approx(ro.price, rel=1e-11)


def test_error_codes(hhlib):
    m = o.make_model()
    res = _ffi.hh_result()
    lib, h = hhlib.lib, hhlib.handle

    def rc(cfg, model=m):
        return lib.hh_mc_solve(h, C.byref(model), C.byref(cfg), C.byref(res), None)

    assert rc(o.make_config(HES, EM, 0, 10, seeds=[1])) == _ffi.HH_ERR_INVALID
    assert rc(o.make_config(HES, EM, 10, 0, seeds=np.arange(10))) == _ffi.HH_ERR_INVALID
    assert rc(o.make_config(HES, EXACT, 10, seeds=[1])) == _ffi.HH_ERR_UNSUPPORTED
    assert rc(o.make_config(GBM, _ffi.HH_BROADIE_KAYA, 10, seeds=[1])) == _ffi.HH_ERR_UNSUPPORTED
    assert rc(o.make_config(HES, _ffi.HH_BROADIE_KAYA, 10, antithetic=1, seeds=[1])) == \
        _ffi.HH_ERR_UNSUPPORTED
    assert rc(o.make_config(HES, EM, 10, 5)) == _ffi.HH_ERR_INVALID          # no seeds
    assert rc(o.make_config(HES, EM, 10, 5, noise_mode=REP)) == _ffi.HH_ERR_INVALID  # no replay
    assert b"replay" in lib.hh_last_error(h)
    bad = o.make_model(S0=-1.0)
    assert rc(o.make_config(HES, EM, 10, 5, seeds=np.arange(10)), bad) == _ffi.HH_ERR_INVALID
    # operand-shape checks: short buffers are rejected on the host, never indexed on the device
    assert rc(o.make_config(HES, EM, 300, 5, seeds=np.arange(299))) == _ffi.HH_ERR_INVALID
    assert b"seeds" in lib.hh_last_error(h)
    short = np.zeros(2 * 5 * 2 * 256 - 1)
    assert rc(o.make_config(HES, EM, 300, 5, noise_mode=REP, replay=short)) == _ffi.HH_ERR_INVALID
    assert rc(o.make_config(HES, EM, 300, 5, noise_mode=REP, replay=np.zeros(300 * 5 * 2 - 1),
                            replay_layout=_ffi.HH_REPLAY_PATH_MAJOR)) == _ffi.HH_ERR_INVALID
    assert rc(o.make_config(GBM, EXACT, 300, noise_mode=REP, replay=np.zeros(299))) == \
        _ffi.HH_ERR_INVALID


def test_timing_hooks(hhlib):
    m = o.make_model()
    c = o.make_config(HES, EM, 4096, 16, seeds=seeds_for(4096))
    hhlib.enable_timing(True)
    for _ in range(3):
        gpu_solve(hhlib, m, c, want_terminal=False)
    t = hhlib.read_timings()
    hhlib.enable_timing(False)
    assert len(t) == 3 and all(0 < x < 50 for x in t)
    assert hhlib.read_timings() == []


def test_ten_million_paths_shards_add_up(hhlib):
    """Largest BASELINE size (10^7 trajectories, the 1->8 GPU scaling size) in GENERATE mode:
    eight path shards' accumulators add up to the single launch (what the all-reduce computes)."""
    import torch
    n, steps, G = 10_000_000, 252, 8
    seeds = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
    m = o.make_model()

    def run(n_paths, off):
        c = o.make_config(HES, EM, n_paths, steps)
        c.seeds, c.seeds_on_device = seeds.data_ptr() + 8 * off, 1
        return gpu_solve(hhlib, m, c, want_terminal=False)[0]

    full = run(n, 0)
    per = n // G  # 1_250_000: not a multiple of 256, so shard tiles differ from the full run's
    parts = [run(per, g * per) for g in range(G)]
    assert sum(p.sum_payoff for p in parts) == pytest.approx(full.sum_payoff, rel=1e-12)
    assert sum(p.sumsq_payoff for p in parts) == pytest.approx(full.sumsq_payoff, rel=1e-12)
    assert sum(p.n_paths_done for p in parts) == full.n_paths_done == n
    assert full.price == pytest.approx(9.242521073959068, abs=4 * full.std_error + 0.02)


def test_ten_million_paths_replay_equals_generate(hhlib):
    """10^7 x 252 in REPLAY mode: a 40.3 GB increment buffer resident in HBM (sized for the 288 GB
    part), streamed once; same draws as GENERATE, so the same accumulators — plain and antithetic —
    and linearity in the payoff sign: call − put = Σ(S_T − K) on identical paths."""
    import torch
    n, steps = 10_000_000, 252
    seeds = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
    m = o.make_model()
    rep = torch.empty(hhlib.lib.hh_replay_elems(n, steps, HES), dtype=torch.float64, device="cuda")
    assert rep.numel() * 8 > 40e9
    hhlib.check(hhlib.lib.hh_wiener_fill(hhlib.handle, HES, m.rho, m.T, steps, n, seeds.data_ptr(), 1,
                                         rep.data_ptr()))
    term = torch.empty(n, dtype=torch.float64, device="cuda")

    def run(noise, anti=0, model=m, terminal=None):
        c = o.make_config(HES, EM, n, steps, antithetic=anti, noise_mode=noise)
        c.seeds, c.seeds_on_device = seeds.data_ptr(), 1
        c.replay, c.replay_on_device, c.replay_len = rep.data_ptr(), 1, rep.numel()
        c.terminal_on_device = 1
        res = _ffi.hh_result()
        hhlib.check(hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(model), C.byref(c), C.byref(res),
                                          terminal.data_ptr() if terminal is not None else None))
        return res

    gen, rp = run(0), run(REP, terminal=term)
    assert rp.sum_payoff == pytest.approx(gen.sum_payoff, rel=1e-13)
    assert rp.sumsq_payoff == pytest.approx(gen.sumsq_payoff, rel=1e-13)
    ga, ra = run(0, anti=1), run(REP, anti=1)
    assert ra.sum_payoff == pytest.approx(ga.sum_payoff, rel=1e-13)
    assert ra.std_error < 0.6 * rp.std_error  # variance reduction (montecarlo_heston.jl:126)
    put = run(REP, model=o.make_model(cp=-1.0))
    assert rp.sum_payoff - put.sum_payoff == pytest.approx(float((term - m.strike).sum().item()),
                                                           rel=1e-11)
    assert rp.price == pytest.approx(9.242521073959068, abs=4 * rp.std_error + 0.02)
    print(f"REPLAY 1e7 x 252: {rp.kernel_ms:.3f} ms = "
          f"{16e-9 * n * steps / rp.kernel_ms * 1e3:.0f} GB/s")
    del rep, term
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dyn,strategy", [(HES, EM), (GBM, EM), (GBM, EXACT)])
@pytest.mark.parametrize("anti", [0, 1])
def test_mixed_active_and_passive_directions(hhlib, oracle, dyn, strategy, anti):
    """Directions that never reach the diffusion (spot, rate, strike) are finished in closed form
    by the record reduction; directions that do (V0, κ, θ, σ) are carried per path; a direction
    may mix both kinds of seed.  All must equal the oracle's step-by-step dual propagation."""
    n, steps, P = 4000 + 11, 33, 7
    D = float(np.exp(-0.03))
    sd = {
        "S0": [1, 0, 0, 0, 0.5, 0, 0],
        "V0": [0, 1, 0, 0, 0.25, 0, 0],
        "r_drift": [0, 0, 1, 0, 0, 0.1, 0],
        "discount": [0, 0, -D, 0, 0, -0.1 * D, 0],
        "strike": [0, 0, 0, 1, 0, 0.3, 0],
        "sigma": [0, 0, 0, 0, 0, 0, 1],
        "kappa": [0, 0, 0, 0, 0.1, 0, 0],
    }
    m = o.make_model(sigma=0.2 if dyn == GBM else 0.3, seeds=sd, n_partials=P)
    c = o.make_config(dyn, strategy, n, steps, antithetic=anti, seeds=seeds_for(n, 3) if
                      strategy == EM else [77], n_partials=P)
    rg, tg = gpu_solve(hhlib, m, c)
    ro, to, _ = oracle.mc_solve(m, c)
    check(rg, tg, ro, to, P, 1e-11, 1e-11, rtol_d=1e-9)
    assert rg.bk_newton_fail == 0 and rg.bk_bisect_fallback == 0  # internal slots stay internal


def test_reference_replay_exchange_format(capsys):
    """tools/check_reference_replay.py on the committed format self-test (oracle-generated, Euler cases
    with em_split=1): every case of julia/parity_replay.jl's manifest — Euler Heston / lognormal,
    antithetic, AD Greeks, exact law, Broadie–Kaya draws, LSM grid — goes through its REPLAY seam of
    the C-ABI and meets its bar; for the Heston Euler case the other step form does not."""
    import importlib.util
    import os

    from tests.conftest import ROOT
    spec = importlib.util.spec_from_file_location(
        "check_reference_replay", os.path.join(ROOT, "tools", "check_reference_replay.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.main(os.path.join(ROOT, "tests", "golden", "replay_selftest", "manifest.json"))
    out = capsys.readouterr().out
    lines = {ln.split()[1]: ln for ln in out.splitlines() if ln.startswith("case ")}
    assert bad == 0, out
    assert set(lines) == {"em_split_probe", "heston_euler", "heston_euler_antithetic", "heston_euler_greeks",
                          "lognormal_euler", "exact_lognormal", "broadie_kaya", "lsm_put"}
    assert out.index("case em_split_probe") < out.index("case heston_euler ")  # the probe comes first …
    # … and decides in one line (the fixtures were written with em_split = 1)
    assert [ln for ln in out.splitlines() if ln.startswith("VERDICT")] and \
        "VERDICT em_split = 1 matches the reference (em_split_probe" in out
    assert all(" OK: " in ln for ln in lines.values()), out
    h = lines["heston_euler"]
    e1 = float(h.split("em_split=1=max_rel_S=")[1].split(",")[0])
    e0 = float(h.split("em_split=0=max_rel_S=")[1].split(",")[0])
    assert e1 < 1e-12 and e0 > 1e-6
    assert "same_stopping_time=1.00000" in lines["lsm_put"] or "same_stopping_time=0.99" in lines["lsm_put"]


def test_launch_limits_are_argument_errors(hhlib):
    """Sizes beyond what one launch can address (2^32 threads, grid.y 65535) come back as
    HH_ERR_INVALID with a message, not as an opaque launch failure."""
    m = o.make_model()
    res = _ffi.hh_result()
    c = o.make_config(HES, EM, 2**32, 10, seeds=[1])
    assert hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(m), C.byref(c), C.byref(res), None) == _ffi.HH_ERR_INVALID
    assert b"n_paths too large" in hhlib.lib.hh_last_error(hhlib.handle)
    c = o.make_config(HES, EM, 1000, 4 * 65535 + 1, seeds=seeds_for(1000, 0))
    assert hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(m), C.byref(c), C.byref(res), None) == _ffi.HH_ERR_INVALID
    assert b"n_steps too large" in hhlib.lib.hh_last_error(hhlib.handle)
    lres = _ffi.hh_lsm_result()
    c = o.make_config(0, 1, 1000, 65535, seeds=seeds_for(1000, 0))
    assert hhlib.lib.hh_lsm_solve(hhlib.handle, C.byref(m), C.byref(c), 3, 0.999, C.byref(lres), None, None,
                                  None) == _ffi.HH_ERR_INVALID


def test_stream_switch_orders_queued_work(hhlib, oracle):
    """hh_ctx_set_stream / reset_stream between two asynchronous accumulates: the second runs on
    another stream but reuses the ctx's scratch records — it must wait for the first."""
    import torch
    dev = torch.device("cuda", 0)
    m = o.make_model()
    side = torch.cuda.Stream(dev)
    accs = torch.zeros(2, _ffi.HH_ACC_LEN, dtype=torch.float64, device=dev)
    cfgs = [o.make_config(HES, EM, 200_000, 120, seeds=seeds_for(200_000, 3)),
            o.make_config(HES, EM, 777, 5, seeds=seeds_for(777, 4))]
    try:
        hhlib.check(hhlib.lib.hh_mc_accumulate(hhlib.handle, C.byref(m), C.byref(cfgs[0]), accs[0].data_ptr(), None))
        hhlib.set_stream(side.cuda_stream)   # the long solve is still running on the ctx's own stream
        hhlib.check(hhlib.lib.hh_mc_accumulate(hhlib.handle, C.byref(m), C.byref(cfgs[1]), accs[1].data_ptr(), None))
        hhlib.synchronize()
        torch.cuda.synchronize(dev)
    finally:
        hhlib.set_stream(None)
    for k in (0, 1):
        want = oracle.mc_solve(m, cfgs[k], want_terminal=False)[2]
        np.testing.assert_allclose(accs[k].cpu().numpy()[:2], want[:2], rtol=1e-11)


def test_context_shared_between_threads(hhlib, oracle):
    """Entry points serialise on the ctx mutex: concurrent solves from several host threads on ONE
    ctx each get their own correct result (ctypes releases the GIL during the call)."""
    import threading
    m = o.make_model()
    want = {}
    cfgs = {}
    for t in range(4):
        n = 3000 + 517 * t
        cfgs[t] = o.make_config(HES, EM, n, 20 + t, antithetic=t & 1, seeds=seeds_for(n, t))
        want[t] = oracle.mc_solve(m, cfgs[t], want_terminal=False)[0].price
    got = {}

    def work(t):
        for _ in range(10):
            got[t] = gpu_solve(hhlib, m, cfgs[t], want_terminal=False)[0].price

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for t in range(4):
        assert got[t] == pytest.approx(want[t], rel=1e-11)


def test_device_normals_against_libm_box_muller(hhlib, oracle):
    """The kernels' Box–Muller uses range-specialised -2ln(u), sqrt and sincospi (csrc/hh_rng.h);
    the oracle uses glibc's log/sqrt/sin/cos on the same Philox bits.  2 million normals, dt = 1 and
    rho = 0 so that the filled increments ARE the normals: absolute agreement to a few ulp of 1,
    tails included."""
    import torch
    n, steps = 200_000, 5
    seeds = seeds_for(n, 99)
    want = oracle.wiener_fill(HES, 0.0, float(steps), steps, seeds)
    buf = torch.empty(want.size, dtype=torch.float64, device="cuda")
    hhlib.check(hhlib.lib.hh_wiener_fill(hhlib.handle, HES, 0.0, float(steps), steps, n,
                                         seeds.ctypes.data, 0, buf.data_ptr()))
    hhlib.synchronize()
    got = buf.cpu().numpy()
    err = np.abs(got - want)
    assert err.max() < 4e-15, err.max()
    live = want != 0
    assert np.max(err[live] / np.maximum(np.abs(want[live]), 1e-3)) < 4e-15
    assert np.abs(want).max() > 5.0  # the sample does reach the tails
    z = got[live]
    assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.var() - 1) < 5 * np.sqrt(2 / z.size)


@pytest.mark.parametrize("n_steps", [7, 8, 11, 12, 13, 15, 16, 17, 23, 24, 25])
@pytest.mark.parametrize("dyn,anti,P", [(HES, 0, 0), (HES, 1, 0), (HES, 0, 1), (HES, 0, 4), (HES, 1, 3),
                                        (GBM, 0, 0), (GBM, 1, 2)])
def test_replay_pipeline_chunk_boundaries(hhlib, oracle, dyn, anti, P, n_steps):
    """The REPLAY pipeline runs full 4-step chunks with unguarded loads (two in flight) and hands the
    ragged end to guarded ones: step counts on both sides of 2, 3, 4 and 6 chunks, for every kernel
    shape (price only, antithetic, carried derivatives), ragged last tile."""
    n_paths = 256 * 3 + 101
    seeds = seeds_for(n_paths, 11)
    names = ["S0", "V0", "kappa", "theta", "sigma"] if dyn == HES else ["S0", "sigma"]
    sd = {nm: [1.0 if j == k else 0.0 for j in range(P)] for k, nm in enumerate(names[:P])} if P else {}
    m = o.make_model(sigma=0.2 if dyn == GBM else 0.3, seeds=sd, n_partials=P)
    dW = oracle.wiener_fill(dyn, m.rho, m.T, n_steps, seeds)
    c = o.make_config(dyn, EM, n_paths, n_steps, antithetic=anti, noise_mode=REP, replay=dW, n_partials=P)
    rg, tg = gpu_solve(hhlib, m, c)
    ro, to, _ = oracle.mc_solve(m, c)
    check(rg, tg, ro, to, P, 1e-12, 1e-12)
