"""N > 1 path: sharding + the single all-reduce, rehearsed with world_size 2 on CPU (gloo).
The per-shard kernel driver is replaced by the CPU oracle (the test is about the host logic:
ranges, seed slicing, path offsets, the collective, finalize)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hedgehog_jl_amd as hh
from hedgehog_jl_amd import _ffi


def test_shard_ranges_cover_exactly():
    for n in (1, 2, 7, 256, 1000, 10**6 + 3):
        for world in (1, 2, 3, 8):
            r = [hh.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(0 <= b - a <= -(-n // world) for a, b in r)


def _oracle_accumulate(model, cfg, device):
    from tests import oracle_ffi as o
    if cfg.n_paths == 0:
        return np.zeros(_ffi.HH_ACC_LEN)
    return o.load().mc_solve(model, cfg, want_terminal=False)[2]


def _problems():
    ref = hh.Date(2021, 1, 1)
    exp = hh.Date(2022, 1, 1)
    call = hh.VanillaOption(100.0, exp, hh.European(), hh.Call(), hh.Spot())
    hes = hh.PricingProblem(call, hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7))
    bs = hh.PricingProblem(call, hh.BlackScholesInputs(ref, 0.05, 100.0, 0.2))
    n = 3001  # odd: ragged shards
    seeds = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(2654435761)
    return [
        (hes, hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(),
                            hh.SimulationConfig(n, steps=20, seeds=seeds,
                                                variance_reduction=hh.Antithetic()))),
        (bs, hh.MonteCarlo(hh.LognormalDynamics(), hh.BlackScholesExact(),
                           hh.SimulationConfig(n, seeds=seeds))),
        (bs, hh.MonteCarlo(hh.LognormalDynamics(), hh.EulerMaruyama(),
                           hh.SimulationConfig(n, steps=7, seeds=seeds))),
        (hes, hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(),
                            hh.SimulationConfig(1, steps=3, seeds=seeds))),  # one rank gets nothing
    ]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)  # what torch.distributed.run exports
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = []
        seen = []

        def on_my_gpu(model, cfg, device):  # the shard must be driven on LOCAL_RANK's device,
            seen.append(device)             # not on MonteCarlo.device's default 0
            return _oracle_accumulate(model, cfg, device)

        sol = hh.solve_sharded(*_problems()[0], accumulate=on_my_gpu)
        assert seen == [rank] and hh.rank_device(0) == rank and hh.rank_device(0, device=5) == 5
        for prob, method in _problems():
            sol = hh.solve_sharded(prob, method, accumulate=_oracle_accumulate)
            res.append((sol.price, sol.std_error, int(sol.result.n_paths_done)))
        # Greeks ride the same all-reduce (slots 2..9)
        prob, method = _problems()[0]
        p2 = hh.set(prob, hh.optic("market_inputs.spot"), hh.Dual(100.0, (1.0, 0.0)))
        p2 = hh.set(p2, hh.optic("market_inputs.rate.rate"), hh.Dual(0.03, (0.0, 1.0)))
        g = hh.solve_sharded(p2, method, accumulate=_oracle_accumulate).price
        res.append((g.value, g.partials[0], g.partials[1]))
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_rank_sharded_solve_equals_single_process():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1]  # every rank holds the same estimator after the all-reduce

    # single-process reference (no process group): same function, world = 1
    single = []
    for prob, method in _problems():
        sol = hh.solve_sharded(prob, method, accumulate=_oracle_accumulate)
        single.append((sol.price, sol.std_error, int(sol.result.n_paths_done)))
    for (p2, s2, n2), (p1, s1, n1) in zip(got[0][:4], single):
        assert n2 == n1
        assert p2 == pytest.approx(p1, rel=1e-13)
        assert s2 == pytest.approx(s1, rel=1e-9, abs=1e-15)
    prob, method = _problems()[0]
    p2 = hh.set(prob, hh.optic("market_inputs.spot"), hh.Dual(100.0, (1.0, 0.0)))
    p2 = hh.set(p2, hh.optic("market_inputs.rate.rate"), hh.Dual(0.03, (0.0, 1.0)))
    g1 = hh.solve_sharded(p2, method, accumulate=_oracle_accumulate).price
    v, d0, d1 = got[0][4]
    assert v == pytest.approx(g1.value, rel=1e-13)
    assert d0 == pytest.approx(g1.partials[0], rel=1e-12)
    assert d1 == pytest.approx(g1.partials[1], rel=1e-12)
