"""LSM on an ensemble sharded over ranks (hh_lsm_shard_*, solve_lsm_sharded): the path's exchange
steps — one SUM all-reduce between consecutive phases of the backward induction.
 * one rank: the phased sequence reproduces the fused hh_lsm_solve bit for bit;
 * two ranks (two processes on THIS GPU, gloo over 127.0.0.1 standing in for RCCL): same stopping
   decisions and price as the single solve of the whole ensemble, up to the order of the sums."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import hedgehog_jl_amd as hh
from tests.shard_worker import problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    dict(model="gbm", n=6000, steps=20, degree=4, anti=1, cp=-1.0, strike=100.0, seed=7),
    dict(model="gbm", n=3001, steps=7, degree=2, anti=0, cp=-1.0, strike=110.0, seed=8),
    dict(model="heston", n=2500, steps=6, degree=3, anti=0, cp=-1.0, strike=100.0, seed=9),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['model']}-{c['n']}x{c['steps']}")
def test_one_rank_phases_equal_the_fused_solve(case):
    prob, method = problem(case)
    fused = hh.solve(prob, method)
    phased = hh.solve_lsm_sharded(prob, method, stopping_info=True)
    assert phased.price == fused.price and phased.std_error == fused.std_error
    np.testing.assert_array_equal(phased.stopping_info[0], fused.stopping_info[0])
    np.testing.assert_array_equal(phased.stopping_info[1], fused.stopping_info[1])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("case", CASES[:2] + CASES[2:], ids=lambda c: f"{c['model']}-{c['n']}x{c['steps']}")
def test_two_ranks_equal_the_single_solve(case, tmp_path):
    world, port = 2, _free_port()
    outs = [str(tmp_path / f"r{r}.json") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"),
                               str(r), str(world), str(port), outs[r], json.dumps(case)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    res = [json.load(open(o)) for o in outs]
    prob, method = problem(case)
    single = hh.solve(prob, method)
    n, anti = case["n"], case["anti"]
    assert res[0]["price"] == res[1]["price"] and res[0]["n_total"] == n * (2 if anti else 1)
    # the shards' stopping decisions, put back in the single solve's trajectory order
    per = -(-n // world)
    tau = np.empty(n * (2 if anti else 1), dtype=np.int64)
    val = np.empty_like(tau, dtype=np.float64)
    for r in range(world):
        a, b = min(n, r * per), min(n, (r + 1) * per)
        t, v = np.array(res[r]["tau"]), np.array(res[r]["val"])
        tau[a:b], val[a:b] = t[:b - a], v[:b - a]
        if anti:
            tau[n + a:n + b], val[n + a:n + b] = t[b - a:], v[b - a:]
    same = tau == single.stopping_info[0]
    assert same.mean() >= 0.998
    np.testing.assert_allclose(val[same], single.stopping_info[1][same], rtol=1e-12)
    assert res[0]["price"] == pytest.approx(single.price, rel=2e-4 if not same.all() else 1e-11)
    assert res[0]["std_error"] == pytest.approx(single.std_error, rel=1e-3)


@pytest.mark.parametrize("case", [dict(model="european", n=70_001, steps=40, anti=1, P=3),
                                  dict(model="european", n=5_000, steps=16, anti=0, P=0)],
                         ids=["greeks-antithetic", "plain"])
def test_two_ranks_european_solve_sharded_on_the_gpu(case, tmp_path):
    """solve_sharded (the European path: ONE all-reduce of the accumulator vector) with the HIP
    kernels in both ranks — two processes on this GPU, gloo — against the single solve."""
    from tests.shard_worker import european_problem
    world, port = 2, _free_port()
    outs = [str(tmp_path / f"e{r}.json") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"),
                               str(r), str(world), str(port), outs[r], json.dumps(case)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    res = [json.load(open(o)) for o in outs]
    prob, method = european_problem(case)
    single = hh.solve(prob, method, ensemble=False)
    assert res[0] == res[1] and res[0]["n_total"] == case["n"]
    assert res[0]["price"] == pytest.approx(hh.value_of(single.price), rel=1e-12)
    assert res[0]["std_error"] == pytest.approx(single.std_error, rel=1e-9)
    for k in range(case["P"]):
        assert res[0]["dprice"][k] == pytest.approx(hh.partials_of(single.price, case["P"])[k], rel=1e-11)
