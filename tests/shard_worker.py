"""One rank of the sharded-LSM test (tests/test_gpu_sharded.py): joins a gloo group over
127.0.0.1, prices its shard on cuda:0 through hedgehog_jl_amd.solve_lsm_sharded and writes what it
got.  usage: shard_worker.py <rank> <world> <port> <out.json> <case-json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import hedgehog_jl_amd as hh  # noqa: E402


def problem(case):
    ref = hh.Date(2020, 1, 1)
    expiry = hh.add_years(ref, 1)
    cp = hh.Put() if case["cp"] < 0 else hh.Call()
    payoff = hh.VanillaOption(case["strike"], expiry, hh.American(), cp, hh.Spot())
    n = case["n"]
    seeds = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(case["seed"])
    vr = hh.Antithetic() if case["anti"] else hh.NoVarianceReduction()
    cfg = hh.SimulationConfig(n, steps=case["steps"], seeds=seeds, variance_reduction=vr)
    if case["model"] == "gbm":
        mkt = hh.BlackScholesInputs(ref, 0.05, 100.0, 0.25)
        mc = hh.MonteCarlo(hh.LognormalDynamics(), hh.BlackScholesExact(), cfg)
    else:
        mkt = hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7)
        mc = hh.MonteCarlo(hh.HestonDynamics(), hh.HestonBroadieKaya(), cfg)
    return hh.PricingProblem(payoff, mkt), hh.LSM(mc, case["degree"])


def european_problem(case):
    """Heston Euler call with (spot, V0, rate) carried as dual numbers: the fused Greek pass."""
    ref, expiry = hh.Date(2021, 1, 1), hh.Date(2022, 1, 1)
    P = case["P"]
    dual = (lambda v, k: hh.Dual(v, tuple(1.0 if j == k else 0.0 for j in range(P)))) if P else \
        (lambda v, k: v)
    mkt = hh.HestonInputs(ref, dual(0.03, 2), dual(100.0, 0), dual(0.04, 1), 2.0, 0.04, 0.3, -0.7)
    payoff = hh.VanillaOption(100.0, expiry, hh.European(), hh.Call(), hh.Spot())
    n = case["n"]
    vr = hh.Antithetic() if case["anti"] else hh.NoVarianceReduction()
    cfg = hh.SimulationConfig(n, steps=case["steps"], seeds=np.arange(1, n + 1, dtype=np.uint64),
                              variance_reduction=vr)
    return hh.PricingProblem(payoff, mkt), hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(), cfg)


if __name__ == "__main__":
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    case = json.loads(sys.argv[5])
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    if case["model"] == "european":
        prob, method = european_problem(case)
        sol = hh.solve_sharded(prob, method)
        price = sol.price
        json.dump({"price": hh.value_of(price), "dprice": list(hh.partials_of(price, case["P"])),
                   "std_error": sol.std_error, "n_total": int(sol.result.n_paths_done)}, open(out, "w"))
    else:
        prob, method = problem(case)
        sol = hh.solve_lsm_sharded(prob, method, stopping_info=True)
        tau, val = sol.stopping_info
        json.dump({"price": sol.price, "std_error": sol.std_error,
                   "n_total": int(sol.result.n_paths_total), "tau": tau.tolist(), "val": val.tolist()},
                  open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()
