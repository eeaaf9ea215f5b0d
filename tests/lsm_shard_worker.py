"""One rank of the sharded-LSM test (tests/test_gpu_lsm_sharded.py): joins a gloo group over
127.0.0.1, prices its shard on cuda:0 through hedgehog_jl_amd.solve_lsm_sharded and writes what it
got.  usage: lsm_shard_worker.py <rank> <world> <port> <out.json> <case-json>"""
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


if __name__ == "__main__":
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    case = json.loads(sys.argv[5])
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    prob, method = problem(case)
    sol = hh.solve_lsm_sharded(prob, method, stopping_info=True)
    tau, val = sol.stopping_info
    json.dump({"price": sol.price, "std_error": sol.std_error, "n_total": int(sol.result.n_paths_total),
               "tau": tau.tolist(), "val": val.tolist()}, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()
