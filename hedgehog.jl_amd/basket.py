"""Host mirror of /root/reference/src/calibration/basket.jl: `BasketPricingProblem`,
`BasketPricingSolution` and `solve(::BasketPricingProblem, ::MonteCarlo)`.

The reference prices a basket as independent solves (basket.jl:35-38).  With the fixed seeds of
`SimulationConfig`, payoffs that share an expiry see the same trajectories, so here every expiry
group is ONE simulation whose terminal samples are reduced against all of the group's strikes
(`hh_mc_solve_basket`).  Results equal the per-payoff solves (tests/test_gpu_basket.py).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any

import numpy as np

from . import _ffi
from .domain import MonteCarloSolution, PricingProblem
from .dual import Dual
from .montecarlo import MonteCarlo, _model_and_config, _price_from, solve_montecarlo


@dataclass(frozen=True)
class BasketPricingProblem:
    """basket.jl:10-13."""
    payoffs: Any
    market_inputs: Any


@dataclass(frozen=True)
class BasketPricingSolution:
    """basket.jl:24-27."""
    problem: BasketPricingProblem
    solutions: Any


def solve_basket(prob: BasketPricingProblem, method, ensemble: bool = False):
    """basket.jl:35-38 for a MonteCarlo method (one simulation per expiry group) or CarrMadan (every
    Fourier integral in one launch)."""
    from .analytic import AnalyticSolution, CarrMadan, solve_carr_madan_basket
    if isinstance(method, CarrMadan):
        prices = solve_carr_madan_basket(prob.payoffs, prob.market_inputs, method)
        return BasketPricingSolution(prob, [AnalyticSolution(PricingProblem(p, prob.market_inputs), method,
                                                             x if isinstance(x, Dual) else float(x))
                                            for p, x in zip(prob.payoffs, prices)])
    payoffs = list(prob.payoffs)
    sols: list = [None] * len(payoffs)
    groups: dict = {}
    for i, p in enumerate(payoffs):
        if isinstance(getattr(p, "strike", None), Dual):  # strike partials: plain per-payoff solve
            sols[i] = solve_montecarlo(PricingProblem(p, prob.market_inputs), method, ensemble)
        else:
            groups.setdefault(getattr(p, "expiry", None), []).append(i)
    cfg = method.config
    mg = _ffi.get_multi_gpu(tuple(method.devices)) if method.devices is not None else None
    ctx = None if mg is not None else _ffi.get_context(method.device)
    for idx in groups.values():
        first = PricingProblem(payoffs[idx[0]], prob.market_inputs)
        model, c, keep, P, discount = _model_and_config(first, method)  # raises MethodError as solve
        for i in idx[1:]:
            _model_and_config(PricingProblem(payoffs[i], prob.market_inputs), method)
        K = len(idx)
        strikes = (C.c_double * K)(*[float(payoffs[i].strike) for i in idx])
        cps = (C.c_double * K)(*[payoffs[i].call_put() for i in idx])
        anti = bool(c.antithetic)
        if mg is not None:  # ONE call, the group's simulation sharded over method.devices (hh_mgpu_solve_basket)
            if ensemble:
                raise ValueError("ensemble=True is not returned by the multi-GPU basket")
            c.seeds, c.seeds_len = cfg.seeds.ctypes.data, cfg.seeds.size
            res = (_ffi.hh_result * K)()
            mg.check(mg.lib.hh_mgpu_solve_basket(mg.handle, C.byref(model), C.byref(c), strikes, cps, K, res))
            for k, i in enumerate(idx):
                sols[i] = MonteCarloSolution(PricingProblem(payoffs[i], prob.market_inputs), method,
                                             _price_from(res[k], discount, P), None,
                                             std_error=res[k].std_error, result=res[k])
            del keep
            continue
        # the context's seed cache: uploaded once per config, not once per objective evaluation
        c.seeds, c.seeds_on_device = cfg.device_seeds(ctx), 1
        c.seeds_len = cfg.seeds.size
        term = np.empty(c.n_paths * (2 if anti else 1)) if ensemble else None
        res = (_ffi.hh_result * K)()
        ctx.check(ctx.lib.hh_mc_solve_basket(ctx.handle, C.byref(model), C.byref(c), strikes, cps, K,
                                             res, term.ctypes.data if ensemble else None))
        ens = None
        if ensemble:
            ens = (term[:c.n_paths], term[c.n_paths:]) if anti else term
        for k, i in enumerate(idx):
            sols[i] = MonteCarloSolution(PricingProblem(payoffs[i], prob.market_inputs), method,
                                         _price_from(res[k], discount, P), ens,
                                         std_error=res[k].std_error, result=res[k])
        del keep
    return BasketPricingSolution(prob, sols)
