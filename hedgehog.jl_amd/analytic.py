"""The two non-Monte-Carlo pricers the reference's MC tests and examples compare against, mirrored
so that such scripts run unchanged:

  BlackScholesAnalytic   /root/reference/src/pricing_methods/black_scholes.jl:38-64 — closed form,
                         evaluated on the host (a dozen flops; nothing to accelerate)
  CarrMadan(α, bound, dynamics)
                         /root/reference/src/pricing_methods/carr_madan.jl:15-92 — the Fourier
                         integral runs on the device (`hh_carr_madan`, csrc/hh_fourier.hip)
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Any

from . import _ffi
from .dates import yearfrac
from .domain import (BlackScholesInputs, European, HestonInputs, PricingProblem, VanillaOption, df,
                     get_vol, zero_rate)
from .montecarlo import AbstractPricingMethod, HestonDynamics, LognormalDynamics, MethodError


@dataclass(frozen=True)
class AnalyticSolution:
    """pricing_solutions.jl AnalyticSolution / CarrMadanSolution (price only)."""
    problem: Any
    method: Any
    price: float


class BlackScholesAnalytic(AbstractPricingMethod):
    def __eq__(self, o): return type(o) is type(self)
    def __hash__(self): return hash("BlackScholesAnalytic")


@dataclass(frozen=True)
class CarrMadan(AbstractPricingMethod):
    """carr_madan.jl:15-45: CarrMadan(α, bound, dynamics)."""
    α: float
    bound: float
    dynamics: Any
    compat_sqrt_alpha: bool = False   # montecarlo.jl:302 quirk Q1 for the lognormal law
    device: int = 0


def _ncdf(x):
    return 0.5 * math.erfc(-x / math.sqrt(2.0))


def solve_black_scholes(prob: PricingProblem, method: BlackScholesAnalytic) -> AnalyticSolution:
    """black_scholes.jl:38-64."""
    payoff, m = prob.payoff, prob.market_inputs
    if not (isinstance(payoff, VanillaOption) and isinstance(payoff.exercise_style, European)
            and isinstance(m, BlackScholesInputs)):
        raise MethodError("BlackScholesAnalytic: European VanillaOption on BlackScholesInputs")
    K = float(payoff.strike)
    sigma = float(get_vol(m.sigma, payoff.expiry, K))
    cp = payoff.call_put()
    T = yearfrac(m.referenceDate, payoff.expiry)
    D = float(df(m.rate, payoff.expiry))
    F = float(m.spot) / D
    if sigma == 0:
        price = D * max(cp * (F - K), 0.0)
    else:
        sq = math.sqrt(T)
        d1 = (math.log(F / K) + 0.5 * sigma * sigma * T) / (sigma * sq)
        d2 = d1 - sigma * sq
        price = D * cp * (F * _ncdf(cp * d1) - K * _ncdf(cp * d2))
    return AnalyticSolution(prob, method, price)


def _carr_madan_model(m, method: CarrMadan):
    """The model scalars every payoff on these market inputs shares (marginal_law, montecarlo.jl:293-320)."""
    model = _ffi.hh_model()
    if isinstance(method.dynamics, HestonDynamics) and isinstance(m, HestonInputs):
        dyn = _ffi.HH_HESTON
        model.V0, model.kappa, model.theta = float(m.V0), float(m.κ), float(m.θ)
        model.sigma, model.rho = float(m.σ), float(m.ρ)
    elif isinstance(method.dynamics, LognormalDynamics) and isinstance(m, BlackScholesInputs):
        dyn = _ffi.HH_LOGNORMAL
        model.sigma = float(get_vol(m.sigma, None, None))
    else:
        raise MethodError("no marginal_law for this dynamics / market-input pair")
    model.S0 = float(m.spot)
    return model, dyn


def solve_carr_madan_basket(payoffs, market_inputs, method: CarrMadan):
    """solve(::BasketPricingProblem, ::CarrMadan) — basket.jl:35-38 over carr_madan.jl:47-71, the
    calibration objective's inner loop (calibration.jl:75-88): every payoff's Fourier integral in ONE
    launch (`hh_carr_madan_basket`, a workgroup per payoff).  Returns the prices in order."""
    import numpy as np
    m = market_inputs
    payoffs = list(payoffs)
    for payoff in payoffs:
        if not (isinstance(payoff, VanillaOption) and isinstance(payoff.exercise_style, European)):
            raise MethodError("CarrMadan: European VanillaOption")
    model, dyn = _carr_madan_model(m, method)
    K = len(payoffs)
    strikes = np.array([float(p.strike) for p in payoffs])
    cps = np.array([p.call_put() for p in payoffs], dtype=np.float64)
    Ts = np.array([yearfrac(m.rate.reference_date, p.expiry) for p in payoffs])   # montecarlo.jl:301,317
    rs = np.array([float(zero_rate(m.rate, p.expiry)) for p in payoffs])          # montecarlo.jl:299,318
    Ds = np.array([float(df(m.rate, p.expiry)) for p in payoffs])                 # carr_madan.jl:89
    out = np.empty(K)
    ctx = _ffi.get_context(method.device)
    ctx.check(ctx.lib.hh_carr_madan_basket(ctx.handle, C.byref(model), dyn, int(method.compat_sqrt_alpha),
                                           float(method.α), float(method.bound), strikes.ctypes.data,
                                           cps.ctypes.data, Ts.ctypes.data, rs.ctypes.data, Ds.ctypes.data,
                                           K, out.ctypes.data))
    return out


def solve_carr_madan(prob: PricingProblem, method: CarrMadan) -> AnalyticSolution:
    """carr_madan.jl:47-71 with marginal_law (montecarlo.jl:293-320)."""
    payoff, m = prob.payoff, prob.market_inputs
    if not (isinstance(payoff, VanillaOption) and isinstance(payoff.exercise_style, European)):
        raise MethodError("CarrMadan: European VanillaOption")
    model, dyn = _carr_madan_model(m, method)
    model.strike, model.cp = float(payoff.strike), payoff.call_put()
    model.T = yearfrac(m.rate.reference_date, payoff.expiry)      # montecarlo.jl:301,317
    model.r_drift = float(zero_rate(m.rate, payoff.expiry))       # montecarlo.jl:299,318
    model.discount = float(df(m.rate, payoff.expiry))             # carr_madan.jl:89
    out = C.c_double()
    ctx = _ffi.get_context(method.device)
    ctx.check(ctx.lib.hh_carr_madan(ctx.handle, C.byref(model), dyn, int(method.compat_sqrt_alpha),
                                    float(method.α), float(method.bound), C.byref(out)))
    return AnalyticSolution(prob, method, out.value)
