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
    """The model scalars every payoff on these market inputs shares (marginal_law, montecarlo.jl:293-320)
    -> (hh_model of their VALUES, dynamics, the scalars themselves — possibly Dual — in the order of
    enum hh_cm_grad: S0, V0, kappa, theta, sigma, rho)."""
    from .dual import value_of
    model = _ffi.hh_model()
    if isinstance(method.dynamics, HestonDynamics) and isinstance(m, HestonInputs):
        dyn = _ffi.HH_HESTON
        scal = [m.spot, m.V0, m.κ, m.θ, m.σ, m.ρ]
        model.V0, model.kappa, model.theta = value_of(m.V0), value_of(m.κ), value_of(m.θ)
        model.sigma, model.rho = value_of(m.σ), value_of(m.ρ)
    elif isinstance(method.dynamics, LognormalDynamics) and isinstance(m, BlackScholesInputs):
        dyn = _ffi.HH_LOGNORMAL
        vol = get_vol(m.sigma, None, None)
        scal = [m.spot, 0.0, 0.0, 0.0, vol, 0.0]
        model.sigma = value_of(vol)
    else:
        raise MethodError("no marginal_law for this dynamics / market-input pair")
    model.S0 = value_of(m.spot)
    return model, dyn, scal


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
    from .dual import Dual, n_partials, partials_of, value_of
    model, dyn, scal = _carr_madan_model(m, method)
    K = len(payoffs)
    if any(isinstance(p.strike, Dual) for p in payoffs):
        raise MethodError("CarrMadan basket: partials along a strike are not carried")
    strikes = np.array([float(p.strike) for p in payoffs])
    cps = np.array([p.call_put() for p in payoffs], dtype=np.float64)
    Ts = np.array([yearfrac(m.rate.reference_date, p.expiry) for p in payoffs])   # montecarlo.jl:301,317
    r_k = [zero_rate(m.rate, p.expiry) for p in payoffs]                          # montecarlo.jl:299,318
    D_k = [df(m.rate, p.expiry) for p in payoffs]                                 # carr_madan.jl:89
    rs, Ds = np.array([value_of(x) for x in r_k]), np.array([value_of(x) for x in D_k])
    out = np.empty(K)
    ctx = _ffi.get_context(method.device)
    args = (ctx.handle, C.byref(model), dyn, int(method.compat_sqrt_alpha), float(method.α),
            float(method.bound), strikes.ctypes.data, cps.ctypes.data, Ts.ctypes.data, rs.ctypes.data,
            Ds.ctypes.data, K, out.ctypes.data)
    P = n_partials(*scal, *r_k, *D_k)
    if P == 0:
        ctx.check(ctx.lib.hh_carr_madan_basket(*args))
        return out
    # a differentiated objective (calibration.jl:75-88 under AutoForwardDiff): the device returns the
    # gradient along the eight scalars of enum hh_cm_grad, the Dual prices are assembled here
    grad = np.empty((K, _ffi.HH_CM_GRAD_LEN))
    ctx.check(ctx.lib.hh_carr_madan_basket_grad(*args, grad.ctypes.data))
    seeds = np.array([partials_of(x, P) for x in scal])                # [6][P]
    prices = []
    for k in range(K):
        d = grad[k, :6] @ seeds + grad[k, 6] * np.array(partials_of(r_k[k], P)) \
            + grad[k, 7] * np.array(partials_of(D_k[k], P))
        prices.append(Dual(out[k], tuple(d)))
    return prices


def solve_carr_madan(prob: PricingProblem, method: CarrMadan) -> AnalyticSolution:
    """carr_madan.jl:47-71 with marginal_law (montecarlo.jl:293-320)."""
    payoff, m = prob.payoff, prob.market_inputs
    if not (isinstance(payoff, VanillaOption) and isinstance(payoff.exercise_style, European)):
        raise MethodError("CarrMadan: European VanillaOption")
    from .dual import n_partials
    model, dyn, scal = _carr_madan_model(m, method)
    r_k, D_k = zero_rate(m.rate, payoff.expiry), df(m.rate, payoff.expiry)
    if n_partials(*scal, r_k, D_k) > 0:
        # a Dual input (solve(GreekProblem(prob, lens), ForwardAD(), CarrMadan(...)), greeks_problem.jl:249-262):
        # the price must carry the partials — the gradient form, as a basket of one
        return AnalyticSolution(prob, method, solve_carr_madan_basket([payoff], m, method)[0])
    model.strike, model.cp = float(payoff.strike), payoff.call_put()
    model.T = yearfrac(m.rate.reference_date, payoff.expiry)      # montecarlo.jl:301,317
    model.r_drift = float(r_k)                                    # montecarlo.jl:299,318
    model.discount = float(D_k)                                   # carr_madan.jl:89
    out = C.c_double()
    ctx = _ffi.get_context(method.device)
    ctx.check(ctx.lib.hh_carr_madan(ctx.handle, C.byref(model), dyn, int(method.compat_sqrt_alpha),
                                    float(method.α), float(method.bound), C.byref(out)))
    return AnalyticSolution(prob, method, out.value)
