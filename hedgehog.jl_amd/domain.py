"""Domain types of the hot path, mirroring the reference's constructors.

  payoffs         /root/reference/src/payoffs/payoffs.jl:62-156
  market inputs   /root/reference/src/market_inputs/market_inputs.jl:21-88
  flat curve/vol  /root/reference/src/market_inputs/rate_curve.jl:35-56,149-150,185-186,
                  /root/reference/src/market_inputs/vol_surface.jl:73-98
  PricingProblem  /root/reference/src/pricing_methods/pricing_methods.jl:19-22
  solution        /root/reference/src/solutions/pricing_solutions.jl:22-27
Only what solve(::PricingProblem, ::MonteCarlo) touches is here; interpolated curves, surfaces,
American/Forward payoffs are outside the accelerated path (the types exist so that unsupported
combinations fail the way the reference's dispatch does).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

from .dates import to_ticks, yearfrac
from .dual import dexp


# ---- payoffs.jl ----
class European: pass
class American: pass
class Spot: pass
class Forward: pass


class Call:
    def __call__(self): return 1.0     # payoffs.jl:76-78


class Put:
    def __call__(self): return -1.0    # payoffs.jl:85-87


def _tag_eq(cls):
    cls.__eq__ = lambda a, b: type(a) is type(b)
    cls.__hash__ = lambda a: hash(type(a).__name__)
    cls.__repr__ = lambda a: type(a).__name__ + "()"
    return cls


for _c in (European, American, Spot, Forward, Call, Put):
    _tag_eq(_c)


@dataclass(frozen=True)
class VanillaOption:
    """payoffs.jl:101-140; expiry is stored in ticks."""
    strike: Any
    expiry: int
    exercise_style: Any
    call_put: Any
    underlying: Any

    def __init__(self, strike, expiry_date, exercise_style, call_put, underlying):
        object.__setattr__(self, "strike", strike)
        object.__setattr__(self, "expiry", to_ticks(expiry_date))
        object.__setattr__(self, "exercise_style", exercise_style)
        object.__setattr__(self, "call_put", call_put)
        object.__setattr__(self, "underlying", underlying)

    def __call__(self, spot):
        """payoffs.jl:154-156: max(cp (S-K), 0), scalar or numpy array."""
        import numpy as np
        return np.maximum(self.call_put() * (np.asarray(spot) - self.strike), 0.0)


# ---- rate curve / vol surface (flat only) ----
@dataclass(frozen=True)
class FlatRateCurve:
    """rate_curve.jl:35-56."""
    reference_date: int
    rate: Any

    def __init__(self, *args, reference_date=None):
        if len(args) == 2:               # FlatRateCurve(reference_ticks, rate)
            ref, rate = to_ticks(args[0]), args[1]
        else:                            # FlatRateCurve(rate; reference_date=Date(0))
            rate = args[0]
            ref = to_ticks(reference_date) if reference_date is not None else 0
        object.__setattr__(self, "reference_date", ref)
        object.__setattr__(self, "rate", rate)


def zero_rate(curve: FlatRateCurve, ticks):
    """rate_curve.jl:185-186."""
    return curve.rate


def df(curve: FlatRateCurve, t):
    """rate_curve.jl:149-150."""
    ticks = to_ticks(t)
    return dexp(-zero_rate(curve, ticks) * yearfrac(curve.reference_date, ticks))


@dataclass(frozen=True)
class FlatVolSurface:
    """vol_surface.jl:73-81."""
    reference_date: Any
    σ: Any

    def __init__(self, *args, reference_date=None):
        if len(args) == 2:
            ref, s = to_ticks(args[0]), args[1]
        else:
            s = args[0]
            ref = to_ticks(reference_date) if reference_date is not None else 0
        object.__setattr__(self, "reference_date", ref)
        object.__setattr__(self, "σ", s)

    @property
    def sigma(self):
        return self.σ


def get_vol(surf: FlatVolSurface, _t=None, _k=None):
    """vol_surface.jl:87-89."""
    return surf.σ


# ---- market inputs ----
@dataclass(frozen=True)
class BlackScholesInputs:
    """market_inputs.jl:21-36."""
    referenceDate: int
    rate: FlatRateCurve
    spot: Any
    sigma: FlatVolSurface

    def __init__(self, reference_date, rate, spot, sigma):
        ref = to_ticks(reference_date)
        if not isinstance(rate, FlatRateCurve):
            rate = FlatRateCurve(ref, rate)
        if not isinstance(sigma, FlatVolSurface):
            sigma = FlatVolSurface(ref, sigma)
        object.__setattr__(self, "referenceDate", ref)
        object.__setattr__(self, "rate", rate)
        object.__setattr__(self, "spot", spot)
        object.__setattr__(self, "sigma", sigma)


@dataclass(frozen=True)
class HestonInputs:
    """market_inputs.jl:55-88; positional order (ref, rate, spot, V0, κ, θ, σ, ρ)."""
    referenceDate: int
    rate: FlatRateCurve
    spot: Any
    V0: Any
    κ: Any
    θ: Any
    σ: Any
    ρ: Any

    def __init__(self, reference_date, rate, spot, V0, κ, θ, σ, ρ):
        ref = to_ticks(reference_date)
        if not isinstance(rate, FlatRateCurve):
            rate = FlatRateCurve(ref, rate)
        for k, v in (("referenceDate", ref), ("rate", rate), ("spot", spot), ("V0", V0), ("κ", κ),
                     ("θ", θ), ("σ", σ), ("ρ", ρ)):
            object.__setattr__(self, k, v)

    # ASCII aliases
    kappa = property(lambda s: s.κ)
    theta = property(lambda s: s.θ)
    sigma = property(lambda s: s.σ)
    rho = property(lambda s: s.ρ)


@dataclass(frozen=True)
class PricingProblem:
    """pricing_methods.jl:19-22."""
    payoff: Any
    market_inputs: Any


@dataclass(frozen=True)
class MonteCarloSolution:
    """pricing_solutions.jl:22-27; `ensemble` holds the samples at expiry (a pair when antithetic).

    `std_error`, `result` are build extensions (the reference computes no standard error)."""
    problem: Any
    method: Any
    price: Any
    ensemble: Any
    std_error: float = field(default=float("nan"), compare=False)
    result: Any = field(default=None, compare=False, repr=False)
