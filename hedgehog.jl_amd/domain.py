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

import numpy as np

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


@dataclass(frozen=True)
class RateCurve:
    """rate_curve.jl:20-24, 60-91: zero rates on a tenor spine, built from discount factors
    (z = -log(df)/t), interpolated LINEARLY in the zero rate with CONSTANT extrapolation (the
    reference's default `interp`).  Host-side scalar lookups only; the kernels see r_drift and
    discount.  Zero rates may be `Dual` (ZeroRateSpineLens, pricing_methods.jl:34-50)."""
    reference_date: int
    tenors: tuple
    zeros: tuple

    def __init__(self, reference_date, tenors, dfs=None, zeros=None):
        tenors = tuple(float(t) for t in tenors)
        if not tenors:
            raise ValueError("Input 'tenors' cannot be empty.")
        if any(b < a for a, b in zip(tenors, tenors[1:])):
            raise ValueError("'tenors' must be sorted.")
        if tenors[0] < 0:
            raise ValueError("First tenor must be non-negative.")
        if zeros is None:
            if len(dfs) != len(tenors):
                raise ValueError("Mismatched lengths for 'tenors' and 'dfs'.")
            if not all(d > 0 for d in dfs):
                raise ValueError("All discount factors must be positive.")
            import math
            zeros = tuple(-math.log(d) / t for d, t in zip(dfs, tenors))
        object.__setattr__(self, "reference_date", to_ticks(reference_date))
        object.__setattr__(self, "tenors", tenors)
        object.__setattr__(self, "zeros", tuple(zeros))

    def interpolate(self, t):
        ts, zs = self.tenors, self.zeros
        if t <= ts[0]:
            return zs[0]
        if t >= ts[-1]:
            return zs[-1]
        import bisect
        i = bisect.bisect_right(ts, t) - 1
        w = (t - ts[i]) / (ts[i + 1] - ts[i])
        return zs[i] + (zs[i + 1] - zs[i]) * w


def spine_zeros(curve):
    """pricing_methods.jl:59, rate_curve.jl spine accessors."""
    return [curve.rate] if isinstance(curve, FlatRateCurve) else list(curve.zeros)


def zero_rate(curve, ticks):
    """rate_curve.jl:182-186: flat -> the rate; RateCurve -> interpolator(yearfrac(ref, ticks))."""
    if isinstance(curve, FlatRateCurve):
        return curve.rate
    return curve.interpolate(yearfrac(curve.reference_date, to_ticks(ticks)))


def df(curve, t):
    """rate_curve.jl:149-150."""
    ticks = to_ticks(t)
    return dexp(-zero_rate(curve, ticks) * yearfrac(curve.reference_date, ticks))


def zero_rate_yf(curve, yf):
    """rate_curve.jl:207-208: zero rate at a year fraction."""
    return curve.rate if isinstance(curve, FlatRateCurve) else curve.interpolate(yf)


def df_yf(curve, yf):
    """rate_curve.jl:171-172."""
    return dexp(-zero_rate_yf(curve, yf) * yf)


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
        if not isinstance(rate, (FlatRateCurve, RateCurve)):
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
        if not isinstance(rate, (FlatRateCurve, RateCurve)):
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


class _DeviceSamples:
    """Samples at expiry still in device memory: downloaded on the first read, then kept as plain
    arrays.  Reading after the owning Context was closed is a clear error, not a HIP fault."""
    __slots__ = ("_fetch", "_value", "_ctx")

    def __init__(self, fetch, ctx=None):
        self._fetch, self._value, self._ctx = fetch, None, ctx

    def get(self):
        if self._fetch is not None:
            if self._ctx is not None and getattr(self._ctx, "handle", None) is None:
                raise RuntimeError("MonteCarloSolution.ensemble was left on the device and its Context has "
                                   "been closed: read .ensemble before closing the context")
            self._value, self._fetch = self._fetch(), None
        return self._value


@dataclass(frozen=True, eq=False)
class MonteCarloSolution:
    """pricing_solutions.jl:22-27: (problem, method, price, ensemble); `ensemble` holds the samples at
    expiry (a pair when antithetic).  A plain frozen dataclass (`dataclasses.replace` / `asdict` /
    `fields` work) whose `ensemble` may be a device-side handle that is downloaded on the first read —
    most callers only read `price`, the download is 8 MB per 10^6 trajectories.  `std_error`, `result`
    are build extensions (the reference computes no standard error)."""
    problem: Any
    method: Any
    price: Any
    ensemble: Any = None
    std_error: float = float("nan")
    result: Any = field(default=None, repr=False, compare=False)

    def __getattribute__(self, name):
        v = object.__getattribute__(self, name)
        if name == "ensemble" and isinstance(v, _DeviceSamples):
            v = v.get()
        return v

    def __eq__(self, other):
        if not isinstance(other, MonteCarloSolution):
            return NotImplemented
        if (self.problem, self.method, self.price) != (other.problem, other.method, other.price):
            return False
        a, b = self.ensemble, other.ensemble
        if a is None or b is None:
            return a is None and b is None
        a, b = (a if isinstance(a, tuple) else (a,)), (b if isinstance(b, tuple) else (b,))
        return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))

    __hash__ = None

    def __repr__(self):
        return f"MonteCarloSolution(problem={self.problem!r}, method={self.method!r}, price={self.price!r})"
