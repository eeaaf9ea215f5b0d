"""Host mirror of /root/reference/src/greeks/greeks_problem.jl (hot-path part):
lenses (:18-130), ForwardAD (:193, :249-262), FiniteDifference (:204-220, :279-329),
BatchGreekProblem (:541-568) and ZeroRateSpineLens (pricing_methods.jl:26-60).

ForwardAD pushes a host `Dual` through `set(prob, lens, ·)`; solve() hands its partials to the
kernels as dual seeds and returns a Dual price — exactly the contract ForwardDiff.derivative needs
at greeks_problem.jl:258-260.  BatchGreekProblem + ForwardAD + MonteCarlo seeds one partial per
lens and makes ONE fused pass (the reference re-simulates the same noise once per lens,
greeks_problem.jl:567, so the results coincide).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass
from typing import Any

from .dual import Dual
from .domain import BlackScholesInputs, FlatRateCurve, FlatVolSurface, RateCurve


def _replace(obj, name, val):
    new = copy.copy(obj)
    object.__setattr__(new, name, val)
    return new


def _set_path(obj, path, val):
    if len(path) == 1:
        return _replace(obj, path[0], val)
    return _replace(obj, path[0], _set_path(getattr(obj, path[0]), path[1:], val))


class GreekLens: pass


@dataclass(frozen=True)
class PropertyLens(GreekLens):
    """Accessors' `@optic _.market_inputs.spot`: optic("market_inputs.spot")."""
    path: tuple

    def __call__(self, prob):
        for p in self.path:
            prob = getattr(prob, p)
        return prob


def optic(path: str) -> PropertyLens:
    return PropertyLens(tuple(path.lstrip("_.").split(".")))


@dataclass(frozen=True)
class SpotLens(GreekLens):
    """greeks_problem.jl:18-49."""
    def __call__(self, p): return p.market_inputs.spot


@dataclass(frozen=True)
class VolLens(GreekLens):
    """greeks_problem.jl:56-80; flat surface only (:119-130)."""
    strike: Any
    expiry: Any

    def __call__(self, prob):
        sigma = getattr(prob.market_inputs, "sigma", None)
        if not isinstance(sigma, FlatVolSurface):
            raise TypeError("VolLens needs market_inputs.sigma::FlatVolSurface")
        return sigma.σ


@dataclass(frozen=True)
class ZeroRateSpineLens(GreekLens):
    """pricing_methods.jl:26-60; getter defined for BlackScholesInputs with a flat curve (:30-32)
    or an interpolated RateCurve (:34-36, 1-based spine index)."""
    i: int

    def __call__(self, prob):
        m = prob.market_inputs
        if not isinstance(m, BlackScholesInputs):
            raise TypeError("ZeroRateSpineLens getter: BlackScholesInputs only")
        if isinstance(m.rate, RateCurve):
            return m.rate.zeros[self.i - 1]
        return m.rate.rate


def set(prob, lens, val):  # noqa: A001 - the reference's name (Accessors.set)
    if isinstance(lens, PropertyLens):
        return _set_path(prob, lens.path, val)
    if isinstance(lens, SpotLens):
        return _set_path(prob, ("market_inputs", "spot"), val)
    if isinstance(lens, VolLens):
        sigma = prob.market_inputs.sigma
        return _set_path(prob, ("market_inputs", "sigma"), FlatVolSurface(sigma.reference_date, val))
    if isinstance(lens, ZeroRateSpineLens):  # pricing_methods.jl:39-57
        curve = prob.market_inputs.rate
        if isinstance(curve, RateCurve):
            z = list(curve.zeros)
            z[lens.i - 1] = val
            new = RateCurve(curve.reference_date, curve.tenors, zeros=z)
        else:
            new = FlatRateCurve(curve.reference_date, val)
        return _set_path(prob, ("market_inputs", "rate"), new)
    raise TypeError(f"unknown lens {lens!r}")


# ---- methods ----
class GreekMethod: pass
class ForwardAD(GreekMethod): pass
class FDScheme: pass
class FDForward(FDScheme): pass
class FDBackward(FDScheme): pass
class FDCentral(FDScheme): pass


@dataclass(frozen=True)
class FiniteDifference(GreekMethod):
    """greeks_problem.jl:204-220: relative bump, central by default."""
    bump: float
    scheme: Any = None

    def __post_init__(self):
        if self.scheme is None:
            object.__setattr__(self, "scheme", FDCentral())


@dataclass(frozen=True)
class GreekProblem:
    pricing_problem: Any
    wrt: Any


@dataclass(frozen=True)
class BatchGreekProblem:
    pricing_problem: Any
    lenses: Any


@dataclass(frozen=True)
class GreekResult:
    greek: Any


def _seed(x0, k, n):
    return Dual(float(x0), tuple(1.0 if j == k else 0.0 for j in range(n)))


def solve_greek_ad(gprob: GreekProblem, pricing_method, solve):
    """greeks_problem.jl:249-262."""
    prob, lens = gprob.pricing_problem, gprob.wrt
    x0 = lens(prob)
    price = solve(set(prob, lens, _seed(x0, 0, 1)), pricing_method).price
    if not isinstance(price, Dual):
        # the method took the Dual's value and returned a plain price (the closed-form host pricers do):
        # a Greek of 0 would be a silent wrong answer
        from .montecarlo import MethodError
        raise MethodError(f"{type(pricing_method).__name__} does not carry dual partials: use FiniteDifference")
    return GreekResult(price.partials[0])


def _prices(probs, pricing_method, solve):
    """The prices of several problems under one method.  Through MonteCarlo they are simulated TOGETHER, on
    the same draws in one pass (solve_montecarlo_many -> hh_mc_solve_multi): the same numbers as one solve
    after the other — which is what happens for every other method, and whenever a pass cannot be shared."""
    from .montecarlo import MonteCarlo, solve_montecarlo_many
    if isinstance(pricing_method, MonteCarlo):
        sols = solve_montecarlo_many(probs, pricing_method)
        if sols is not None:
            return [s.price for s in sols]
    return [solve(p, pricing_method).price for p in probs]


def solve_greek_fd(gprob: GreekProblem, method: FiniteDifference, pricing_method, solve):
    """greeks_problem.jl:279-329 (common random numbers come from the fixed seeds): the two solves of a
    scheme, in the reference's order."""
    prob, lens, eps = gprob.pricing_problem, gprob.wrt, method.bump
    x0 = lens(prob)
    if isinstance(method.scheme, FDForward):
        v_up, v0 = _prices([set(prob, lens, x0 * (1 + eps)), prob], pricing_method, solve)
        d = (v_up - v0) / (x0 * eps)
    elif isinstance(method.scheme, FDBackward):
        v_down, v0 = _prices([set(prob, lens, x0 * (1 - eps)), prob], pricing_method, solve)
        d = (v0 - v_down) / (x0 * eps)
    else:
        v_up, v_down = _prices([set(prob, lens, x0 * (1 + eps)), set(prob, lens, x0 * (1 - eps))],
                               pricing_method, solve)
        d = (v_up - v_down) / (2 * eps * x0)
    return GreekResult(d)


@dataclass(frozen=True)
class SecondOrderGreekProblem:
    """greeks_problem.jl:341-345."""
    pricing_problem: Any
    wrt1: Any
    wrt2: Any


def solve_second_order_fd(gprob: SecondOrderGreekProblem, method: FiniteDifference,
                          pricing_method, solve):
    """greeks_problem.jl:396-422: ABSOLUTE bump ε, 3-point (same lens) or 4-point cross stencil on
    top of plain solves (common random numbers through the fixed seeds) — what the reference's own
    test uses for the MC gamma (greeks_agreement.jl:221-224).  Second-order ForwardAD through the
    Monte Carlo path is not offered (the reference flags it as unstable there)."""
    prob, l1, l2, eps = gprob.pricing_problem, gprob.wrt1, gprob.wrt2, method.bump
    x0, y0 = l1(prob), l2(prob)
    at = lambda x, y: set(set(prob, l1, x), l2, y)
    if l1 == l2:
        f_plus, f_0, f_minus = _prices([at(x0 + eps, y0 + eps), at(x0, y0), at(x0 - eps, y0 - eps)],
                                       pricing_method, solve)
        d = (f_plus - 2 * f_0 + f_minus) / eps**2
    else:
        f_pp, f_pm, f_mp, f_mm = _prices([at(x0 + eps, y0 + eps), at(x0 + eps, y0 - eps), at(x0 - eps, y0 + eps),
                                          at(x0 - eps, y0 - eps)], pricing_method, solve)
        d = (f_pp - f_pm - f_mp + f_mm) / (4 * eps**2)
    return GreekResult(d)


def solve_batch(gprob: BatchGreekProblem, method, pricing_method, solve):
    """greeks_problem.jl:559-568 -> {lens: greek}; ForwardAD through MonteCarlo is one fused pass."""
    from .montecarlo import MonteCarlo
    lenses, prob = tuple(gprob.lenses), gprob.pricing_problem
    if isinstance(method, ForwardAD) and isinstance(pricing_method, MonteCarlo) and \
            0 < len(lenses) <= 8:
        n = len(lenses)
        p = prob
        for k, lens in enumerate(lenses):
            x0 = lens(p)
            if isinstance(x0, Dual):  # two lenses on the same input
                x0 = Dual(x0.value, tuple(a + (1.0 if j == k else 0.0)
                                          for j, a in enumerate(x0.partials)))
            else:
                x0 = _seed(x0, k, n)
            p = set(p, lens, x0)
        price = solve(p, pricing_method).price
        return {lens: price.partials[k] for k, lens in enumerate(lenses)}
    if isinstance(method, FiniteDifference) and isinstance(pricing_method, MonteCarlo) and \
            isinstance(method.scheme, FDCentral) and 0 < 2 * len(lenses) <= 16:
        # greeks_problem.jl:567 runs compute_fd_derivative once per lens: 2 solves each, all on the same
        # seeds — here every bumped problem of the batch shares the draws (up to 4 per pass)
        eps, xs = method.bump, [lens(prob) for lens in lenses]
        probs = [set(prob, lens, x * (1 + sg * eps)) for lens, x in zip(lenses, xs) for sg in (1, -1)]
        v = _prices(probs, pricing_method, solve)
        return {lens: (v[2 * i] - v[2 * i + 1]) / (2 * eps * x) for i, (lens, x) in enumerate(zip(lenses, xs))}
    return {lens: solve(GreekProblem(prob, lens), method, pricing_method).greek for lens in lenses}
