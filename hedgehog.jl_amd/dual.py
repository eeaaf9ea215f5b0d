"""Forward-mode dual number used on the HOST side of the boundary.

Mirrors what ForwardDiff.Dual does for the reference at greeks_problem.jl:258-260: a lens `set`
replaces one input by Dual(x0, 1); `solve` must then return a price that carries the partials.
Here the host arithmetic on duals is limited to what solve() does before/after the kernels
(log S0, exp(-z T), discount * mean); everything per-path happens on the GPU, where the partials
travel as the `d*` seed vectors of hh_model.
"""
from __future__ import annotations

import math


class Dual:
    __slots__ = ("value", "partials")

    def __init__(self, value: float, partials=()):
        self.value = float(value)
        self.partials = tuple(float(p) for p in partials)

    # -- helpers
    @staticmethod
    def _lift(x, n):
        if isinstance(x, Dual):
            if len(x.partials) != n:
                raise ValueError("Dual numbers with different numbers of partials")
            return x
        return Dual(x, (0.0,) * n)

    def _bin(self, other, f, dfa, dfb):
        o = Dual._lift(other, len(self.partials))
        return Dual(f(self.value, o.value),
                    tuple(dfa(self.value, o.value) * a + dfb(self.value, o.value) * b
                          for a, b in zip(self.partials, o.partials)))

    # -- arithmetic
    def __add__(self, o): return self._bin(o, lambda a, b: a + b, lambda a, b: 1.0, lambda a, b: 1.0)
    __radd__ = __add__
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b, lambda a, b: 1.0, lambda a, b: -1.0)
    def __rsub__(self, o): return Dual._lift(o, len(self.partials)).__sub__(self)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b, lambda a, b: b, lambda a, b: a)
    __rmul__ = __mul__
    def __truediv__(self, o):
        return self._bin(o, lambda a, b: a / b, lambda a, b: 1.0 / b, lambda a, b: -a / (b * b))
    def __rtruediv__(self, o): return Dual._lift(o, len(self.partials)).__truediv__(self)
    def __neg__(self): return Dual(-self.value, tuple(-p for p in self.partials))
    def __pos__(self): return self
    def __pow__(self, n):
        if isinstance(n, Dual):
            raise TypeError("Dual ** Dual is not needed on this path")
        return Dual(self.value ** n, tuple(n * self.value ** (n - 1) * p for p in self.partials))

    # -- comparisons act on the value part (as ForwardDiff does)
    def __float__(self): return self.value
    def __lt__(self, o): return self.value < value_of(o)
    def __le__(self, o): return self.value <= value_of(o)
    def __gt__(self, o): return self.value > value_of(o)
    def __ge__(self, o): return self.value >= value_of(o)
    def __eq__(self, o): return self.value == value_of(o) and self.partials == partials_of(o, len(self.partials))
    def __hash__(self): return hash((self.value, self.partials))
    def __repr__(self): return f"Dual({self.value!r}, {self.partials!r})"


def value_of(x) -> float:
    return x.value if isinstance(x, Dual) else float(x)


def partials_of(x, n: int):
    if isinstance(x, Dual):
        if len(x.partials) != n:
            raise ValueError("Dual numbers with different numbers of partials")
        return x.partials
    return (0.0,) * n


def n_partials(*xs) -> int:
    n = 0
    for x in xs:
        if isinstance(x, Dual):
            if n and len(x.partials) != n:
                raise ValueError("Dual numbers with different numbers of partials")
            n = len(x.partials)
    return n


def dexp(x):
    if isinstance(x, Dual):
        e = math.exp(x.value)
        return Dual(e, tuple(e * p for p in x.partials))
    return math.exp(x)


def dlog(x):
    if isinstance(x, Dual):
        return Dual(math.log(x.value), tuple(p / x.value for p in x.partials))
    return math.log(x)


def dsqrt(x):
    if isinstance(x, Dual):
        r = math.sqrt(x.value)
        return Dual(r, tuple(p / (2.0 * r) for p in x.partials))
    return math.sqrt(x)
