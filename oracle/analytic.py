"""Closed-form / Fourier prices the reference's own Monte Carlo tests compare against.

TEST INFRASTRUCTURE ONLY (see oracle/hh_oracle.c header): imported by tests/, never by the product.
PARITY STATUS: bs_price is PINNED by the reference's own known answers (QuantLib values of
test/unit/black_scholes.jl:93-127, tests/golden/reference_known_answers.json) and crr_price by its
regression values (test/unit/binomial_tree.jl:18,26); the Carr–Madan / Heston-CF restatement has no
golden vector in the reference (its tests compare it with Black–Scholes and with Monte Carlo only):
parity unpinned, cross-checked against bs_price at vanishing vol-of-vol and against put-call parity
(tests/test_oracle_pins.py).

Restates, with numpy/scipy:
  * BlackScholesAnalytic        /root/reference/src/pricing_methods/black_scholes.jl:38-64
  * Heston log-price CF         /root/reference/src/distributions/heston.jl:307-319
  * CarrMadan(α, bound)         /root/reference/src/pricing_methods/carr_madan.jl:47-92
    (the reference integrates with QuadGK over (-bound, bound); here scipy.integrate.quad on the
    real part, which is even in v)
  * parity_transform            /root/reference/src/payoffs/payoffs.jl:172-193

Pinned by the reference's QuantLib known answers (test/unit/black_scholes.jl:93-127) in
tests/test_oracle_pins.py.
"""
from __future__ import annotations

import cmath
import math

from scipy import integrate
from scipy.stats import norm


def bs_price(S0: float, K: float, r: float, sigma: float, T: float, cp: float = 1.0) -> float:
    """black_scholes.jl:38-64 (forward-measure form; flat curve so D = exp(-rT))."""
    D = math.exp(-r * T)
    F = S0 / D
    if sigma == 0:
        return D * max(cp * (F - K), 0.0)
    sqrtT = math.sqrt(T)
    d1 = (math.log(F / K) + 0.5 * sigma**2 * T) / (sigma * sqrtT)
    d2 = d1 - sigma * sqrtT
    return D * cp * (F * norm.cdf(cp * d1) - K * norm.cdf(cp * d2))


def bs_greeks(S0: float, K: float, r: float, sigma: float, T: float, cp: float = 1.0) -> dict:
    """Analytic delta / vega / rho of bs_price (greeks_problem.jl:437-530 gives the same closed forms)."""
    sqrtT = math.sqrt(T)
    d1 = (math.log(S0 / K) + (r + 0.5 * sigma**2) * T) / (sigma * sqrtT)
    d2 = d1 - sigma * sqrtT
    return {
        "delta": cp * norm.cdf(cp * d1),
        "vega": S0 * norm.pdf(d1) * sqrtT,
        "rho": cp * K * T * math.exp(-r * T) * norm.cdf(cp * d2),
    }


def heston_cf(u: complex, S0, V0, kappa, theta, sigma, rho, r, T) -> complex:
    """heston.jl:307-319, characteristic function of log S_T."""
    iu = 1j * u
    d1 = cmath.sqrt((kappa - rho * sigma * iu) ** 2 + sigma**2 * (iu + u * u))
    g = (kappa - rho * sigma * iu - d1) / (kappa - rho * sigma * iu + d1)
    C = (kappa * theta / sigma**2) * (
        (kappa - rho * sigma * iu - d1) * T
        - 2 * cmath.log((1 - g * cmath.exp(-d1 * T)) / (1 - g))
    )
    Dv = ((kappa - rho * sigma * iu - d1) / sigma**2) * (
        (1 - cmath.exp(-d1 * T)) / (1 - g * cmath.exp(-d1 * T))
    )
    return cmath.exp(C + Dv * V0 + iu * math.log(S0) + iu * r * T)


def carr_madan_heston(S0, K, r, V0, kappa, theta, sigma, rho, T, cp=1.0, alpha=1.0,
                      bound=32.0) -> float:
    """carr_madan.jl:47-92 with the Heston marginal law (montecarlo.jl:310-320)."""
    logK = math.log(K)
    D = math.exp(-r * T)
    damp = math.exp(-alpha * logK) / (2 * math.pi)

    def integrand(v):
        num = D * heston_cf(v - (alpha + 1) * 1j, S0, V0, kappa, theta, sigma, rho, r, T)
        den = alpha**2 + alpha - v * v + v * (2 * alpha + 1) * 1j
        return (damp * num / den * cmath.exp(-1j * v * logK)).real

    val, _ = integrate.quad(integrand, -bound, bound, limit=2000, epsabs=1e-12, epsrel=1e-12)
    call = val
    if cp > 0:
        return call
    return call - S0 + K * D  # put-call parity (payoffs.jl:172-193)


def crr_price(S0, K, r, sigma, T, steps, cp=1.0, american=True, on_forward=False) -> float:
    """Cox–Ross–Rubinstein tree exactly as /root/reference/src/pricing_methods/
    cox_ross_rubinstein.jl:99-141 builds it (tree on the FORWARD, u = e^{σ√ΔT}, p = 1/(1+u);
    spot node = e^{-r (steps-i) ΔT} · forward node, :75-81).  The reference's LSM tests compare
    against it (test/agreement/american_options.jl)."""
    import numpy as np
    dT = T / steps
    fwd = S0 / math.exp(-r * T)
    u = math.exp(sigma * math.sqrt(dT))
    p = 1.0 / (1.0 + u)
    disc = math.exp(-r * dT)

    def forward_at(i):
        return fwd * u ** np.arange(-i, i + 1, 2, dtype=np.float64)

    def underlying_at(i):
        f = forward_at(i)
        return f if on_forward else math.exp(-r * (steps - i) * dT) * f

    payoff = lambda s: np.maximum(cp * (s - K), 0.0)
    value = payoff(forward_at(steps))
    for step in range(steps - 1, -1, -1):
        cont = disc * (p * value[1:] + (1 - p) * value[:-1])
        value = np.maximum(cont, payoff(underlying_at(step))) if american else cont
    return float(value[0])
