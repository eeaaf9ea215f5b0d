"""Broadie–Kaya exact Heston sampling — CPU restatement (numpy/scipy) of the reference's
`MonteCarlo(HestonDynamics(), HestonBroadieKaya(), config)` path.

TEST INFRASTRUCTURE ONLY (see oracle/hh_oracle.c header).  PARITY STATUS: per-draw parity with the
reference is UNPINNED — its uniform/normal/NCχ² draws come from one sequential Xoshiro stream
through Distributions.jl/Rmath, its Bessel function from SpecialFunctions.jl (AMOS) and its root
finder from Roots.jl (`Order2`), none of which are under /root/reference.  What is kept exactly is
the reference's own arithmetic:

  sample_V_T            src/distributions/heston.jl:125-133   (constants d, λ, c)
  HestonCFIterator      heston.jl:150-176
  evaluate_chf          heston.jl:184-212   (incl. the continuous unwrapping of the Bessel argument)
  moments_from_cf       src/distributions/sample_from_cf.jl:50-64
  cdf_from_cf           sample_from_cf.jl:75-96  (series + stopping rule)
  sample_from_cf        sample_from_cf.jl:27-41
  inverse_cdf           sample_from_cf.jl:105-135 (secant, then the fall-back ladder)
  sample_log_S_T        heston.jl:278-300
  rand / log_sample     heston.jl:246-276, src/pricing_methods/montecarlo.jl:416-419

and what is RESTATED from published algorithms (third-party in the reference):
  * besseli(ν, z), complex z: scipy.special.ive — the same AMOS library SpecialFunctions.jl wraps
  * NoncentralChisq(d, λ) draw: d > 1: (Z+√λ)² + χ²(d−1); else Poisson(λ/2) mixture of central χ²;
    gamma by Marsaglia–Tsang (2000), Poisson by inversion (mean < 10) or Hörmann's PTRS (1993)
  * find_zero(f, x0, Order2(); atol, maxeval): by default a secant iteration from (x0 + dx, x0),
    dx = h + |x0| h², h = eps^(1/3) (Roots.jl's default secant start), |f| ≤ atol, ≤ maxeval evals
  * find_zero(f, (0, b); xtol, maxeval): by default plain bisection
    — both ALSO as Roots.jl states them (Steffensen guarded by secant; bisection over bit patterns) and with the
    `maxeval` keyword ignored: the fork table above inverse_cdf, decided by the iterate probe of a Julia run
  * draws: Philox4x32-10 keyed by seeds[1], counter = (trajectory index, draw block) — the
    reference's ONE sequential stream (montecarlo.jl:456) is replaced by per-trajectory counters so
    that trajectories are independent of each other and of the sharding.

Draw blocks per trajectory G (counter word 2):  0: normals (Z for log S_T, Z' for the NCχ² shift);
1: uniforms (U for the CDF inversion, U' for the gamma boost);  2+2i / 3+2i: normal / uniform of
rejection iteration i (gamma, PTRS).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
from scipy import special

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
DOM_BK = 2
TWO_PI = 2.0 * math.pi


def _clib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_HERE, "libhh_oracle.so"))
        for name in ("hho_normal_pair", "hho_uniform_pair"):
            f = getattr(_lib, name)
            f.restype = None
            f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                          C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return _lib


class Draws:
    """Counter-based draws of one trajectory."""

    def __init__(self, key: int, G: int):
        self.key, self.c0, self.c1 = key, G & 0xFFFFFFFF, (G >> 32) & 0xFFFFFFFF

    def normals(self, block):
        a, b = C.c_double(), C.c_double()
        _clib().hho_normal_pair(self.key, self.c0, self.c1, block, DOM_BK, C.byref(a), C.byref(b))
        return a.value, b.value

    def uniforms(self, block):
        a, b = C.c_double(), C.c_double()
        _clib().hho_uniform_pair(self.key, self.c0, self.c1, block, DOM_BK, C.byref(a), C.byref(b))
        return a.value, b.value


# ---------------------------------------------------------------------------------------------
# NCχ² (restated; third-party in the reference: Distributions.NoncentralChisq, heston.jl:131)
# ---------------------------------------------------------------------------------------------

def _gamma_mt(shape, dr: Draws, it0):
    """Marsaglia–Tsang for shape ≥ 1; returns (Gamma(shape, 1), next iteration index)."""
    d = shape - 1.0 / 3.0
    c = 1.0 / math.sqrt(9.0 * d)
    it = it0
    while True:
        x, _ = dr.normals(2 + 2 * it)
        u, _ = dr.uniforms(3 + 2 * it)
        it += 1
        v = 1.0 + c * x
        if v <= 0.0:
            continue
        v = v * v * v
        x2 = x * x
        if u < 1.0 - 0.0331 * x2 * x2 or math.log(u) < 0.5 * x2 + d * (1.0 - v + math.log(v)):
            return d * v, it
        if it - it0 > 200:
            return d * v, it


def _gamma(shape, dr: Draws, it0, u_boost):
    if shape >= 1.0:
        return _gamma_mt(shape, dr, it0)
    g, it = _gamma_mt(shape + 1.0, dr, it0)
    return g * u_boost ** (1.0 / shape), it


def _poisson(mu, dr: Draws, it0):
    if mu < 10.0:  # inversion by sequential search
        u, _ = dr.uniforms(3 + 2 * it0)
        p = math.exp(-mu)
        F, k = p, 0
        while u > F and k < 1000:
            k += 1
            p *= mu / k
            F += p
        return k, it0 + 1
    # PTRS, Hörmann (1993)
    smu = math.sqrt(mu)
    b = 0.931 + 2.53 * smu
    a = -0.059 + 0.02483 * b
    inv_alpha = 1.1239 + 1.1328 / (b - 3.4)
    vr = 0.9277 - 3.6224 / (b - 2.0)
    it = it0
    while True:
        u1, V = dr.uniforms(3 + 2 * it)
        it += 1
        U = u1 - 0.5
        us = 0.5 - abs(U)
        k = math.floor((2.0 * a / us + b) * U + mu + 0.43)
        if us >= 0.07 and V <= vr:
            return int(k), it
        if k < 0 or (us < 0.013 and V > us):
            if it - it0 > 200:
                return max(int(k), 0), it
            continue
        if math.log(V) + math.log(inv_alpha) - math.log(a / (us * us) + b) <= \
                -mu + k * math.log(mu) - math.lgamma(k + 1.0):
            return int(k), it
        if it - it0 > 200:
            return int(k), it


def noncentral_chisq(d, lam, dr: Draws):
    _, zshift = dr.normals(0)
    _, u_boost = dr.uniforms(1)
    if d > 1.0:
        g, _ = _gamma(0.5 * (d - 1.0), dr, 0, u_boost)
        s = zshift + math.sqrt(lam)
        return s * s + 2.0 * g
    n, it = _poisson(0.5 * lam, dr, 0)
    g, _ = _gamma(0.5 * d + n, dr, it, u_boost)
    return 2.0 * g


# ---------------------------------------------------------------------------------------------
# the reference's arithmetic
# ---------------------------------------------------------------------------------------------

class LogHestonDistribution:
    """heston.jl:102-111."""

    def __init__(self, S0, V0, kappa, theta, sigma, rho, r, T):
        self.S0, self.V0, self.kappa, self.theta = S0, V0, kappa, theta
        self.sigma, self.rho, self.r, self.T = sigma, rho, r, T


def sample_V_T(dr: Draws, dist):
    """heston.jl:125-133."""
    k, th, s, V0, T = dist.kappa, dist.theta, dist.sigma, dist.V0, dist.T
    d = 4 * k * th / s**2
    lam = 4 * k * math.exp(-k * T) * V0 / (s**2 * (-math.expm1(-k * T)))
    c = s**2 * (-math.expm1(-k * T)) / (4 * k)
    return c * noncentral_chisq(d, lam, dr)


class OutsideReferenceRange(ArithmeticError):
    """besseli(ν, z) under- or overflows in fp64: heston.jl:207's log(besseli(...)) is then ±Inf, the
    characteristic function NaN, and the reference's `while true` series loop (sample_from_cf.jl:80-95)
    never meets its stopping test.  The oracle stops here instead of looping with it."""


def log_besseli(nu, z):
    """log(besseli(ν, z)) for complex z (array ok), formed through the scaled AMOS routine so that
    large |Re z| does not overflow.  Raises OutsideReferenceRange where even the scaled function
    leaves the fp64 range (large orders: I_ν(z) ~ (z/2)^ν / Γ(ν+1) underflows from ν ≈ 300 on)."""
    z = np.asarray(z, dtype=np.complex128)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.log(special.ive(nu, z)) + np.abs(z.real)
    if not np.all(np.isfinite(out)) and np.all(np.isfinite(z)):
        raise OutsideReferenceRange(f"besseli({nu}, z) leaves the fp64 range for |z| in "
                                    f"[{np.min(np.abs(z)):.3g}, {np.max(np.abs(z)):.3g}]")
    return out


class HestonCFIterator:
    """heston.jl:150-176."""

    def __init__(self, VT, dist):
        k, s, V0, T = dist.kappa, dist.sigma, dist.V0, dist.T
        d = 4 * k * dist.theta / s**2
        self.VT, self.dist = VT, dist
        self.nu = 0.5 * d - 1
        self.zeta_k = (-math.expm1(-k * T)) / k
        self.eta_k = k * (1 + math.exp(-k * T)) / (-math.expm1(-k * T))
        nu_k = math.sqrt(V0 * VT) * 4 * k * math.exp(-0.5 * k * T) / s**2 / (-math.expm1(-k * T))
        self.logI_k = float(log_besseli(self.nu, nu_k).real)

    def chf_block(self, a, theta_prev):
        """evaluate_chf (heston.jl:184-212) for a vector of points `a` visited IN ORDER, threading
        the unwrapped angle from one point to the next.  Returns (ϕ[], θ_unwrapped of the last)."""
        d = self.dist
        k, s, V0, T, VT, nu = d.kappa, d.sigma, d.V0, d.T, self.VT, self.nu
        a = np.asarray(a, dtype=np.float64)
        g = np.sqrt(k**2 - 2 * s**2 * a * 1j)
        e = np.exp(-g * T)
        zeta_g = (1 - e) / g
        eta_g = g * (1 + e) / (1 - e)
        nu_g = math.sqrt(V0 * VT) * 4 * g * np.exp(-0.5 * g * T) / s**2 / (1 - e)
        first = np.exp(-0.5 * (g - k) * T) * (self.zeta_k / zeta_g)
        second = np.exp((V0 + VT) / s**2 * (self.eta_k - eta_g))
        th = np.angle(nu_g)
        thu = np.empty_like(th)
        prev = theta_prev
        for i in range(len(th)):  # heston.jl:199-205
            if math.isnan(prev):
                cur = th[i]
            else:
                dlt = th[i] - prev
                dlt -= TWO_PI * np.round(dlt / TWO_PI)
                cur = prev + dlt
            thu[i] = cur
            prev = cur
        z_unw = np.abs(nu_g) * np.exp(1j * thu)  # abs(νγ)·cis(θ_unwrapped)
        logI_g = log_besseli(nu, z_unw) + 1j * nu * (thu - th)
        phi = first * second * np.exp(logI_g - self.logI_k)
        return phi, prev


def moments_from_cf(it: HestonCFIterator, h=1e-2):
    """sample_from_cf.jl:50-64."""
    phi, _ = it.chf_block([h, 0.0, -h], float("nan"))
    pp, p0, pm = phi
    first = (pp - pm) / (2 * h)
    second = (pp - 2 * p0 + pm) / h**2
    mean = (-1j * first).real
    var = (-second - mean**2).real
    return mean, var


class CdfCounter:
    terms = 0
    last_len = 0   # series length of the last CDF evaluation (the same for every x of a trajectory)


def cdf_from_cf(it: HestonCFIterator, x, h, cf_tol=1e-3, block=32, counter=None):
    """sample_from_cf.jl:75-96."""
    if x < 0:
        return 0.0
    result = h * x / math.pi
    pref = 2 / math.pi
    prev = float("nan")
    j0 = 1
    while j0 < 10**9:
        j = np.arange(j0, j0 + block, dtype=np.float64)
        a = h * j
        phi, prev_new = it.chf_block(a, prev)
        stop = np.abs(phi) / j < math.pi * cf_tol / 2
        n = int(np.argmax(stop)) + 1 if stop.any() else block
        result += float(np.sum((pref * np.sin(a[:n] * x) / j[:n] * phi[:n].real)))
        if counter is not None:
            counter.terms += n
            counter.last_len = j0 - 1 + n
        if stop.any():
            break
        prev = prev_new
        j0 += block
    return result


def _cdf_seq(it, x, h, cf_tol, counter):
    """Same series summed term by term in the reference's order (used to pin the blocked form)."""
    if x < 0:
        return 0.0
    result = h * x / math.pi
    prev = float("nan")
    for j in range(1, 10**6):
        a = h * j
        phi, prev = it.chf_block([a], prev)
        result += 2 / math.pi * math.sin(a * x) / j * phi[0].real
        if counter is not None:
            counter.terms += 1
        if abs(phi[0]) / j < math.pi * cf_tol / 2:
            break
    return result


SECANT_H = np.finfo(np.float64).eps ** (1.0 / 3.0)


DEC_BISECT, DEC_MAXGUESS, DEC_ITERS_SHIFT = 1 << 8, 2 << 8, 16  # decision word: hh_bk_decisions (hedgehog_mc.h)

# ---- the forks of `find_zero` ------------------------------------------------------------------------------------
# inverse_cdf (sample_from_cf.jl:105-135) calls Roots.jl twice — `find_zero(func, x0, Order2(); atol, maxeval)` and
# `find_zero(func, (0, max_guess); xtol, maxeval)` — and Roots (2.2.6 in the reference's Project.toml) is not under
# /root/reference.  Three things about those calls cannot be decided from the reference's text, so each exists here
# in both readings (and in the kernels: hh_config.bk_root_form / bk_bracket_form / bk_caps); the defaults are what
# this repository has shipped since round 1, and ONE run of julia/parity_replay.jl (its `bk_root_probe` case: every x
# the reference's own inverse_cdf asks of its CDF) decides between them — tools/check_reference_replay.py prints the
# verdict.
#   root_form     0  SECANT: the plain secant iteration from (x0 + dx, x0)
#                 1  ORDER2: Roots' `Order2()` as its documentation and source (as remembered; the probe decides)
#                    state it — a Steffensen step (two evaluations: at x1 - sgn·f1, then at the new iterate), replaced
#                    by a secant step while f is large (1000·|f(x1)| > max(1, |x1|)); convergence when |f| <= max(atol,
#                    4 eps |x|), or when the iterates stall (|x1 - x0| <= max(eps, eps·max|x|)) with |f| below the cube
#                    root of that tolerance
#   bracket_form  0  MIDPOINT: arithmetic bisection, stopped at width <= atol (what `xtol = atol` asks for)
#                 1  ROOTS: Roots' `Bisection()` for Float64 — the midpoint taken over the BIT PATTERNS of the two ends
#                    (`_middle`), run until the ends are adjacent floats (~62 steps from (0, b)), which is what happens
#                    if `xtol` is not a keyword Roots 2 knows (its tolerances are xatol / xrtol) and is ignored
#   caps          0  AS_WRITTEN: `maxeval` caps the evaluations (secant) / iterations (bisection) as the reference's
#                    author meant
#                 1  ROOTS_DEFAULT: `maxeval` is not a keyword Roots 2 knows (`maxiters`, alias `maxevals`) and is
#                    ignored: Order2 stops after Roots' own 40 steps, the bisection only at the last bit
ROOT_SECANT, ROOT_ORDER2 = 0, 1
BRACKET_MIDPOINT, BRACKET_ROOTS = 0, 1
CAPS_AS_WRITTEN, CAPS_ROOTS_DEFAULT = 0, 1
ROOTS_MAXITERS = 40
_EPS = float(np.finfo(np.float64).eps)


def _sign(x):
    return int(x > 0) - int(x < 0)


def roots_middle(a, b):
    """Roots.jl `_middle(x::Float64, y::Float64)`: halfway between the two numbers' bit patterns (same sign, finite);
    0.0 when the signs differ."""
    if not (math.isfinite(a) and math.isfinite(b)):
        return a + b
    if _sign(a) != _sign(b) and a != 0.0 and b != 0.0:
        return 0.0
    negate = a < 0 or b < 0
    ia = int(np.float64(abs(a)).view(np.uint64))
    ib = int(np.float64(abs(b)).view(np.uint64))
    m = float(np.uint64((ia + ib) >> 1).view(np.float64))
    return -m if negate else m


def _secant_search(func, guess, atol, max_evals):
    """-> (converged, x, f(x), evaluations)"""
    x1 = guess
    x0 = x1 + SECANT_H + abs(x1) * SECANT_H * SECANT_H
    f0, f1 = func(x0), func(x1)
    evals = 2
    while True:
        if abs(f1) <= atol:
            return True, x1, f1, evals
        if evals >= max_evals or f1 == f0:
            return False, x1, f1, evals
        x2 = x1 - f1 * (x1 - x0) / (f1 - f0)
        if not math.isfinite(x2):
            return False, x1, f1, evals
        x0, f0 = x1, f1
        x1, f1 = x2, func(x2)
        evals += 1


def _order2_search(func, guess, atol, max_steps):
    """Roots.Order2 (see the fork table above) -> (converged, x, f(x), evaluations)"""
    x1 = guess
    x0 = x1 + SECANT_H + abs(x1) * SECANT_H * SECANT_H   # Roots' default second point of a secant-type method
    f0, f1 = func(x0), func(x1)
    evals, steps = 2, 0
    while True:
        if not (math.isfinite(x1) and math.isfinite(f1)):
            return False, x1, f1, evals
        tol = max(atol, abs(x1) * 4.0 * _EPS)
        if abs(f1) <= tol:                                                   # :f_converged
            return True, x1, f1, evals
        if abs(x1 - x0) <= max(_EPS, max(abs(x1), abs(x0)) * _EPS):         # :x_converged — accepted if f is small-ish
            return abs(f1) <= float(np.cbrt(tol)), x1, f1, evals
        if steps >= max_steps:
            return False, x1, f1, evals
        if 1000.0 * abs(f1) > max(1.0, abs(x1)):                             # guarded: a secant step
            d = f1 * (x1 - x0) / (f1 - f0) if f1 != f0 else float("inf")
            if not math.isfinite(d):
                return False, x1, f1, evals
            x2 = x1 - d
            f2 = func(x2)
            evals += 1
        else:                                                                # a Steffensen step
            sgn = _sign((f1 - f0) / (x1 - x0)) if x1 != x0 else 0
            fs = func(x1 - sgn * f1)
            evals += 1
            d = -sgn * f1 * f1 / (fs - f1) if fs != f1 else float("inf")
            if not math.isfinite(d):
                return False, x1, f1, evals
            x2 = x1 - d
            f2 = func(x2)
            evals += 1
        x0, f0, x1, f1 = x1, f1, x2, f2
        steps += 1


def inverse_cdf(cdf, u, initial_guess, max_guess, atol=1e-4, maxiter_newton=10,
                maxiter_bisection=100, stats=None, trace=None, root_form=ROOT_SECANT,
                bracket_form=BRACKET_MIDPOINT, caps=CAPS_AS_WRITTEN, xs=None):
    """sample_from_cf.jl:105-135.  trace (a list): receives the trajectory's decision word —
    evaluations of the first search | branch << 8 | bisection iterations << 16 — what the search DID.
    xs (a list): receives every abscissa the CDF is asked for, in order (the exchange format of the iterate probe:
    julia/parity_replay.jl `bk_root_probe`)."""
    def func(y):
        if xs is not None:
            xs.append(float(y))
        return cdf(y) - u
    if root_form == ROOT_ORDER2:
        ok, sol, fsol, evals = _order2_search(func, initial_guess, atol,
                                              maxiter_newton if caps == CAPS_AS_WRITTEN else ROOTS_MAXITERS)
    else:
        ok, sol, fsol, evals = _secant_search(func, initial_guess, atol,
                                              maxiter_newton if caps == CAPS_AS_WRITTEN else 2 + ROOTS_MAXITERS)
    evals = min(evals, 0xff)
    # (the reference evaluates func(sol) once more for its `abs(func(sol)) > atol` test: the same number again)
    if ok and not (sol < 0 or abs(fsol) > atol):
        if trace is not None:
            trace.append(evals)
        return sol
    if stats is not None:
        stats["newton_fail"] += 1
    # --- fall-back ladder (:124-133)
    fa, fb = func(0.0), func(max_guess)
    if fa * fb > 0:
        if stats is not None:
            stats["maxguess"] += 1
        if trace is not None:
            trace.append(evals | DEC_MAXGUESS)
        return max_guess
    if stats is not None:
        stats["bisect"] += 1
    a, b = 0.0, max_guess
    iters, out = 0, None
    if bracket_form == BRACKET_ROOTS:
        cap = maxiter_bisection if caps == CAPS_AS_WRITTEN else 4096
        while iters < cap:
            mid = roots_middle(a, b)
            if not (a < mid < b):          # the ends are adjacent floats: nothing between them
                break
            fm = func(mid)
            iters += 1
            if fm == 0.0:
                out = mid
                break
            if _sign(fa) * _sign(fm) < 0:
                b, fb = mid, fm
            else:
                a, fa = mid, fm
        if out is None:
            out = a if abs(fa) < abs(fb) else b  # the end with the smaller residual
    else:
        for _ in range(maxiter_bisection if caps == CAPS_AS_WRITTEN else 4096):
            mid = 0.5 * (a + b)
            fm = func(mid)
            iters += 1
            if fm == 0.0:
                out = mid
                break
            if (fm < 0) == (fa < 0):
                a, fa = mid, fm
            else:
                b = mid
            if b - a <= atol:
                break
    if trace is not None:
        trace.append(evals | DEC_BISECT | ((iters & 0xff) << DEC_ITERS_SHIFT))
    return 0.5 * (a + b) if out is None else out


def sample_from_cf(u, it: HestonCFIterator, n=5, cf_tol=1e-3, atol=1e-4, moment_h=1e-2,
                   maxiter_newton=10, maxiter_bisection=100, stats=None, counter=None,
                   sequential=False, trace=None, root_form=ROOT_SECANT, bracket_form=BRACKET_MIDPOINT,
                   caps=CAPS_AS_WRITTEN, xs=None, setup=None):
    """sample_from_cf.jl:27-41 (the uniform u is supplied by the caller)."""
    mean, variance = moments_from_cf(it, moment_h)
    s2 = max(variance, 1e-12)
    normal_sample = mean + math.sqrt(s2) * special.ndtri(u)
    initial_guess = normal_sample if normal_sample > 0 else mean * 0.01
    max_guess = mean + 11 * math.sqrt(s2)
    h = math.pi / (mean + n * math.sqrt(s2))
    if sequential:
        cdf = lambda x: _cdf_seq(it, x, h, cf_tol, counter)
    else:
        cdf = lambda x: cdf_from_cf(it, x, h, cf_tol, counter=counter)
    if setup is not None:  # what the search starts from (the iterate probe exports it)
        setup.update(initial_guess=initial_guess, max_guess=max_guess, h=h)
    return inverse_cdf(cdf, u, initial_guess, max_guess, atol, maxiter_newton, maxiter_bisection,
                       stats, trace, root_form, bracket_form, caps, xs)


def sample_log_S_T(V_T, integral_V, Z, d):
    """heston.jl:278-300."""
    mu = math.log(d.S0) + d.r * d.T - 0.5 * integral_V + \
        (d.rho / d.sigma) * (V_T - d.V0 - d.kappa * d.theta * d.T + d.kappa * integral_V)
    sigma2 = (1 - d.rho**2) * integral_V
    return mu + math.sqrt(sigma2) * Z


def rand_path(dist, key: int, G: int, stats=None, counter=None, sequential=False, draws=None, trace=None,
              **kw):
    """heston.jl:246-259 for trajectory G -> (log S_T, V_T, ∫V).  draws = (V_T, u, Z): the
    trajectory's three draws supplied by the caller (HH_NOISE_REPLAY) instead of drawn here."""
    if draws is not None:
        V_T, u, Z = (float(x) for x in draws)
    else:
        dr = Draws(key, G)
        V_T = sample_V_T(dr, dist)
        u, _ = dr.uniforms(1)
        Z, _ = dr.normals(0)
    it = HestonCFIterator(V_T, dist)
    I = sample_from_cf(u, it, stats=stats, counter=counter, sequential=sequential, trace=trace, **kw)
    return sample_log_S_T(V_T, I, Z, dist), V_T, I


def mc_solve(S0, V0, kappa, theta, sigma, rho, r, T, strike, cp, discount, n_paths, seed0,
             path_offset=0, replay=None, **kw):
    """solve(prob, MonteCarlo(HestonDynamics(), HestonBroadieKaya(), cfg)), montecarlo.jl:454-493.
    Returns dict(price, std_error, terminal, V_T, integral_V, stats, cf_terms, decisions, series_len) —
    decisions / series_len: what each trajectory's root search did (the words of hh_bk_decisions)."""
    dist = LogHestonDistribution(S0, V0, kappa, theta, sigma, rho, r, T)
    stats = {"newton_fail": 0, "bisect": 0, "maxguess": 0}
    counter = CdfCounter()
    counter.terms = 0
    logS = np.empty(n_paths)
    VT = np.empty(n_paths)
    IV = np.empty(n_paths)
    decisions, series_len = [], np.zeros(n_paths, dtype=np.uint32)
    for i in range(n_paths):
        dr = None if replay is None else (replay[0][i], replay[1][i], replay[2][i])  # [V_T | u | Z]
        logS[i], VT[i], IV[i] = rand_path(dist, int(seed0), path_offset + i, stats, counter,
                                          draws=dr, trace=decisions, **kw)
        series_len[i] = counter.last_len
    S = np.exp(logS)  # final_sample(law, sample, NoVR) = exp.(sample)  montecarlo.jl:384
    pay = np.maximum(cp * (S - strike), 0.0)
    price = discount * pay.mean()
    se = discount * pay.std(ddof=1) / math.sqrt(n_paths) if n_paths > 1 else 0.0
    return dict(price=price, std_error=se, terminal=S, V_T=VT, integral_V=IV, stats=stats,
                cf_terms=counter.terms, decisions=np.array(decisions, dtype=np.uint32), series_len=series_len)


def exact_grid(S0, V0, kappa, theta, sigma, rho, r, T, n_steps, seeds, stats=None, counter=None,
               **kw):
    """Per-date exact Heston paths: the NoiseProblem of sde_problem(::HestonDynamics,
    ::HestonBroadieKaya) (montecarlo.jl:209-231) on HestonNoise (heston.jl:82-91), stepped with
    dt = T / n_steps.  Each step: S, V = exp(W[1]), W[2]; (log S', V') ~ LogHestonDistribution(S, V,
    κ, θ, σ, ρ, r, dt) (heston.jl:84-86), where sample_log_S_T takes log(S) again (:289).
    Trajectory i draws from Philox keyed by seeds[i] (montecarlo.jl:331), counter = transition index.
    Returns (spot[(n_steps+1), n], var[(n_steps+1), n]); spot rows are exp(log S)."""
    n = len(seeds)
    dt = T / n_steps
    spot = np.empty((n_steps + 1, n))
    var = np.empty((n_steps + 1, n))
    spot[0], var[0] = S0, V0
    for i in range(n):
        for k in range(n_steps):
            dist = LogHestonDistribution(spot[k, i], var[k, i], kappa, theta, sigma, rho, r, dt)
            logS, VT, _ = rand_path(dist, int(seeds[i]), k, stats, counter, **kw)
            spot[k + 1, i], var[k + 1, i] = math.exp(logS), VT
    return spot, var
