"""Host mirror of /root/reference/src/pricing_methods/montecarlo.jl — types and `solve`.

Same names, argument meaning and error behaviour as the reference; the body of `solve` is one
call through the C-ABI (include/hedgehog_mc.h) into the HIP kernels.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any

import numpy as np

from . import _ffi
from .dates import yearfrac
from .dual import Dual, n_partials, partials_of, value_of
from .domain import (BlackScholesInputs, European, HestonInputs, MonteCarloSolution, PricingProblem,
                    Spot, VanillaOption, _DeviceSamples, df, get_vol, zero_rate)


# ---- montecarlo.jl:8-43 ----
class PriceDynamics: pass
class LognormalDynamics(PriceDynamics): pass
class HestonDynamics(PriceDynamics): pass
class VarianceReductionStrategy: pass
class NoVarianceReduction(VarianceReductionStrategy): pass
class Antithetic(VarianceReductionStrategy): pass
# ---- montecarlo.jl:86-115 ----
class SimulationStrategy: pass
class EulerMaruyama(SimulationStrategy): pass
class ExactSimulation(SimulationStrategy): pass
class HestonBroadieKaya(ExactSimulation): pass
class BlackScholesExact(ExactSimulation): pass

for _c in (LognormalDynamics, HestonDynamics, NoVarianceReduction, Antithetic, EulerMaruyama,
           HestonBroadieKaya, BlackScholesExact):
    _c.__eq__ = lambda a, b: type(a) is type(b)
    _c.__hash__ = lambda a: hash(type(a).__name__)
    _c.__repr__ = lambda a: type(a).__name__ + "()"


class AbstractPricingMethod: pass


class SimulationConfig:
    """montecarlo.jl:58-79.  `SimulationConfig(trajectories; steps=1, seeds=nothing,
    variance_reduction=NoVarianceReduction())`; too few seeds -> ValueError (ArgumentError there)."""

    def __init__(self, trajectories, steps=1, seeds=None, variance_reduction=None):
        if variance_reduction is None:
            variance_reduction = NoVarianceReduction()
        if seeds is None:  # montecarlo.jl:77: rand(UInt64, trajectories)
            seeds = np.random.default_rng().integers(0, 2**64, size=int(trajectories),
                                                     dtype=np.uint64)
        if isinstance(seeds, np.ndarray) and seeds.dtype == np.uint64 and not seeds.flags.writeable \
                and seeds.flags.c_contiguous:
            pass  # already a frozen seed vector (replace(), slices of another config): shared
        else:     # the config's OWN copy, read-only: its device copies (below) can never go stale
            seeds = np.array(np.asarray(seeds).astype(np.uint64, copy=False), dtype=np.uint64, order="C",
                             copy=True)
            seeds.flags.writeable = False
        if len(seeds) < trajectories:  # montecarlo.jl:65-66
            raise ValueError(f"Number of seeds ({len(seeds)}) must be ≥ number of trajectories "
                             f"({trajectories}).")
        self.trajectories = int(trajectories)
        self.steps = int(steps)
        self.variance_reduction = variance_reduction
        self.seeds = seeds
        self._fingerprint = 0

    def device_seeds(self, ctx):
        """Device address of the seed vector in ctx's cache (hh_seeds_cache: content-addressed, bounded, freed
        with the context), asked for right before every solve: repeated solves (Greeks by finite differences,
        calibration loops) do not move the 8 MB per 10^6 trajectories again.  `seeds` is this config's own
        read-only copy, so its fingerprint is computed once."""
        if not self._fingerprint:
            self._fingerprint = ctx.lib.hh_seeds_fingerprint(self.seeds.ctypes.data, self.seeds.size)
        return ctx.seeds_on_device(self.seeds, self._fingerprint)

    def replace(self, **kw):
        """Accessors' `@set config.seeds = …` (test/agreement/montecarlo_heston.jl:87)."""
        d = dict(trajectories=self.trajectories, steps=self.steps, seeds=self.seeds,
                 variance_reduction=self.variance_reduction)
        d.update(kw)
        return SimulationConfig(**d)


@dataclass(frozen=True)
class MonteCarlo(AbstractPricingMethod):
    """montecarlo.jl:127-131.  `em_split` / `compat_sqrt_alpha` / `device` are build options.
    em_split=True is the split-step form SURVEY §8a-4 attributes to StochasticDiffEq's EM() (the
    reference's behaviour as far as it can be read without running it).  compat_sqrt_alpha=False is
    NOT the reference's behaviour: it is the CORRECTED exact-lognormal drift (r - σ²/2)·T; the
    reference computes (r - σ²/2)·√T (montecarlo.jl:302, quirk Q1), which compat_sqrt_alpha=True
    reproduces bug for bug — use that for parity runs at T != 1 (at T = 1 both coincide)."""
    dynamics: Any
    strategy: Any
    config: SimulationConfig
    em_split: bool = True
    compat_sqrt_alpha: bool = False
    device: int = 0
    devices: Any = None  # e.g. range(8): ONE solve sharded over these GPUs inside the library (hh_mgpu_solve)


class MethodError(TypeError):
    """Julia's MethodError: no `solve` / `sde_problem` method for this combination."""


# ------------------------------------------------------------------------------------------------

def _model_and_config(prob: PricingProblem, method: MonteCarlo, n_paths=None, path_offset=0):
    """Resolve dates/curves exactly as sde_problem / marginal_law / solve do and pack the C structs.

    Returns (hh_model, hh_config, keepalive, P, discount)."""
    payoff, m = prob.payoff, prob.market_inputs
    # solve's signature: VanillaOption{…, European, C, Spot}  (montecarlo.jl:479)
    if not (isinstance(payoff, VanillaOption) and isinstance(payoff.exercise_style, European)
            and isinstance(payoff.underlying, Spot)):
        raise MethodError("solve(::PricingProblem, ::MonteCarlo) needs a European VanillaOption on Spot")
    dyn, strat, cfg = method.dynamics, method.strategy, method.config
    euler = isinstance(strat, EulerMaruyama)

    if isinstance(dyn, LognormalDynamics) and isinstance(m, BlackScholesInputs) and \
            isinstance(strat, (EulerMaruyama, BlackScholesExact)):
        dynamics = _ffi.HH_LOGNORMAL
        strategy = _ffi.HH_EULER_MARUYAMA if euler else _ffi.HH_EXACT_LAW
        sigma, V0, kappa, theta, rho = get_vol(m.sigma, None, None), 0.0, 0.0, 0.0, 0.0
    elif isinstance(dyn, HestonDynamics) and isinstance(m, HestonInputs) and \
            isinstance(strat, (EulerMaruyama, HestonBroadieKaya)):
        dynamics = _ffi.HH_HESTON
        strategy = _ffi.HH_EULER_MARUYAMA if euler else _ffi.HH_BROADIE_KAYA
        sigma, V0, kappa, theta, rho = m.σ, m.V0, m.κ, m.θ, m.ρ
    else:
        raise MethodError(f"no sde_problem / marginal_law for {type(dyn).__name__} + "
                          f"{type(strat).__name__} on {type(m).__name__}")

    if euler:
        T = yearfrac(m.referenceDate, payoff.expiry)          # montecarlo.jl:173,197
        r_drift = zero_rate(m.rate, 0.0)                      # montecarlo.jl:176,200
    else:
        T = yearfrac(m.rate.reference_date, payoff.expiry)    # montecarlo.jl:301,317
        r_drift = zero_rate(m.rate, payoff.expiry)            # montecarlo.jl:299,318
    discount = df(m.rate, payoff.expiry)                      # montecarlo.jl:489

    scal = dict(S0=m.spot, V0=V0, kappa=kappa, theta=theta, sigma=sigma, r_drift=r_drift,
                discount=discount, strike=payoff.strike)
    if isinstance(rho, Dual):
        raise MethodError("differentiation with respect to ρ is not supported "
                          "(the reference cannot push a Dual through svd(Γ), heston.jl:18-20)")
    P = n_partials(*scal.values())
    if P > _ffi.HH_MAX_PARTIALS:
        raise ValueError(f"at most {_ffi.HH_MAX_PARTIALS} partials per solve")
    keep = []
    model = _ffi.hh_model()
    for k, v in scal.items():
        setattr(model, k, value_of(v))
        arr = _ffi.seed_array(partials_of(v, P), P) if P else None
        if arr is not None:
            keep.append(arr)
            setattr(model, "d" + k, C.cast(arr, C.POINTER(C.c_double)))
    model.rho = float(rho)
    model.T = float(T)
    model.cp = payoff.call_put()

    c = _ffi.hh_config()
    c.dynamics, c.strategy = dynamics, strategy
    c.antithetic = int(isinstance(cfg.variance_reduction, Antithetic))
    c.em_split = int(method.em_split)
    c.compat_sqrt_alpha = int(method.compat_sqrt_alpha)
    c.noise_mode = _ffi.HH_NOISE_GENERATE
    c.n_steps = cfg.steps
    c.n_partials = P
    c.n_paths = cfg.trajectories if n_paths is None else n_paths
    c.path_offset = path_offset
    return model, c, keep, P, discount


@dataclass(frozen=True)
class NormalLaw:
    """Distributions.Normal(μ, σ) as far as the reference's callers use it (mean, std, var)."""
    mu: Any
    sigma: Any

    def mean(self):
        return self.mu

    def std(self):
        return self.sigma

    def var(self):
        return self.sigma * self.sigma


def marginal_law(prob: PricingProblem, dynamics, t, compat_sqrt_alpha: bool = True):
    """montecarlo.jl:293-303: the law of log S_t under Black–Scholes dynamics,
    Normal(log S0 + (r − σ²/2)·√α, σ·√α) with α = yearfrac(rate.reference_date, t) — AS WRITTEN THERE (quirk
    Q1: the mean carries √α where the lognormal law has α; identical at α = 1).  That is what a caller of
    the reference's `marginal_law` gets, so it is this function's default; `compat_sqrt_alpha=False` gives
    the corrected mean (r − σ²/2)·α, which is what `solve(…, MonteCarlo(…, BlackScholesExact(), …))` samples
    from unless its method says compat_sqrt_alpha=True.  Host arithmetic only (dual numbers pass through).
    The Heston law (:310-320, LogHestonDistribution) has no closed form to return: sample it with
    `solve(prob, MonteCarlo(HestonDynamics(), HestonBroadieKaya(), config)).ensemble`."""
    from .dual import dsqrt, dlog
    m = prob.market_inputs
    if not (isinstance(dynamics, LognormalDynamics) and isinstance(m, BlackScholesInputs)):
        raise MethodError("marginal_law: LognormalDynamics on BlackScholesInputs (the Heston law is sampled, not returned)")
    rate = zero_rate(m.rate, t)
    sigma = get_vol(m.sigma, None, None)
    alpha = yearfrac(m.rate.reference_date, t)
    ra = dsqrt(alpha)
    return NormalLaw(dlog(m.spot) + (rate - sigma * sigma / 2) * (ra if compat_sqrt_alpha else alpha), sigma * ra)


def _price_from(res, discount, P):
    if P == 0:
        return res.price
    return Dual(res.price, tuple(res.dprice[k] for k in range(P)))


def solve_montecarlo(prob: PricingProblem, method: MonteCarlo, ensemble: bool = True,
                     replay=None, replay_layout=_ffi.HH_REPLAY_PATH_MAJOR) -> MonteCarloSolution:
    """solve(prob, method::MonteCarlo) — montecarlo.jl:478-493.

    `replay` (optional, build extension): Wiener increments to consume instead of drawing them —
    numpy array [path][step][comp] (or tile-major), the noise-replay parity mode of DESIGN.md."""
    model, c, keep, P, discount = _model_and_config(prob, method)
    cfg = method.config
    if method.devices is not None:
        return _solve_multi_gpu(prob, method, model, c, P, discount, ensemble, replay, replay_layout)
    ctx = _ffi.get_context(method.device)
    c.seeds, c.seeds_on_device = cfg.device_seeds(ctx), 1
    c.seeds_len = cfg.seeds.size
    if replay is not None:
        replay = np.ascontiguousarray(replay, dtype=np.float64)
        c.noise_mode = _ffi.HH_NOISE_REPLAY
        c.replay_layout = replay_layout
        c.replay = replay.ctypes.data
        c.replay_len = replay.size
    anti = bool(c.antithetic)
    term_dev, fetch = None, None
    n_paths = int(c.n_paths)
    if ensemble:  # the samples stay on the device until MonteCarloSolution.ensemble is read
        term_dev = _ffi.DeviceBuffer(ctx, 8 * n_paths * (2 if anti else 1), pooled=True)
        c.terminal_on_device = 1

        def fetch(buf=term_dev):  # montecarlo.jl:398-402: vector, or tuple of two for antithetic
            term = buf.download(np.empty(n_paths * (2 if anti else 1), dtype=np.float64))
            buf.recycle()
            return (term[:n_paths], term[n_paths:]) if anti else term
    res = _ffi.hh_result()
    ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(model), C.byref(c), C.byref(res),
                                  term_dev.ptr if term_dev else None))
    del keep
    ens = _DeviceSamples(fetch, ctx) if fetch is not None else None
    return MonteCarloSolution(prob, method, _price_from(res, discount, P), ens,
                              std_error=res.std_error, result=res)


def solve_montecarlo_many(probs, method: MonteCarlo, replay=None, replay_layout=_ffi.HH_REPLAY_PATH_MAJOR):
    """Several problems under ONE method on the same draws — the solves a bumped Greek is made of
    (compute_fd_derivative, greeks_problem.jl:279-303; the second-order stencils :396-422) — in one pass of
    the kernels (hh_mc_solve_multi): result k is solve(probs[k], method) bit for bit, without its ensemble.
    Returns None when the problems cannot share a pass (dual numbers in one of them, more than HH_MAX_MODELS
    problems): the caller then solves them one by one, as the reference does."""
    if not 1 < len(probs) <= _ffi.HH_MAX_MODELS:
        return None
    packed = [_model_and_config(p, method) for p in probs]
    if any(P for _, _, _, P, _ in packed):
        return None
    c = packed[0][1]
    cfg = method.config
    if method.devices is not None:  # the trajectories sharded over several GPUs by the library (hh_mgpu_solve_multi)
        if replay is not None:
            return None
        mg = _ffi.get_multi_gpu(tuple(method.devices))
        c.seeds, c.seeds_len = cfg.seeds.ctypes.data, cfg.seeds.size
        res = mg.solve_multi([m for m, _, _, _, _ in packed], c)
        return [MonteCarloSolution(p, method, res[k].price, None, std_error=res[k].std_error, result=res[k])
                for k, p in enumerate(probs)]
    ctx = _ffi.get_context(method.device)
    c.seeds, c.seeds_on_device, c.seeds_len = cfg.device_seeds(ctx), 1, cfg.seeds.size
    if replay is not None:
        replay = np.ascontiguousarray(replay, dtype=np.float64)
        c.noise_mode, c.replay_layout = _ffi.HH_NOISE_REPLAY, replay_layout
        c.replay, c.replay_len = replay.ctypes.data, replay.size
    models = (_ffi.hh_model * len(probs))(*[m for m, _, _, _, _ in packed])
    res = (_ffi.hh_result * len(probs))()
    ctx.check(ctx.lib.hh_mc_solve_multi(ctx.handle, models, len(probs), C.byref(c), res, None))
    return [MonteCarloSolution(p, method, res[k].price, None, std_error=res[k].std_error, result=res[k])
            for k, p in enumerate(probs)]


def _solve_multi_gpu(prob, method, model, c, P, discount, ensemble, replay, replay_layout):
    """solve(prob, method) with `method.devices`: the trajectories sharded over those GPUs by ONE
    library call from this thread (hh_mgpu_solve: contiguous ranges, one RCCL all-reduce of the
    accumulator vector or the host's ordered sum) — still one `solve`, as montecarlo.jl:478-493."""
    cfg = method.config
    mg = _ffi.get_multi_gpu(tuple(method.devices))
    c.seeds, c.seeds_len = cfg.seeds.ctypes.data, cfg.seeds.size
    if replay is not None:
        replay = np.ascontiguousarray(replay, dtype=np.float64)
        c.noise_mode, c.replay_layout = _ffi.HH_NOISE_REPLAY, replay_layout
        c.replay, c.replay_len = replay.ctypes.data, replay.size
    n, anti = int(c.n_paths), bool(c.antithetic)
    term = np.empty(n * (2 if anti else 1), dtype=np.float64) if ensemble else None
    res = mg.solve(model, c, term)
    ens = None if term is None else ((term[:n], term[n:]) if anti else term)
    return MonteCarloSolution(prob, method, _price_from(res, discount, P), ens,
                              std_error=res.std_error, result=res)
