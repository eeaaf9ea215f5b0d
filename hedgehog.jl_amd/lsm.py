"""Host mirror of /root/reference/src/pricing_methods/least_squares_montecarlo.jl:
`LSM` (:12-34), `LSMSolution` (src/solutions/pricing_solutions.jl) and
`solve(::PricingProblem{VanillaOption{…,American,…}}, ::LSM)` (:99-136), running on the HIP path
(`hh_lsm_solve`).  Supported path sources: LognormalDynamics + BlackScholesExact, the pair the
reference's LSM is used with (test/agreement/american_options.jl), and HestonDynamics +
HestonBroadieKaya (per-date exact transitions, montecarlo.jl:209-231), regressed on the SPOT rows.
For every log-state problem (the Euler ones, and HestonNoise) the reference's extract_spot_grid
(:47-85) hands the regression log-prices; that as-run behaviour is not reproduced: the Euler
combinations raise MethodError here, and the exact Heston paths are exponentiated first."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from . import _ffi
from .dates import MILLISECONDS_IN_YEAR_365, yearfrac
from .domain import (American, BlackScholesInputs, HestonInputs, PricingProblem, VanillaOption, df,
                     get_vol, zero_rate)
from .montecarlo import (AbstractPricingMethod, Antithetic, BlackScholesExact, HestonBroadieKaya,
                         HestonDynamics, LognormalDynamics, MethodError, MonteCarlo)


@dataclass(frozen=True)
class LSM(AbstractPricingMethod):
    """least_squares_montecarlo.jl:12-34: LSM(mc_method, degree) or
    LSM(dynamics, strategy, config, degree)."""
    mc_method: MonteCarlo
    degree: int

    def __init__(self, *args):
        if len(args) == 2:
            mc, degree = args
        elif len(args) == 4:
            mc, degree = MonteCarlo(args[0], args[1], args[2]), args[3]
        else:
            raise TypeError("LSM(mc_method, degree) or LSM(dynamics, strategy, config, degree)")
        object.__setattr__(self, "mc_method", mc)
        object.__setattr__(self, "degree", int(degree))


@dataclass(frozen=True)
class LSMSolution:
    """pricing_solutions.jl LSMSolution: stopping_info = (times, values) arrays, spot_paths the
    (nsteps+1, npaths) matrix (None unless requested)."""
    problem: Any
    method: Any
    price: float
    stopping_info: Any
    spot_paths: Any
    std_error: float = field(default=float("nan"), compare=False)
    result: Any = field(default=None, compare=False, repr=False)


def _lsm_structs(prob: PricingProblem, mc: MonteCarlo):
    """hh_model / hh_config of the path source behind an LSM solve (or a bare path simulation)."""
    payoff, m = prob.payoff, prob.market_inputs
    cfg = mc.config
    from .dual import n_partials
    inputs = [getattr(m, a) for a in ("spot", "V0", "κ", "θ", "σ", "ρ") if hasattr(m, a)] + [payoff.strike]
    if isinstance(m, BlackScholesInputs):
        inputs.append(get_vol(m.sigma, None, None))
    if n_partials(*inputs, zero_rate(m.rate, 0.0)) > 0:
        # stopping decisions are not differentiable and the kernels carry no partials along full paths:
        # a Dual must not be dropped silently (float(Dual) would) — FiniteDifference works on plain solves
        raise MethodError("ForwardAD through the full-path kernels (LSM, exact Heston paths) is not carried; "
                          "use FiniteDifference")
    T = yearfrac(m.referenceDate, payoff.expiry)                       # :104, montecarlo.jl:147,219
    model = _ffi.hh_model()
    c = _ffi.hh_config()
    if isinstance(mc.dynamics, LognormalDynamics) and isinstance(mc.strategy, BlackScholesExact) \
            and isinstance(m, BlackScholesInputs):
        model.sigma = float(get_vol(m.sigma, None, None))
        c.dynamics, c.strategy = _ffi.HH_LOGNORMAL, _ffi.HH_EXACT_LAW
    elif isinstance(mc.dynamics, HestonDynamics) and isinstance(mc.strategy, HestonBroadieKaya) \
            and isinstance(m, HestonInputs):
        model.V0, model.kappa, model.theta = float(m.V0), float(m.κ), float(m.θ)
        model.sigma, model.rho = float(m.σ), float(m.ρ)
        c.dynamics, c.strategy = _ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA
    else:
        raise MethodError("full-path simulation on the HIP path: LognormalDynamics + BlackScholesExact "
                          "on BlackScholesInputs, or HestonDynamics + HestonBroadieKaya on HestonInputs")
    model.S0 = float(m.spot)
    model.r_drift = float(zero_rate(m.rate, 0.0))                      # montecarlo.jl:150,222
    model.T, model.strike, model.cp = float(T), float(payoff.strike), payoff.call_put()
    model.discount = 1.0
    c.antithetic = int(isinstance(cfg.variance_reduction, Antithetic))
    c.n_steps, c.n_paths = cfg.steps, cfg.trajectories
    c.seeds = cfg.seeds.ctypes.data
    c.seeds_len = cfg.seeds.size
    return model, c, T


def solve_lsm(prob: PricingProblem, method: LSM, spot_paths: bool = False,
              stopping_info: bool = True) -> LSMSolution:
    payoff, m = prob.payoff, prob.market_inputs
    if not (isinstance(payoff, VanillaOption) and isinstance(payoff.exercise_style, American)):
        raise MethodError("solve(::PricingProblem, ::LSM) needs an American VanillaOption")
    mc = method.mc_method
    model, c, T = _lsm_structs(prob, mc)
    cfg = mc.config
    nsteps = cfg.steps
    # discount = df(rate, add_yearfrac(referenceDate, T / nsteps))      # :107
    step_discount = float(df(m.rate, m.referenceDate + (T / nsteps) * MILLISECONDS_IN_YEAR_365))
    ntot = cfg.trajectories * (2 if c.antithetic else 1)
    tau = np.empty(ntot, dtype=np.int32) if stopping_info else None
    val = np.empty(ntot) if stopping_info else None
    grid = np.empty((nsteps + 1, ntot)) if spot_paths else None
    res = _ffi.hh_lsm_result()
    if mc.devices is not None:  # ONE call, the trajectories sharded over these GPUs inside the library
        if spot_paths:
            raise ValueError("spot_paths is not returned by the multi-GPU form")
        mg = _ffi.get_multi_gpu(tuple(mc.devices))
        mg.check(mg.lib.hh_mgpu_lsm_solve(mg.handle, C.byref(model), C.byref(c), method.degree, step_discount,
                                          C.byref(res), tau.ctypes.data if stopping_info else None,
                                          val.ctypes.data if stopping_info else None))
        return LSMSolution(prob, method, res.price, (tau, val) if stopping_info else None, None,
                           std_error=res.std_error, result=res)
    ctx = _ffi.get_context(mc.device)
    ctx.check(ctx.lib.hh_lsm_solve(ctx.handle, C.byref(model), C.byref(c), method.degree,
                                   step_discount, C.byref(res),
                                   tau.ctypes.data if stopping_info else None,
                                   val.ctypes.data if stopping_info else None,
                                   grid.ctypes.data if spot_paths else None))
    return LSMSolution(prob, method, res.price, (tau, val) if stopping_info else None, grid,
                       std_error=res.std_error, result=res)


@dataclass(frozen=True)
class HestonExactPaths:
    """What simulate_paths(sde_problem(prob, HestonDynamics(), HestonBroadieKaya()), method,
    NoVarianceReduction()) holds per trajectory (montecarlo.jl:209-231, 342-353): the state at the
    steps+1 dates, here as two (steps+1, trajectories) matrices.  `log_spot` is the first state
    component of the reference's solution objects."""
    spot: Any
    variance: Any
    times: Any
    result: Any = field(default=None, compare=False, repr=False)

    @property
    def log_spot(self):
        return np.log(self.spot)


def simulate_heston_exact_paths(prob: PricingProblem, mc: MonteCarlo) -> HestonExactPaths:
    """Per-date Broadie–Kaya transitions (HestonNoise, heston.jl:82-91) through hh_heston_exact_grid."""
    if not (isinstance(mc.dynamics, HestonDynamics) and isinstance(mc.strategy, HestonBroadieKaya)):
        raise MethodError("simulate_heston_exact_paths needs HestonDynamics + HestonBroadieKaya")
    model, c, T = _lsm_structs(prob, mc)
    n, steps = mc.config.trajectories, mc.config.steps
    spot = np.empty((steps + 1, n))
    var = np.empty((steps + 1, n))
    res = _ffi.hh_result()
    ctx = _ffi.get_context(mc.device)
    ctx.check(ctx.lib.hh_heston_exact_grid(ctx.handle, C.byref(model), C.byref(c), spot.ctypes.data,
                                           var.ctypes.data, 0, C.byref(res)))
    return HestonExactPaths(spot, var, np.linspace(0.0, T, steps + 1), res)
