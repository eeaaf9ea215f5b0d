"""MI355X-native Monte Carlo engine behind Hedgehog.jl's `solve(problem, MonteCarlo(...))` API.

Host mirror (Python) of the reference's operator interface for this path; the arithmetic runs in
hand-written HIP kernels behind the C-ABI of include/hedgehog_mc.h (lib/libhedgehog_mc.so).
Import as `hedgehog_jl_amd` (shim at the repository root; the directory name carries a dot).
"""
from . import _ffi
from ._ffi import Context, HedgehogMCError, get_context, load_library
from .analytic import (AnalyticSolution, BlackScholesAnalytic, CarrMadan, solve_black_scholes,
                       solve_carr_madan)
from .basket import BasketPricingProblem, BasketPricingSolution, solve_basket
from .dates import Date, DateTime, add_years, to_ticks, yearfrac
from .dual import Dual, partials_of, value_of
from .greeks import (BatchGreekProblem, FDBackward, FDCentral, FDForward, FiniteDifference,
                     ForwardAD, GreekProblem, GreekResult, PropertyLens, SecondOrderGreekProblem, SpotLens,
                     VolLens,
                     ZeroRateSpineLens, optic, set)
from .lsm import LSM, HestonExactPaths, LSMSolution, simulate_heston_exact_paths, solve_lsm
from .montecarlo import (AbstractPricingMethod, Antithetic, BlackScholesExact, EulerMaruyama,
                         HestonBroadieKaya, HestonDynamics, LognormalDynamics, MethodError,
                         MonteCarlo, NoVarianceReduction, NormalLaw, SimulationConfig, marginal_law,
                         solve_montecarlo, solve_montecarlo_many)
from .distributed import rank_device, shard_range, solve_lsm_sharded, solve_sharded
from .domain import (American, BlackScholesInputs, Call, European, FlatRateCurve, FlatVolSurface,
                    Forward, HestonInputs, MonteCarloSolution, PricingProblem, Put, RateCurve, Spot,
                    VanillaOption, df, df_yf, get_vol, spine_zeros, zero_rate, zero_rate_yf)


def solve(*args, **kw):
    """The reference's single verb (src/Hedgehog.jl:59-98), for the methods on the hot path:

        solve(prob::PricingProblem, method::MonteCarlo)                      montecarlo.jl:478
        solve(gprob::GreekProblem, ::ForwardAD, method)                      greeks_problem.jl:249
        solve(gprob::GreekProblem, ::FiniteDifference, method)               greeks_problem.jl:318
        solve(gprob::SecondOrderGreekProblem, ::FiniteDifference, method)    greeks_problem.jl:396
        solve(gprob::BatchGreekProblem, ::GreekMethod, method)               greeks_problem.jl:559
        solve(prob::BasketPricingProblem, method::MonteCarlo | ::CarrMadan)  basket.jl:35
        solve(prob, ::CarrMadan) / solve(prob, ::BlackScholesAnalytic)       carr_madan.jl:47, black_scholes.jl:38
        solve(prob::PricingProblem{<:VanillaOption{…,American,…}}, ::LSM)     least_squares_montecarlo.jl:99
    """
    from . import greeks as _g
    if len(args) == 2 and isinstance(args[0], PricingProblem) and isinstance(args[1], MonteCarlo):
        return solve_montecarlo(args[0], args[1], **kw)
    if len(args) == 2 and isinstance(args[0], PricingProblem) and isinstance(args[1], CarrMadan):
        return solve_carr_madan(args[0], args[1])
    if len(args) == 2 and isinstance(args[0], PricingProblem) and \
            isinstance(args[1], BlackScholesAnalytic):
        return solve_black_scholes(args[0], args[1])
    if len(args) == 2 and isinstance(args[0], PricingProblem) and isinstance(args[1], LSM):
        return solve_lsm(args[0], args[1], **kw)
    if len(args) == 2 and isinstance(args[0], BasketPricingProblem) and isinstance(args[1], (MonteCarlo, CarrMadan)):
        return solve_basket(args[0], args[1], **kw)
    if len(args) == 3 and isinstance(args[0], GreekProblem):
        if isinstance(args[1], ForwardAD):
            return _g.solve_greek_ad(args[0], args[2], solve)
        if isinstance(args[1], FiniteDifference):
            return _g.solve_greek_fd(args[0], args[1], args[2], solve)
    if len(args) == 3 and isinstance(args[0], SecondOrderGreekProblem) and \
            isinstance(args[1], FiniteDifference):
        return _g.solve_second_order_fd(args[0], args[1], args[2], solve)
    if len(args) == 3 and isinstance(args[0], BatchGreekProblem):
        return _g.solve_batch(args[0], args[1], args[2], solve)
    raise MethodError("no method matching solve(" + ", ".join(type(a).__name__ for a in args) + ")")
