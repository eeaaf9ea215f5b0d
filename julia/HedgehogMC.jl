# HedgehogMC.jl — thin Julia host layer over libhedgehog_mc.so (include/hedgehog_mc.h).
#
# WRITTEN WITHOUT A JULIA TOOLCHAIN: neither the build image nor the GPU box has `julia`, so this
# file has never been executed.  It is the reference-side binding a Hedgehog.jl maintainer would
# add (INTEGRATION.md); the same C-ABI is exercised for real by the Python mirror and the tests.
#
# Usage (on a host with Julia, Hedgehog.jl, ForwardDiff and a built libhedgehog_mc.so):
#     include("julia/HedgehogMC.jl"); using .HedgehogMC
#     HedgehogMC.install!()          # route Hedgehog.solve(::PricingProblem, ::MonteCarlo) to the GPU
#     sol = Hedgehog.solve(prob, MonteCarlo(HestonDynamics(), EulerMaruyama(), cfg))
module HedgehogMC

using Hedgehog
using ForwardDiff
import Hedgehog: PricingProblem, VanillaOption, European, Spot, MonteCarlo, MonteCarloSolution,
                 LognormalDynamics, HestonDynamics, EulerMaruyama, BlackScholesExact,
                 HestonBroadieKaya, Antithetic, BlackScholesInputs, HestonInputs,
                 yearfrac, zero_rate, df, get_vol

const LIB = Ref{String}(get(ENV, "HEDGEHOG_MC_LIB",
                            joinpath(@__DIR__, "..", "hedgehog.jl_amd", "lib", "libhedgehog_mc.so")))

const HH_MAX_PARTIALS = 8

# ---- C structs (layout of include/hedgehog_mc.h) ---------------------------------------------
struct HHModel
    S0::Cdouble; V0::Cdouble; kappa::Cdouble; theta::Cdouble; sigma::Cdouble; rho::Cdouble
    r_drift::Cdouble; discount::Cdouble; T::Cdouble; strike::Cdouble; cp::Cdouble
    dS0::Ptr{Cdouble}; dV0::Ptr{Cdouble}; dkappa::Ptr{Cdouble}; dtheta::Ptr{Cdouble}
    dsigma::Ptr{Cdouble}; dr_drift::Ptr{Cdouble}; ddiscount::Ptr{Cdouble}; dstrike::Ptr{Cdouble}
end

struct HHConfig
    dynamics::Int32; strategy::Int32; antithetic::Int32; em_split::Int32
    compat_sqrt_alpha::Int32; noise_mode::Int32; replay_layout::Int32
    seeds_on_device::Int32; replay_on_device::Int32; terminal_on_device::Int32
    n_steps::UInt32; n_partials::UInt32
    n_paths::UInt64; path_offset::UInt64
    seeds::Ptr{UInt64}; replay::Ptr{Cdouble}
    bk_n_sigma::Cdouble; bk_cf_tol::Cdouble; bk_atol::Cdouble; bk_moment_h::Cdouble
    bk_newton_maxiter::Int32; bk_bisect_maxiter::Int32
    seeds_len::UInt64; replay_len::UInt64
end

struct HHResult
    price::Cdouble; std_error::Cdouble; sum_payoff::Cdouble; sumsq_payoff::Cdouble
    dprice::NTuple{8,Cdouble}
    n_paths_done::UInt64
    bk_newton_fail::UInt64; bk_bisect_fallback::UInt64; bk_maxguess_fallback::UInt64
    bk_cf_terms::UInt64
    kernel_ms::Cdouble; total_ms::Cdouble
end

# ---- context ----------------------------------------------------------------------------------
mutable struct Context
    handle::Ptr{Cvoid}
end

function Context(device::Integer = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:hh_ctx_create, LIB[]), Cint, (Ref{Ptr{Cvoid}}, Cint), h, device)
    rc == 0 || error("hh_ctx_create failed ($rc): no HIP device? There is no CPU fallback.")
    ctx = Context(h[])
    finalizer(c -> ccall((:hh_ctx_destroy, LIB[]), Cvoid, (Ptr{Cvoid},), c.handle), ctx)
    return ctx
end

const CTX = Ref{Union{Nothing,Context}}(nothing)
context() = (CTX[] === nothing && (CTX[] = Context(0)); CTX[]::Context)

last_error(ctx) = unsafe_string(ccall((:hh_last_error, LIB[]), Cstring, (Ptr{Cvoid},), ctx.handle))

# ---- dual-number plumbing (greeks_problem.jl:258-260) ------------------------------------------
_val(x) = ForwardDiff.value(x)
_npartials(x) = x isa ForwardDiff.Dual ? ForwardDiff.npartials(x) : 0
_partials(x, P) = x isa ForwardDiff.Dual ? collect(Float64, ForwardDiff.partials(x)) : zeros(P)
_dualtype(xs...) = (i = findfirst(x -> x isa ForwardDiff.Dual, xs); i === nothing ? nothing : typeof(xs[i]))

# ---- packing shared by the single-payoff and the basket solve --------------------------------------
"""
Resolve dates / curves exactly as the reference does (montecarlo.jl:173-201, 299-318, 489) and return
everything `hh_model` / `hh_config` need: plain values, the dual seed vectors (kept alive by the
caller) and the Dual type to rebuild prices with.
"""
function _resolve(payoff, m, method::MonteCarlo)
    cfg = method.config
    dyn, strat = method.dynamics, method.strategy
    euler = strat isa EulerMaruyama
    if dyn isa LognormalDynamics && m isa BlackScholesInputs && (euler || strat isa BlackScholesExact)
        dynamics, strategy = 0, euler ? 0 : 1
        sigma, V0, kappa, theta, rho = get_vol(m.sigma, nothing, nothing), 0.0, 0.0, 0.0, 0.0
    elseif dyn isa HestonDynamics && m isa HestonInputs && (euler || strat isa HestonBroadieKaya)
        dynamics, strategy = 1, euler ? 0 : 2
        sigma, V0, kappa, theta, rho = m.σ, m.V0, m.κ, m.θ, m.ρ
    else
        return nothing                                           # no method: MethodError at the caller
    end
    if euler
        T = yearfrac(m.referenceDate, payoff.expiry)           # montecarlo.jl:173,197
        r_drift = zero_rate(m.rate, 0.0)                       # montecarlo.jl:176,200
    else
        T = yearfrac(m.rate.reference_date, payoff.expiry)     # montecarlo.jl:301,317
        r_drift = zero_rate(m.rate, payoff.expiry)             # montecarlo.jl:299,318
    end
    discount = df(m.rate, payoff.expiry)                        # montecarlo.jl:489
    scal = (m.spot, V0, kappa, theta, sigma, r_drift, discount, payoff.strike)
    P = maximum(_npartials, scal)
    P <= HH_MAX_PARTIALS || error("at most $HH_MAX_PARTIALS partials per solve")
    seedvecs = [_partials(x, P) for x in scal]
    seeds = convert(Vector{UInt64}, cfg.seeds .% UInt64)
    anti = cfg.variance_reduction isa Antithetic
    return (; dynamics, strategy, scal, rho = Float64(rho), T = Float64(T), P, seedvecs, seeds, anti,
            n = Int(cfg.trajectories), steps = Int(cfg.steps), cp = payoff.call_put(),
            DT = _dualtype(scal...))
end

# hh_model / hh_config from a resolved problem; the caller holds r.seedvecs and r.seeds in GC.@preserve
function _structs(r; em_split::Bool = true, compat_sqrt_alpha::Bool = false)
    ptr(i) = (r.P == 0 || all(iszero, r.seedvecs[i])) ? Ptr{Cdouble}(C_NULL) : pointer(r.seedvecs[i])
    v = map(_val, r.scal)
    model = HHModel(v[1], v[2], v[3], v[4], v[5], r.rho, v[6], v[7], r.T, v[8], r.cp,
                    ptr(1), ptr(2), ptr(3), ptr(4), ptr(5), ptr(6), ptr(7), ptr(8))
    config = HHConfig(r.dynamics, r.strategy, r.anti, em_split, compat_sqrt_alpha,
                      0, 0, 0, 0, 0, UInt32(r.steps), UInt32(r.P), UInt64(r.n), UInt64(0),
                      pointer(r.seeds), Ptr{Cdouble}(C_NULL), 0.0, 0.0, 0.0, 0.0, 0, 0,
                      UInt64(length(r.seeds)), UInt64(0))
    return model, config
end

_price(r, res::HHResult) = r.DT === nothing ? res.price :
    r.DT(res.price, ForwardDiff.Partials(ntuple(k -> res.dprice[k], r.P)))   # same tag as the input Dual

# ---- solve(prob, ::MonteCarlo) on the GPU (montecarlo.jl:478-493) -------------------------------
function solve_hip(prob::PricingProblem{VanillaOption{TS,TE,European,C,Spot},I},
                   method::MonteCarlo; ensemble::Bool = true,
                   em_split::Bool = true, compat_sqrt_alpha::Bool = false) where {TS,TE,C,I}
    r = _resolve(prob.payoff, prob.market_inputs, method)
    r === nothing && throw(MethodError(Hedgehog.solve, (prob, method)))
    terminal = ensemble ? Vector{Float64}(undef, r.anti ? 2r.n : r.n) : Float64[]
    res = Ref{HHResult}()
    ctx = context()
    seedvecs, seeds = r.seedvecs, r.seeds
    GC.@preserve seedvecs seeds terminal begin
        model, config = _structs(r; em_split, compat_sqrt_alpha)
        rc = ccall((:hh_mc_solve, LIB[]), Cint,
                   (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Ref{HHResult}, Ptr{Cdouble}),
                   ctx.handle, model, config, res,
                   ensemble ? pointer(terminal) : Ptr{Cdouble}(C_NULL))
        rc == -2 && throw(MethodError(Hedgehog.solve, (prob, method)))
        rc == 0 || error("hh_mc_solve failed ($rc): $(last_error(ctx))")
    end
    ens = !ensemble ? nothing : r.anti ? (terminal[1:r.n], terminal[r.n+1:2r.n]) : terminal
    return MonteCarloSolution(prob, method, _price(r, res[]), ens)       # pricing_solutions.jl:22-27
end

# ---- same-expiry baskets (src/calibration/basket.jl:35-38) ---------------------------------------
"""
    solve_basket_hip(prob::BasketPricingProblem, method::MonteCarlo)

One simulation per expiry group, every (strike, call/put) of the group reduced on the same terminal
samples (`hh_mc_solve_basket`).  Equal to the reference's independent per-payoff solves because the
seeds in `method.config` are fixed.  Strike partials are not carried through a basket.
"""
function solve_basket_hip(prob::Hedgehog.BasketPricingProblem, method::MonteCarlo)
    sols = Vector{Any}(undef, length(prob.payoffs))
    groups = Dict{Any,Vector{Int}}()
    for (i, p) in enumerate(prob.payoffs)
        push!(get!(groups, p.expiry, Int[]), i)
    end
    ctx = context()
    for idx in values(groups)
        first_payoff = prob.payoffs[idx[1]]
        r = _resolve(first_payoff, prob.market_inputs, method)
        r === nothing && throw(MethodError(Hedgehog.solve, (prob, method)))
        strikes = Float64[_val(prob.payoffs[i].strike) for i in idx]
        cps = Float64[prob.payoffs[i].call_put() for i in idx]
        res = Vector{HHResult}(undef, length(idx))
        seedvecs, seeds = r.seedvecs, r.seeds
        GC.@preserve seedvecs seeds strikes cps res begin
            model, config = _structs(r)
            rc = ccall((:hh_mc_solve_basket, LIB[]), Cint,
                       (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Ptr{Cdouble}, Ptr{Cdouble}, UInt32,
                        Ptr{HHResult}, Ptr{Cdouble}),
                       ctx.handle, model, config, pointer(strikes), pointer(cps), UInt32(length(idx)),
                       pointer(res), Ptr{Cdouble}(C_NULL))
            rc == 0 || error("hh_mc_solve_basket failed ($rc): $(last_error(ctx))")
        end
        for (k, i) in enumerate(idx)
            sols[i] = MonteCarloSolution(PricingProblem(prob.payoffs[i], prob.market_inputs), method,
                                         _price(r, res[k]), nothing)
        end
    end
    return Hedgehog.BasketPricingSolution(prob, sols)
end

# ---- LSM (src/pricing_methods/least_squares_montecarlo.jl:99-136) --------------------------------
struct HHLsmResult
    price::Cdouble; std_error::Cdouble
    n_paths_total::UInt64
    rows_regressed::UInt32; rows_skipped::UInt32
    kernel_ms::Cdouble; total_ms::Cdouble
end

"""
    solve_lsm_hip(prob, method::LSM)

`hh_lsm_solve`: GBM-process paths of (LognormalDynamics, BlackScholesExact), backward induction with
polynomial regression of degree `method.degree`.  Returns an `LSMSolution` whose `stopping_info` is
rebuilt from the (time, value) arrays and whose `spot_paths` is the (nsteps+1) x npaths matrix.
"""
function solve_lsm_hip(prob::PricingProblem{VanillaOption{TS,TE,Hedgehog.American,C,S},I},
                       method::Hedgehog.LSM) where {TS,TE,C,S,I<:BlackScholesInputs}
    mc, m, payoff = method.mc_method, prob.market_inputs, prob.payoff
    (mc.dynamics isa LognormalDynamics && mc.strategy isa BlackScholesExact) ||
        throw(MethodError(Hedgehog.solve, (prob, method)))
    cfg = mc.config
    T = yearfrac(m.referenceDate, payoff.expiry)
    nsteps = Int(cfg.steps)
    step_discount = df(m.rate, Hedgehog.add_yearfrac(m.referenceDate, T / nsteps))   # :107
    anti = cfg.variance_reduction isa Antithetic
    n = Int(cfg.trajectories); ntot = anti ? 2n : n
    seeds = convert(Vector{UInt64}, cfg.seeds .% UInt64)
    tau = Vector{Int32}(undef, ntot); val = Vector{Float64}(undef, ntot)
    grid = Matrix{Float64}(undef, ntot, nsteps + 1)          # column-major: [path, step] = C [step][path]
    res = Ref{HHLsmResult}()
    ctx = context()
    GC.@preserve seeds tau val grid begin
        model = HHModel(Float64(m.spot), 0.0, 0.0, 0.0, Float64(get_vol(m.sigma, nothing, nothing)), 0.0,
                        Float64(zero_rate(m.rate, 0.0)), 1.0, Float64(T), Float64(payoff.strike),
                        payoff.call_put(), ntuple(_ -> Ptr{Cdouble}(C_NULL), 8)...)
        config = HHConfig(0, 1, anti, 1, 0, 0, 0, 0, 0, 0, UInt32(nsteps), UInt32(0), UInt64(n),
                          UInt64(0), pointer(seeds), Ptr{Cdouble}(C_NULL), 0.0, 0.0, 0.0, 0.0, 0, 0,
                          UInt64(length(seeds)), UInt64(0))
        rc = ccall((:hh_lsm_solve, LIB[]), Cint,
                   (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Int32, Cdouble, Ref{HHLsmResult},
                    Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}),
                   ctx.handle, model, config, Int32(method.degree), Float64(step_discount), res,
                   pointer(tau), pointer(val), pointer(grid))
        rc == 0 || error("hh_lsm_solve failed ($rc): $(last_error(ctx))")
    end
    stopping_info = [(Int(tau[p]), val[p]) for p in 1:ntot]
    return Hedgehog.LSMSolution(prob, method, res[].price, stopping_info, permutedims(grid))
end

"""
    heston_exact_paths_hip(prob, method::MonteCarlo) -> (spot, variance)

`hh_heston_exact_grid`: what `simulate_paths(sde_problem(prob, HestonDynamics(), HestonBroadieKaya()),
method, NoVarianceReduction())` (montecarlo.jl:209-231, 342-353 on `HestonNoise`, heston.jl:82-91)
holds per trajectory, as two (nsteps+1) x npaths matrices; `log.(spot)` is the first state component
of the reference's solution objects.  One seed per trajectory (montecarlo.jl:331).
"""
function heston_exact_paths_hip(prob::PricingProblem{P,I}, method::MonteCarlo) where {P,I<:HestonInputs}
    (method.dynamics isa HestonDynamics && method.strategy isa HestonBroadieKaya) ||
        throw(MethodError(heston_exact_paths_hip, (prob, method)))
    m, payoff, cfg = prob.market_inputs, prob.payoff, method.config
    T = yearfrac(m.referenceDate, payoff.expiry)                                   # montecarlo.jl:219
    nsteps = Int(cfg.steps); n = Int(cfg.trajectories)
    seeds = convert(Vector{UInt64}, cfg.seeds .% UInt64)
    spot = Matrix{Float64}(undef, n, nsteps + 1)             # column-major: [path, step] = C [step][path]
    var = Matrix{Float64}(undef, n, nsteps + 1)
    ctx = context()
    GC.@preserve seeds spot var begin
        model = HHModel(Float64(m.spot), Float64(m.V0), Float64(m.κ), Float64(m.θ), Float64(m.σ),
                        Float64(m.ρ), Float64(zero_rate(m.rate, 0.0)), 1.0, Float64(T),
                        Float64(payoff.strike), 1.0, ntuple(_ -> Ptr{Cdouble}(C_NULL), 8)...)
        config = HHConfig(1, 2, false, 1, 0, 0, 0, 0, 0, 0, UInt32(nsteps), UInt32(0), UInt64(n),
                          UInt64(0), pointer(seeds), Ptr{Cdouble}(C_NULL), 0.0, 0.0, 0.0, 0.0, 0, 0,
                          UInt64(length(seeds)), UInt64(0))
        rc = ccall((:hh_heston_exact_grid, LIB[]), Cint,
                   (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Ptr{Cdouble}, Ptr{Cdouble}, Int32, Ptr{Cvoid}),
                   ctx.handle, model, config, pointer(spot), pointer(var), Int32(0), C_NULL)
        rc == 0 || error("hh_heston_exact_grid failed ($rc): $(last_error(ctx))")
    end
    return permutedims(spot), permutedims(var)
end

"""
    install!()

Overwrite `Hedgehog.solve(::PricingProblem{<:VanillaOption{…,European,…,Spot}}, ::MonteCarlo)`
(montecarlo.jl:478-481) with the GPU implementation.  `GreekProblem`/`BatchGreekProblem`/
`FiniteDifference` solvers (greeks_problem.jl:249-329, 559-568) then run through it unchanged.
"""
function install!()
    @eval Hedgehog function solve(
        prob::PricingProblem{VanillaOption{TS,TE,European,C,Spot},I}, method::MonteCarlo,
    ) where {TS,TE,C,I<:AbstractMarketInputs}
        return $(solve_hip)(prob, method)
    end
    return nothing
end

end # module
