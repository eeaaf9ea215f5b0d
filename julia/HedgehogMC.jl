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
#     G   = Hedgehog.solve(BatchGreekProblem(prob, lenses), ForwardAD(), mc)    # ONE fused pass
#
# What is bound (include/hedgehog_mc.h, HH_ABI_VERSION 6): hh_mc_solve (solve_hip, with the REPLAY
# keywords), hh_mc_solve_multi / hh_mgpu_solve_multi (prices_hip: the solves of a bumped Greek on shared draws),
# hh_seeds_cache (the library's own seed cache), hh_mgpu_create / hh_mgpu_solve (solve_hip(...; devices = 0:7): several GPUs behind the one
# call), hh_mc_accumulate + hh_mc_finalize (solve_sharded_hip: one process per GPU), hh_mc_solve_basket, hh_carr_madan,
# hh_carr_madan_basket (+ _grad), hh_ctx_set_option, hh_lsm_solve, hh_heston_exact_grid, hh_replay_elems, the device-memory helpers.  The struct mirrors
# below are checked field by field against the C header by tests/test_julia_layout.py (offsets from
# a compiled offsetof dump), so a drift between the two shows up on the CPU, without Julia.
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
const HH_ACC_LEN = 16
const HH_ABI_VERSION = 6
const HH_NOISE_GENERATE, HH_NOISE_REPLAY = Int32(0), Int32(1)
const HH_REPLAY_TILE_MAJOR, HH_REPLAY_PATH_MAJOR = Int32(0), Int32(1)

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
    bk_root_form::Int32; bk_bracket_form::Int32; bk_caps::Int32; reserved0::Int32   # readings of find_zero: all 0 = shipped
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

# build options of a context (include/hedgehog_mc.h, enum hh_option; none changes a result)
const HH_OPT_LSM_FORM = Int32(1)        # 0 launch per date, 1 one persistent launch, 2 auto (default)
const HH_OPT_BK_TERM_CACHE = Int32(2)   # cached CDF-series terms per trajectory, 8 … 1024 (default 256)
const HH_OPT_GRID_FORM = Int32(3)       # exact Heston grid: 0 one kernel chain per date, 1 dates batched (default)
function set_option!(ctx::Context, option::Integer, value::Integer)
    rc = ccall((:hh_ctx_set_option, LIB[]), Cint, (Ptr{Cvoid}, Int32, Int64), ctx.handle, option, value)
    rc == 0 || error("hh_ctx_set_option failed ($rc): " * last_error(ctx))
    return nothing
end

# ---- several GPUs behind ONE call (hh_mgpu_*: include/hedgehog_mc.h) -------------------------------
# One hh_mgpu = one hh_ctx per listed device, driven from the calling Julia thread; the library shards
# the trajectories, launches every device's kernels back to back and combines the HH_ACC_LEN-double
# accumulator vectors by one ncclAllReduce (RCCL bound at run time) — or by an ordered sum on the host
# when RCCL is missing or refuses.  No MPI, no extra Julia processes: `solve(prob, method)` stays ONE call.
const HH_MGPU_AUTO, HH_MGPU_HOST_SUM, HH_MGPU_RCCL = Cint(0), Cint(1), Cint(2)
mutable struct MultiGpu
    handle::Ptr{Cvoid}
    devices::Vector{Cint}
end

function MultiGpu(devices; flags::Integer = HH_MGPU_AUTO)
    devs = collect(Cint, devices)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:hh_mgpu_create, LIB[]), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cint}, Cint, Cint),
               h, devs, length(devs), flags)
    rc == 0 || error("hh_mgpu_create($(devs)) failed ($rc): no HIP device, a bad ordinal, or RCCL required but unavailable")
    mg = MultiGpu(h[], devs)
    finalizer(m -> ccall((:hh_mgpu_destroy, LIB[]), Cvoid, (Ptr{Cvoid},), m.handle), mg)
    return mg
end

const MGPUS = Dict{Vector{Cint},MultiGpu}()
multi_gpu(devices) = get!(() -> MultiGpu(devices), MGPUS, collect(Cint, devices))
last_error(mg::MultiGpu) = unsafe_string(ccall((:hh_mgpu_last_error, LIB[]), Cstring, (Ptr{Cvoid},), mg.handle))
"`:rccl` or `:host` — how this context combines the shards' accumulator vectors"
reduce_mode(mg::MultiGpu) = ccall((:hh_mgpu_reduce_mode, LIB[]), Cint, (Ptr{Cvoid},), mg.handle) == 1 ? :rccl : :host

# ---- a config's seeds on the device, once ---------------------------------------------------------
# Repeated solves on one SimulationConfig (finite-difference Greeks, calibration loops) would move its seed
# vector to the GPU again at every call.  The LIBRARY keeps device copies (hh_seeds_cache: found by length + a
# fingerprint of every element, so a vector mutated in place is simply another key; at most 8 per context,
# freed with it) — nothing is cached, hashed or freed on this side.  The pointer is valid until the next call
# that misses: ask right before each solve.
function _device_seeds(ctx::Context, host::Vector{UInt64})
    p = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve host begin
        rc = ccall((:hh_seeds_cache, LIB[]), Cint, (Ptr{Cvoid}, Ptr{UInt64}, UInt64, UInt64, Ref{Ptr{Cvoid}}),
                   ctx.handle, pointer(host), UInt64(length(host)), UInt64(0), p)
    end
    rc == 0 || error("hh_seeds_cache failed ($rc): $(last_error(ctx))")
    return p[]
end

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
    seeds = cfg.seeds isa Vector{UInt64} ? cfg.seeds : convert(Vector{UInt64}, cfg.seeds .% UInt64)
    anti = cfg.variance_reduction isa Antithetic
    return (; dynamics, strategy, scal, rho = Float64(rho), T = Float64(T), P, seedvecs, seeds, anti,
            n = Int(cfg.trajectories), steps = Int(cfg.steps), cp = payoff.call_put(),
            DT = _dualtype(scal...))
end

# hh_model / hh_config from a resolved problem; the caller holds r.seedvecs, r.seeds (and `replay`)
# in GC.@preserve.  n_paths / path_offset / seed_offset describe a shard (solve_sharded_hip).
# replay: Wiener increments (Euler: path-major [path][step][comp] as diff(sol.W.W) gives them, or the
# tile-major layout of hh_replay_elems), standard normals (exact law) or [V_T | u | Z] (Broadie–Kaya).
function _structs(r; em_split::Bool = true, compat_sqrt_alpha::Bool = false,
                  replay::Union{Nothing,Vector{Float64}} = nothing,
                  replay_layout::Int32 = HH_REPLAY_PATH_MAJOR,
                  n_paths::Int = r.n, path_offset::Int = 0, seed_offset::Int = 0,
                  seeds_dev::Ptr{Cvoid} = C_NULL)
    ptr(i) = (r.P == 0 || all(iszero, r.seedvecs[i])) ? Ptr{Cdouble}(C_NULL) : pointer(r.seedvecs[i])
    v = map(_val, r.scal)
    model = HHModel(v[1], v[2], v[3], v[4], v[5], r.rho, v[6], v[7], r.T, v[8], r.cp,
                    ptr(1), ptr(2), ptr(3), ptr(4), ptr(5), ptr(6), ptr(7), ptr(8))
    noise = replay === nothing ? HH_NOISE_GENERATE : HH_NOISE_REPLAY
    config = HHConfig(Int32(r.dynamics), Int32(r.strategy), Int32(r.anti), Int32(em_split),
                      Int32(compat_sqrt_alpha), noise, replay_layout, Int32(seeds_dev != C_NULL), Int32(0), Int32(0),
                      UInt32(r.steps), UInt32(r.P), UInt64(n_paths),
                      UInt64(path_offset),
                      (seeds_dev != C_NULL ? Ptr{UInt64}(seeds_dev) : pointer(r.seeds)) + 8 * seed_offset,
                      replay === nothing ? Ptr{Cdouble}(C_NULL) : pointer(replay),
                      0.0, 0.0, 0.0, 0.0, Int32(0), Int32(0), Int32(0), Int32(0), Int32(0), Int32(0),
                      UInt64(length(r.seeds) - seed_offset),
                      UInt64(replay === nothing ? 0 : length(replay)))
    return model, config
end

_price(r, res::HHResult) = r.DT === nothing ? res.price :
    r.DT(res.price, ForwardDiff.Partials(ntuple(k -> res.dprice[k], r.P)))   # same tag as the input Dual

# ---- solve(prob, ::MonteCarlo) on the GPU (montecarlo.jl:478-493) -------------------------------
# `devices = 0:7` shards the SAME solve over those GPUs inside the library (hh_mgpu_solve: contiguous
# trajectory ranges, one RCCL all-reduce of the accumulator vector or the host's ordered sum) — still
# one call from one Julia thread, the result differing from the one-GPU one by the order of the final
# sum only (<= 1e-13 relative); `devices = nothing` (default) is the one-GPU hh_mc_solve.
function solve_hip(prob::PricingProblem{VanillaOption{TS,TE,European,C,Spot},I},
                   method::MonteCarlo; ensemble::Bool = true,
                   em_split::Bool = true, compat_sqrt_alpha::Bool = false,
                   replay::Union{Nothing,AbstractArray{Float64}} = nothing,
                   replay_layout::Int32 = HH_REPLAY_PATH_MAJOR,
                   devices = nothing) where {TS,TE,C,I}
    r = _resolve(prob.payoff, prob.market_inputs, method)
    r === nothing && throw(MethodError(Hedgehog.solve, (prob, method)))
    terminal = ensemble ? Vector{Float64}(undef, r.anti ? 2r.n : r.n) : Float64[]
    res = Ref{HHResult}()
    ctx = devices === nothing ? context() : multi_gpu(devices)
    seedvecs, seeds = r.seedvecs, r.seeds
    rep = replay === nothing ? nothing : collect(Float64, vec(replay))   # noise replay (montecarlo.jl:258,370)
    # one GPU, increments drawn in the kernel: the seeds are read from their device copy (made on first use)
    sdev = (devices === nothing && rep === nothing) ? _device_seeds(ctx, seeds) : Ptr{Cvoid}(C_NULL)
    GC.@preserve seedvecs seeds terminal rep begin
        model, config = _structs(r; em_split, compat_sqrt_alpha, replay = rep, replay_layout, seeds_dev = sdev)
        term = ensemble ? pointer(terminal) : Ptr{Cdouble}(C_NULL)
        rc = devices === nothing ?
            ccall((:hh_mc_solve, LIB[]), Cint,
                  (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Ref{HHResult}, Ptr{Cdouble}),
                  ctx.handle, model, config, res, term) :
            ccall((:hh_mgpu_solve, LIB[]), Cint,
                  (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Ref{HHResult}, Ptr{Cdouble}),
                  ctx.handle, model, config, res, term)
        rc == -2 && throw(MethodError(Hedgehog.solve, (prob, method)))
        rc == 0 || error("hh_mc_solve failed ($rc): $(last_error(ctx))")
    end
    ens = !ensemble ? nothing : r.anti ? (terminal[1:r.n], terminal[r.n+1:2r.n]) : terminal
    return MonteCarloSolution(prob, method, _price(r, res[]), ens)       # pricing_solutions.jl:22-27
end

# ---- several problems on the SAME draws in one pass (hh_mc_solve_multi) ------------------------------
"""
    prices_hip(probs, method::MonteCarlo) -> Vector of prices

The solves a bumped Greek is made of — `compute_fd_derivative` (greeks_problem.jl:279-303) runs 2, the
second-order stencils (:396-422) 3 or 4, all on the seeds of one SimulationConfig — simulated together:
`hh_mc_solve_multi` steps every model on each draw.  Price k is `solve(probs[k], method).price` bit for bit.
Plain numbers only (a Dual input falls back to one solve per problem).
"""
function prices_hip(probs::AbstractVector, method::MonteCarlo; devices = DEVICES[])
    rs = [_resolve(p.payoff, p.market_inputs, method) for p in probs]
    any(r -> r === nothing, rs) && throw(MethodError(Hedgehog.solve, (probs[1], method)))
    if length(probs) < 2 || length(probs) > 16 || any(r -> r.P != 0, rs)
        return [solve_hip(p, method; ensemble = false, devices).price for p in probs]
    end
    seeds = rs[1].seeds
    ctx = devices === nothing ? context() : multi_gpu(devices)
    sdev = devices === nothing ? _device_seeds(ctx, seeds) : Ptr{Cvoid}(C_NULL)
    res = Vector{HHResult}(undef, length(probs))
    GC.@preserve seeds res begin
        models = [_structs(r)[1] for r in rs]
        _, config = _structs(rs[1]; seeds_dev = sdev)
        rc = devices === nothing ?
            ccall((:hh_mc_solve_multi, LIB[]), Cint,
                  (Ptr{Cvoid}, Ptr{HHModel}, UInt32, Ref{HHConfig}, Ptr{HHResult}, Ptr{Ptr{Cdouble}}),
                  ctx.handle, models, UInt32(length(models)), config, res, C_NULL) :
            ccall((:hh_mgpu_solve_multi, LIB[]), Cint,
                  (Ptr{Cvoid}, Ptr{HHModel}, UInt32, Ref{HHConfig}, Ptr{HHResult}),
                  ctx.handle, models, UInt32(length(models)), config, res)
        rc == -2 && throw(MethodError(Hedgehog.solve, (probs[1], method)))
        rc == 0 || error("hh_mc_solve_multi failed ($rc): $(last_error(ctx))")
    end
    return [x.price for x in res]
end

# ---- BatchGreekProblem + ForwardAD in ONE pass (greeks_problem.jl:559-568) ------------------------
struct HipBatchTag end   # the ForwardDiff tag of the fused pass's dual numbers

"""
    solve_batch_greeks_hip(gprob::BatchGreekProblem, mc::MonteCarlo) -> Dict(lens => greek)

The reference loops over the lenses and runs one full dual-number simulation per lens
(greeks_problem.jl:567, each `ForwardDiff.derivative` at :258-260).  Here every lens seeds ONE
partial of the same `Dual{Tag,Float64,L}` — `set(prob, lens, Dual(lens(prob), e_k))` — and a single
`hh_mc_solve` with `n_partials = L` returns all the derivatives: the kernels carry only the
directions that reach the diffusion, the rest is finished in closed form (include/hedgehog_mc.h,
`hh_model`).  Result and key type are the reference's: `Dict(lens => greek)`.
"""
function solve_batch_greeks_hip(gprob::Hedgehog.BatchGreekProblem, mc::MonteCarlo)
    prob = gprob.pricing_problem
    lenses = collect(gprob.lenses)
    L = length(lenses)
    L <= HH_MAX_PARTIALS || error("at most $HH_MAX_PARTIALS lenses per fused pass")
    DT = ForwardDiff.Dual{HipBatchTag,Float64,L}
    p = prob
    for (k, lens) in enumerate(lenses)
        x = lens(p)                                 # already a Dual when two lenses read the same input
        e = ForwardDiff.Partials(ntuple(j -> j == k ? 1.0 : 0.0, L))
        xd = x isa ForwardDiff.Dual ? DT(ForwardDiff.value(x), ForwardDiff.partials(x) + e) :
                                      DT(Float64(x), e)
        p = Hedgehog.set(p, lens, xd)
    end
    price = solve_hip(p, mc; ensemble = false, devices = DEVICES[]).price   # a Dual carrying all L partials
    parts = ForwardDiff.partials(price)
    return Dict(lens => parts[k] for (k, lens) in enumerate(lenses))
end

# ---- path-sharded solve: hh_mc_accumulate -> (all-reduce) -> hh_mc_finalize ------------------------
"""
    solve_sharded_hip(prob, method; rank, world, allreduce! = identity, device = rank)

One rank's part of a multi-GPU solve (one Julia process per GPU): trajectories
`[rank·⌈N/G⌉, min(N, (rank+1)·⌈N/G⌉))` (SURVEY §8e), seeds sliced by the same range, exact laws
offset by `path_offset`; the HH_ACC_LEN-double accumulator vector is summed over the ranks by
`allreduce!(acc::Vector{Float64})` (e.g. `MPI.Allreduce!(acc, +, comm)` — the path's ONE exchange;
any SUM all-reduce works, the vector is 128 bytes) and finalised on the host.  Every rank returns the
same `MonteCarloSolution` (no ensemble).
"""
function solve_sharded_hip(prob::PricingProblem{VanillaOption{TS,TE,European,C,Spot},I},
                           method::MonteCarlo; rank::Integer, world::Integer,
                           allreduce! = identity, device::Integer = rank) where {TS,TE,C,I}
    r = _resolve(prob.payoff, prob.market_inputs, method)
    r === nothing && throw(MethodError(Hedgehog.solve, (prob, method)))
    per = cld(r.n, world)
    start = min(r.n, rank * per)
    stop = min(r.n, start + per)
    ctx = Context(device)
    acc = zeros(Float64, HH_ACC_LEN)
    dev_acc = Ref{Ptr{Cvoid}}(C_NULL)
    seedvecs, seeds = r.seedvecs, r.seeds
    euler = r.strategy == 0
    GC.@preserve seedvecs seeds acc begin
        model, config = _structs(r; n_paths = stop - start, path_offset = start,
                                 seed_offset = euler ? start : 0)
        if stop > start
            rc = ccall((:hh_device_malloc, LIB[]), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}),
                       ctx.handle, 8 * HH_ACC_LEN, dev_acc)
            rc == 0 || error("hh_device_malloc failed ($rc): $(last_error(ctx))")
            rc = ccall((:hh_mc_accumulate, LIB[]), Cint,
                       (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Ptr{Cvoid}, Ptr{Cdouble}),
                       ctx.handle, model, config, dev_acc[], Ptr{Cdouble}(C_NULL))
            rc == -2 && throw(MethodError(Hedgehog.solve, (prob, method)))
            rc == 0 || error("hh_mc_accumulate failed ($rc): $(last_error(ctx))")
            rc = ccall((:hh_memcpy_d2h, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                       ctx.handle, pointer(acc), dev_acc[], 8 * HH_ACC_LEN)
            rc == 0 || error("hh_memcpy_d2h failed ($rc): $(last_error(ctx))")
            # the sums were left on the device by an asynchronous call: did a record reduction inside the kernel give up?
            rc = ccall((:hh_ctx_check_last, LIB[]), Cint, (Ptr{Cvoid},), ctx.handle)
            rc == 0 || error("hh_mc_accumulate lost its sums ($rc): $(last_error(ctx))")
            ccall((:hh_device_free, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.handle, dev_acc[])
        end
        allreduce!(acc)                                        # SUM over the ranks, in place
        res = Ref{HHResult}()
        rc = ccall((:hh_mc_finalize, LIB[]), Cint,
                   (Ref{HHModel}, Ref{HHConfig}, Ptr{Cdouble}, Ref{HHResult}),
                   model, config, pointer(acc), res)
        rc == 0 || error("hh_mc_finalize failed ($rc)")
        return MonteCarloSolution(prob, method, _price(r, res[]), nothing)
    end
end

# ---- Carr–Madan on the device (src/pricing_methods/carr_madan.jl:47-71) ----------------------------
"""
    carr_madan_hip(prob, method::Hedgehog.CarrMadan; compat_sqrt_alpha = false)

`hh_carr_madan`: the damped Fourier price the reference's Monte Carlo tests compare against
(α = method.α, bound = method.bound), for Heston or lognormal dynamics.
"""
function carr_madan_hip(prob::PricingProblem{VanillaOption{TS,TE,European,C,Spot},I},
                        method::Hedgehog.CarrMadan; compat_sqrt_alpha::Bool = false) where {TS,TE,C,I}
    m, payoff = prob.market_inputs, prob.payoff
    T = yearfrac(m.rate.reference_date, payoff.expiry)
    r = zero_rate(m.rate, payoff.expiry)
    D = df(m.rate, payoff.expiry)
    scal = m isa HestonInputs ? (m.spot, m.V0, m.κ, m.θ, m.σ, m.ρ, r, D) : (m.spot, get_vol(m.sigma, nothing, nothing), r, D)
    if _dualtype(scal...) !== nothing   # greeks_problem.jl:249-262: the price must carry the partials
        basket = Hedgehog.BasketPricingProblem([payoff], m)
        return carr_madan_basket_hip(basket, method; compat_sqrt_alpha = compat_sqrt_alpha).solutions[1]
    end
    none = ntuple(_ -> Ptr{Cdouble}(C_NULL), 8)
    if m isa HestonInputs
        dynamics = Int32(1)
        model = HHModel(Float64(m.spot), Float64(m.V0), Float64(m.κ), Float64(m.θ), Float64(m.σ),
                        Float64(m.ρ), Float64(r), Float64(D), Float64(T), Float64(payoff.strike),
                        payoff.call_put(), none...)
    else
        dynamics = Int32(0)
        model = HHModel(Float64(m.spot), 0.0, 0.0, 0.0, Float64(get_vol(m.sigma, nothing, nothing)), 0.0,
                        Float64(r), Float64(D), Float64(T), Float64(payoff.strike), payoff.call_put(),
                        none...)
    end
    out = Ref{Cdouble}(0.0)
    ctx = context()
    rc = ccall((:hh_carr_madan, LIB[]), Cint,
               (Ptr{Cvoid}, Ref{HHModel}, Int32, Int32, Cdouble, Cdouble, Ref{Cdouble}),
               ctx.handle, model, dynamics, Int32(compat_sqrt_alpha), Float64(method.α),
               Float64(method.bound), out)
    rc == 0 || error("hh_carr_madan failed ($rc): $(last_error(ctx))")
    return Hedgehog.AnalyticSolution(prob, method, out[])
end

"""
    carr_madan_basket_hip(prob::BasketPricingProblem, method::CarrMadan)

`hh_carr_madan_basket`: every payoff's Fourier integral in ONE launch (a workgroup per payoff) —
`solve(::BasketPricingProblem, ::CarrMadan)` (src/calibration/basket.jl:35-38), what the calibration
objective evaluates at every iterate (src/calibration/calibration.jl:75-88).  When an input is a
`ForwardDiff.Dual` (the objective under `AutoForwardDiff`), `hh_carr_madan_basket_grad` returns the
gradient along (S0, V0, κ, θ, σ, ρ, r_drift_k, discount_k) with the prices and the Dual prices are
assembled here: `price + Σ_j grad[j] · partials(parameter_j)`.
"""
function carr_madan_basket_hip(prob::Hedgehog.BasketPricingProblem, method::Hedgehog.CarrMadan;
                               compat_sqrt_alpha::Bool = false)
    m = prob.market_inputs
    n = length(prob.payoffs)
    strikes = Float64[p.strike for p in prob.payoffs]
    cps = Float64[p.call_put() for p in prob.payoffs]
    Ts = Float64[yearfrac(m.rate.reference_date, p.expiry) for p in prob.payoffs]
    r_k = [zero_rate(m.rate, p.expiry) for p in prob.payoffs]
    D_k = [df(m.rate, p.expiry) for p in prob.payoffs]
    rs, Ds = Float64[_val(x) for x in r_k], Float64[_val(x) for x in D_k]
    none = ntuple(_ -> Ptr{Cdouble}(C_NULL), 8)
    if m isa HestonInputs
        dynamics = Int32(1)
        scal = (m.spot, m.V0, m.κ, m.θ, m.σ, m.ρ)
    else
        dynamics = Int32(0)
        scal = (m.spot, 0.0, 0.0, 0.0, get_vol(m.sigma, nothing, nothing), 0.0)
    end
    model = HHModel(Float64(_val(scal[1])), Float64(_val(scal[2])), Float64(_val(scal[3])),
                    Float64(_val(scal[4])), Float64(_val(scal[5])), Float64(_val(scal[6])),
                    0.0, 1.0, 1.0, 1.0, 1.0, none...)
    out = Vector{Float64}(undef, n)
    ctx = context()
    P = max(maximum(_npartials, scal), maximum(_npartials, r_k), maximum(_npartials, D_k))
    mk(p, price) = Hedgehog.AnalyticSolution(PricingProblem(p, m), method, price)
    if P == 0
        rc = ccall((:hh_carr_madan_basket, LIB[]), Cint,
                   (Ptr{Cvoid}, Ref{HHModel}, Int32, Int32, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble},
                    Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, UInt32, Ptr{Cdouble}),
                   ctx.handle, model, dynamics, Int32(compat_sqrt_alpha), Float64(method.α),
                   Float64(method.bound), strikes, cps, Ts, rs, Ds, UInt32(n), out)
        rc == 0 || error("hh_carr_madan_basket failed ($rc): $(last_error(ctx))")
        return Hedgehog.BasketPricingSolution(prob, [mk(p, out[k]) for (k, p) in enumerate(prob.payoffs)])
    end
    grad = Matrix{Float64}(undef, 8, n)     # column k = hh_cm_grad of payoff k (C: [n][8])
    rc = ccall((:hh_carr_madan_basket_grad, LIB[]), Cint,
               (Ptr{Cvoid}, Ref{HHModel}, Int32, Int32, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, UInt32, Ptr{Cdouble}, Ptr{Cdouble}),
               ctx.handle, model, dynamics, Int32(compat_sqrt_alpha), Float64(method.α),
               Float64(method.bound), strikes, cps, Ts, rs, Ds, UInt32(n), out, grad)
    rc == 0 || error("hh_carr_madan_basket_grad failed ($rc): $(last_error(ctx))")
    DT = _dualtype(scal..., r_k..., D_k...)
    seeds = [_partials(x, P) for x in scal]
    sols = map(enumerate(prob.payoffs)) do (k, p)
        sr, sD = _partials(r_k[k], P), _partials(D_k[k], P)
        d = ntuple(q -> sum(grad[j, k] * seeds[j][q] for j in 1:6) + grad[7, k] * sr[q] + grad[8, k] * sD[q], P)
        mk(p, DT(out[k], ForwardDiff.Partials(d)))
    end
    return Hedgehog.BasketPricingSolution(prob, sols)
end

# ---- same-expiry baskets (src/calibration/basket.jl:35-38) ---------------------------------------
"""
    solve_basket_hip(prob::BasketPricingProblem, method::MonteCarlo)

One simulation per expiry group, every (strike, call/put) of the group reduced on the same terminal
samples (`hh_mc_solve_basket`).  Equal to the reference's independent per-payoff solves because the
seeds in `method.config` are fixed.  Strike partials are not carried through a basket.
"""
function solve_basket_hip(prob::Hedgehog.BasketPricingProblem, method::MonteCarlo; devices = nothing)
    # only what solve_hip itself prices (European vanilla payoffs on the spot, no strike partials); anything
    # else in the basket: the reference's own loop (basket.jl:36), payoff by payoff
    fused = all(p -> p isa VanillaOption{<:Any,<:Any,European,<:Any,Spot} && !(p.strike isa ForwardDiff.Dual),
                prob.payoffs)
    if !fused
        return Hedgehog.BasketPricingSolution(
            prob, [Hedgehog.solve(PricingProblem(p, prob.market_inputs), method) for p in prob.payoffs])
    end
    sols = Vector{Any}(undef, length(prob.payoffs))
    groups = Dict{Any,Vector{Int}}()
    for (i, p) in enumerate(prob.payoffs)
        push!(get!(groups, p.expiry, Int[]), i)
    end
    ctx = devices === nothing ? context() : multi_gpu(devices)   # devices = 0:7: hh_mgpu_solve_basket, one call
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
            rc = devices === nothing ?
                ccall((:hh_mc_solve_basket, LIB[]), Cint,
                      (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Ptr{Cdouble}, Ptr{Cdouble}, UInt32,
                       Ptr{HHResult}, Ptr{Cdouble}),
                      ctx.handle, model, config, pointer(strikes), pointer(cps), UInt32(length(idx)),
                      pointer(res), Ptr{Cdouble}(C_NULL)) :
                ccall((:hh_mgpu_solve_basket, LIB[]), Cint,
                      (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Ptr{Cdouble}, Ptr{Cdouble}, UInt32, Ptr{HHResult}),
                      ctx.handle, model, config, pointer(strikes), pointer(cps), UInt32(length(idx)), pointer(res))
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
    form::Int32; persistent_fallbacks::Int32
end

# (a docstring must sit DIRECTLY above its function: nothing between the closing quotes and `function`)
"""
    solve_lsm_hip(prob, method::LSM; devices = nothing)

`hh_lsm_solve`: GBM-process paths of (LognormalDynamics, BlackScholesExact), backward induction with
polynomial regression of degree `method.degree`.  Returns an `LSMSolution` whose `stopping_info` is
rebuilt from the (time, value) arrays and whose `spot_paths` is the (nsteps+1) x npaths matrix.

`devices = 0:7`: the same solve with the trajectories sharded over those GPUs inside the library
(`hh_mgpu_lsm_solve`: the phased induction on every device, its per-date sums all-reduced in the library);
the spot grid is then not returned (`spot_paths` of the solution is an empty matrix).
"""
function solve_lsm_hip(prob::PricingProblem{VanillaOption{TS,TE,Hedgehog.American,C,S},I},
                       method::Hedgehog.LSM; devices = nothing) where {TS,TE,C,S,I<:BlackScholesInputs}
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
        config = HHConfig(Int32(0), Int32(1), Int32(anti), Int32(1), Int32(0), Int32(0), Int32(0), Int32(0),
                          Int32(0), Int32(0), UInt32(nsteps), UInt32(0), UInt64(n),
                          UInt64(0), pointer(seeds), Ptr{Cdouble}(C_NULL), 0.0, 0.0, 0.0, 0.0, Int32(0), Int32(0),
                          Int32(0), Int32(0), Int32(0), Int32(0), UInt64(length(seeds)), UInt64(0))
        if devices !== nothing
            mg = multi_gpu(devices)
            rc = ccall((:hh_mgpu_lsm_solve, LIB[]), Cint,
                       (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Int32, Cdouble, Ref{HHLsmResult},
                        Ptr{Int32}, Ptr{Cdouble}),
                       mg.handle, model, config, Int32(method.degree), Float64(step_discount), res,
                       pointer(tau), pointer(val))
            rc == 0 || error("hh_mgpu_lsm_solve failed ($rc): $(last_error(mg))")
            return Hedgehog.LSMSolution(prob, method, res[].price,
                                        [(Int(tau[p]), val[p]) for p in 1:ntot], Matrix{Float64}(undef, 0, 0))
        end
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
                          UInt64(0), pointer(seeds), Ptr{Cdouble}(C_NULL), 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0, 0,
                          UInt64(length(seeds)), UInt64(0))
        rc = ccall((:hh_heston_exact_grid, LIB[]), Cint,
                   (Ptr{Cvoid}, Ref{HHModel}, Ref{HHConfig}, Ptr{Cdouble}, Ptr{Cdouble}, Int32, Ptr{Cvoid}),
                   ctx.handle, model, config, pointer(spot), pointer(var), Int32(0), C_NULL)
        rc == 0 || error("hh_heston_exact_grid failed ($rc): $(last_error(ctx))")
    end
    return permutedims(spot), permutedims(var)
end

const DEVICES = Ref{Any}(nothing)   # install!(devices = 0:7): every routed solve is sharded over these GPUs

"""
    install!(; devices = nothing)

Overwrite `Hedgehog.solve(::PricingProblem{<:VanillaOption{…,European,…,Spot}}, ::MonteCarlo)`
(montecarlo.jl:478-481) with the GPU implementation and add the fused
`solve(::BatchGreekProblem, ::ForwardAD, ::MonteCarlo)`, LSM on BlackScholesExact paths, Monte Carlo and
Carr–Madan baskets.  `GreekProblem` + `ForwardAD` (greeks_problem.jl:249-262) then runs through the first
unchanged; `FiniteDifference` through MonteCarlo gets its 2-4 prices from ONE pass on shared draws
(`prices_hip`).  Idempotent: a second call only changes `devices`.  Must not run during precompilation (call it
from `__init__` or from the session).  `install!(devices = 0:7)`
shards every routed solve over those GPUs inside the library (hh_mgpu_solve) — `solve(prob, method)`
stays one call, as montecarlo.jl:478-493.
"""
const INSTALLED = Ref(false)

function install!(; devices = nothing)
    ccall((:hh_abi_version, LIB[]), Cint, ()) == HH_ABI_VERSION ||
        error("libhedgehog_mc.so has another ABI version than this file ($HH_ABI_VERSION)")
    DEVICES[] = devices                   # read at every routed call: a second install! only changes this
    INSTALLED[] && return nothing         # the methods are defined ONCE (no "method overwritten" on later calls)
    # NOT during precompilation: `@eval Hedgehog …` adds methods to another module, which a precompiling
    # package may not do (INTEGRATION.md) — call install!() from __init__ or from the session
    ccall(:jl_generating_output, Cint, ()) == 1 &&
        error("HedgehogMC.install!() must not run while a package is being precompiled: call it from __init__()")
    @eval Hedgehog function solve(
        prob::PricingProblem{VanillaOption{TS,TE,European,C,Spot},I}, method::MonteCarlo,
    ) where {TS,TE,C,I<:AbstractMarketInputs}
        return $(solve_hip)(prob, method; devices = $(DEVICES)[])
    end
    # the fused form of greeks_problem.jl:559-568 (more specific than the reference's generic method:
    # ForwardAD + MonteCarlo); analytic methods keep the reference's loop
    @eval Hedgehog function solve(
        gprob::BatchGreekProblem{P,L}, ::ForwardAD, pricing_method::MonteCarlo,
    ) where {P,L}
        return $(solve_batch_greeks_hip)(gprob, pricing_method)
    end
    # FiniteDifference through MonteCarlo: the reference's formulas (greeks_problem.jl:279-303, 396-422) on
    # prices that come out of ONE pass on shared draws instead of 2-4 solves — the same numbers
    @eval Hedgehog function compute_fd_derivative(::FDForward, prob, lens, ε, pricing_method::MonteCarlo)
        x₀ = lens(prob)
        v_up, v₀ = $(prices_hip)([set(prob, lens, x₀ * (1 + ε)), prob], pricing_method)
        return (v_up - v₀) / (x₀ * ε)
    end
    @eval Hedgehog function compute_fd_derivative(::FDBackward, prob, lens, ε, pricing_method::MonteCarlo)
        x₀ = lens(prob)
        v_down, v₀ = $(prices_hip)([set(prob, lens, x₀ * (1 - ε)), prob], pricing_method)
        return (v₀ - v_down) / (x₀ * ε)
    end
    @eval Hedgehog function compute_fd_derivative(::FDCentral, prob, lens, ε, pricing_method::MonteCarlo)
        x₀ = lens(prob)
        v_up, v_down = $(prices_hip)([set(prob, lens, x₀ * (1 + ε)), set(prob, lens, x₀ * (1 - ε))], pricing_method)
        return (v_up - v_down) / (2ε * x₀)
    end
    @eval Hedgehog function solve(gprob::SecondOrderGreekProblem, method::FiniteDifference, pricing_method::MonteCarlo)
        prob, lens1, lens2, ε = gprob.pricing_problem, gprob.wrt1, gprob.wrt2, method.bump
        x₀, y₀ = lens1(prob), lens2(prob)
        at(x, y) = set(set(prob, lens1, x), lens2, y)
        if lens1 === lens2
            f_plus, f_0, f_minus = $(prices_hip)([at(x₀ + ε, y₀ + ε), at(x₀, y₀), at(x₀ - ε, y₀ - ε)], pricing_method)
            return GreekResult((f_plus - 2f_0 + f_minus) / (ε^2))
        end
        f_pp, f_pm, f_mp, f_mm = $(prices_hip)([at(x₀ + ε, y₀ + ε), at(x₀ + ε, y₀ - ε), at(x₀ - ε, y₀ + ε),
                                                at(x₀ - ε, y₀ - ε)], pricing_method)
        return GreekResult((f_pp - f_pm - f_mp + f_mm) / (4ε^2))
    end
    # LSM on the path source the reference's own LSM tests use (test/agreement/american_options.jl):
    # LognormalDynamics + BlackScholesExact on BlackScholesInputs.  More specific than the reference's method
    # (least_squares_montecarlo.jl:99-102) in BOTH arguments, so every other LSM combination keeps running the
    # reference's code; same-expiry Monte Carlo baskets likewise (basket.jl:35-38 stays for other methods).
    @eval Hedgehog function solve(
        prob::PricingProblem{VanillaOption{TS,TE,American,C,S},I},
        method::LSM{MonteCarlo{LognormalDynamics,BlackScholesExact,CF}},
    ) where {TS,TE,C,S,I<:BlackScholesInputs,CF}
        return $(solve_lsm_hip)(prob, method; devices = $(DEVICES)[])
    end
    @eval Hedgehog function solve(prob::BasketPricingProblem, method::MonteCarlo)
        return $(solve_basket_hip)(prob, method; devices = $(DEVICES)[])
    end
    # the calibration objective's inner loop (calibration.jl:75-88): every quote in one launch, Dual
    # inputs (AutoForwardDiff) included
    @eval Hedgehog function solve(prob::BasketPricingProblem, method::CarrMadan)
        return $(carr_madan_basket_hip)(prob, method)
    end
    INSTALLED[] = true
    return nothing
end

end # module
