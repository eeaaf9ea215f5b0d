# parity_replay.jl — exports the REFERENCE's own draws and results, so that every kernel of the HIP
# path can be checked per trajectory against Hedgehog.jl itself (closes the "parity unpinned" gap of
# DESIGN.md §2 in ONE run).
#
# NOT RUN in this build: there is no Julia on the build image or on the GPU box.  On a host with
# Julia + Hedgehog.jl (+ its dependencies; nothing else):
#
#     julia --project julia/parity_replay.jl out_dir                  # writes out_dir/manifest.json + *.bin
#     julia --project julia/parity_replay.jl out_dir probe            # the two deciding cases only (em_split, bk_root_probe: seconds)
#     python tools/check_reference_replay.py out_dir/manifest.json    # on the MI355X box
#
# Mechanism: every case simulates with the reference's OWN functions and records what its RNGs
# produced, at the seam the C-ABI offers for it (HH_NOISE_REPLAY):
#   euler            Wiener increments diff(sol.W.W) of simulate_paths(...; save_noise) — exactly what the
#                    reference itself replays for antithetic variates (montecarlo.jl:258,370)
#   exact_lognormal  the standard normals behind rand(rng, Normal(μ̃, σ̃), n) (montecarlo.jl:413,456)
#   bk               V_T, u, Z of rand(rng, ::LogHestonDistribution) (heston.jl:246-259), re-drawn on a
#                    copy of the generator in the reference's order
#   bk_root_probe    every abscissa the reference's inverse_cdf (sample_from_cf.jl:105-135) asks of its CDF, for 64
#                    (V_T, u) pairs: the iterates of Roots.jl's two find_zero calls, which decide the reading
#   lsm              the spot grid extract_spot_grid hands the regression (least_squares_montecarlo.jl:47-85)
# and stores the reference's results next to them: terminal samples, price, AD Greeks, stopping info.
# File formats: little-endian Float64 / Int32, shapes in the manifest; tests/golden/replay_selftest/
# holds the same format written by the CPU oracle (tests/golden/make_replay_selftest.py), which is what
# the `-m gpu` tests exercise until a Julia host has produced the real thing.
using Hedgehog, StochasticDiffEq, Dates, Random, Distributions, ForwardDiff, Accessors
import Hedgehog: sde_problem, get_ensemble_problem, final_sample, reduce_payoffs, marginal_law

outdir = length(ARGS) >= 1 ? ARGS[1] : "replay_out"
mkpath(outdir)
cases = String[]

jsonval(x::AbstractString) = "\"" * x * "\""
jsonval(x::Bool) = x ? "true" : "false"
jsonval(x::Real) = repr(Float64(x))
jsonval(x::Integer) = string(x)
jsonval(x::AbstractVector) = "[" * join(jsonval.(x), ", ") * "]"
jsonval(x::AbstractDict) = "{" * join(["\"$k\": " * jsonval(v) for (k, v) in x], ", ") * "}"
function add_case!(name; kw...)
    push!(cases, "{" * join(["\"name\": " * jsonval(name); ["\"$k\": " * jsonval(v) for (k, v) in kw]], ", ") * "}")
end
wbin(name, a) = (write(joinpath(outdir, name), a); name)

ref = Date(2021, 1, 1); expiry = Date(2022, 1, 1)                     # T = 1 under ACT/365
heston = (S0 = 100.0, K = 100.0, r = 0.03, V0 = 0.04, κ = 2.0, θ = 0.04, σ = 0.3, ρ = -0.7)
hprob = PricingProblem(VanillaOption(heston.K, expiry, European(), Call(), Spot()),
                       HestonInputs(ref, heston.r, heston.S0, heston.V0, heston.κ, heston.θ, heston.σ, heston.ρ))
bs = (S0 = 100.0, K = 100.0, r = 0.05, σ = 0.2)
bprob = PricingProblem(VanillaOption(bs.K, expiry, European(), Call(), Spot()),
                       BlackScholesInputs(ref, bs.r, bs.S0, bs.σ))
model_json(p::typeof(heston), T) = Dict("S0" => p.S0, "strike" => p.K, "r" => p.r, "V0" => p.V0, "kappa" => p.κ,
                                        "theta" => p.θ, "sigma" => p.σ, "rho" => p.ρ, "T" => T, "cp" => 1.0)
model_json(p::typeof(bs), T) = Dict("S0" => p.S0, "strike" => p.K, "r" => p.r, "sigma" => p.σ, "T" => T, "cp" => 1.0)

# ---- Euler–Maruyama: increments of the saved noise, per trajectory ------------------------------------
function export_euler(name, prob, params, dynamics, N, M; antithetic = false, lenses = ())
    vr = antithetic ? Antithetic() : Hedgehog.NoVarianceReduction()
    cfg = SimulationConfig(N; steps = M, seeds = collect(UInt64, 1:N), variance_reduction = vr)
    method = MonteCarlo(dynamics, EulerMaruyama(), cfg)
    sde = sde_problem(prob, method)
    T = sde.tspan[2]
    ens = StochasticDiffEq.solve(get_ensemble_problem(sde, cfg), EM(); dt = T / M, trajectories = N,
                                 save_noise = true)
    nc = dynamics isa HestonDynamics ? 2 : 1
    dW = Array{Float64}(undef, nc, M, N)          # column-major [comp, step, path] == C [path][step][comp]
    for i in 1:N, s in 1:M
        dW[:, s, i] .= ens.u[i].W.W[s + 1] .- ens.u[i].W.W[s]
    end
    # what the reference computes from these very draws (same seeds => same noise)
    sol = solve(prob, method)
    ST = antithetic ? vcat(sol.ensemble[1], sol.ensemble[2]) : sol.ensemble
    greeks = Dict{String,Float64}()
    for (key, lens) in lenses                     # greeks_problem.jl:249-262 on the same seeds
        greeks[key] = solve(GreekProblem(prob, lens), ForwardAD(), method).greek
    end
    add_case!(name; kind = "euler", dynamics = dynamics isa HestonDynamics ? "heston" : "lognormal",
              n_paths = N, n_steps = M, antithetic = antithetic, model = model_json(params, T),
              price = sol.price, dW = wbin("$name.dW.bin", dW), ST = wbin("$name.ST.bin", collect(Float64, ST)),
              layout = "dW: path-major [path][step][comp]; ST: [n_paths] (+ [n_paths] mirrored)",
              greeks = greeks)
end

# FIRST, and alone when called as `parity_replay.jl out_dir probe` (seconds, 64 trajectories): the case that
# settles the largest unpinned choice of the build — whether StochasticDiffEq's EM() evaluates the diffusion at
# u or at K = u + dt·f(u) (`em_split`, SURVEY §8a-4).  8 steps of dt = 1/8 make the two forms differ by percents
# per trajectory; tools/check_reference_replay.py prints `VERDICT em_split = 0|1 matches the reference`.
export_euler("em_split_probe", hprob, heston, HestonDynamics(), 64, 8)

# SECOND, also in `probe` mode (seconds): ---- Broadie–Kaya: what inverse_cdf ASKS of its CDF (the iterates of Roots.jl's two find_zero calls) --------
# The one place where the restatement is KNOWN to be a reading: `find_zero(func, x0, Order2(); atol, maxeval)` and
# `find_zero(func, (0, max_guess); xtol, maxeval)` (sample_from_cf.jl:118,128) — Roots.jl is a dependency, and whether
# Order2 is a secant or a Steffensen iteration there, whether its bisection halves the interval or the bit patterns,
# and whether `maxeval` / `xtol` are keywords it knows at all, cannot be read from the reference.  For 64 (V_T, u)
# pairs — 48 as the generator gives them, 16 with the uniform pushed far into a tail so that the fall-back ladder
# runs — the reference's OWN inverse_cdf is called with a CDF closure that records every abscissa it is asked for.
# tools/check_reference_replay.py runs the CPU restatement in every reading and prints
# `VERDICT bk_root_form = …, bk_bracket_form = …, bk_caps = …`.  Everything up to the closure is the reference's own
# code (HestonCFIterator, moments_from_cf, cdf_from_cf); the five lines between them are sample_from_cf.jl:31-37.
let N = 64
    law = marginal_law(hprob, HestonDynamics(), hprob.payoff.expiry)
    rng = Xoshiro(20240611)
    VT = Vector{Float64}(undef, N); U = similar(VT); G0 = similar(VT); GM = similar(VT); HH = similar(VT); SOL = similar(VT)
    counts = Vector{Int32}(undef, N); xs_all = Float64[]; threw = Vector{Int32}(undef, N)
    for i in 1:N
        VT[i] = Hedgehog.sample_V_T(rng, law)
        u = Distributions.rand(rng, Uniform(0, 1))
        i > 48 && (u = isodd(i) ? 1e-9 * (i - 47) : 1 - 1e-9 * (i - 47))
        U[i] = u
        ϕ = Hedgehog.HestonCFIterator(VT[i], law)
        m1, variance = Hedgehog.moments_from_cf(ϕ)         # (`m1`: `mean` is Statistics.mean further down this file)
        σ² = max(variance, 1e-12)
        normal_sample = m1 + sqrt(σ²) * quantile(Normal(), u)
        G0[i] = normal_sample > 0 ? normal_sample : m1 * 0.01
        GM[i] = m1 + 11 * sqrt(σ²)
        HH[i] = π / (m1 + 5 * √σ²)
        xs = Float64[]
        cdf = x -> (push!(xs, Float64(x)); Hedgehog.cdf_from_cf(ϕ, x, HH[i]))
        threw[i] = 0
        SOL[i] = try
            Hedgehog.inverse_cdf(cdf, u, G0[i], GM[i])
        catch                                     # an exception that escapes inverse_cdf's own catch block
            threw[i] = 1
            NaN
        end
        counts[i] = length(xs)
        append!(xs_all, xs)
    end
    T = yearfrac(hprob.market_inputs.rate.reference_date, hprob.payoff.expiry)
    add_case!("bk_root_probe"; kind = "bk_root_probe", n = N, model = model_json(heston, T),
              VT = wbin("bk_root_probe.VT.bin", VT), u = wbin("bk_root_probe.u.bin", U),
              initial_guess = wbin("bk_root_probe.guess.bin", G0), max_guess = wbin("bk_root_probe.max_guess.bin", GM),
              h = wbin("bk_root_probe.h.bin", HH), sol = wbin("bk_root_probe.sol.bin", SOL),
              counts = wbin("bk_root_probe.counts.bin", counts), threw = wbin("bk_root_probe.threw.bin", threw),
              xs = wbin("bk_root_probe.xs.bin", xs_all),
              layout = "VT, u, initial_guess, max_guess, h, sol: Float64[n]; counts, threw: Int32[n]; xs: Float64[sum(counts)] — " *
                       "trajectory i's requests are the counts[i] numbers behind those of the trajectories before it")
end

if !(length(ARGS) >= 2 && ARGS[2] == "probe")
export_euler("heston_euler", hprob, heston, HestonDynamics(), 20_000, 252)
export_euler("heston_euler_antithetic", hprob, heston, HestonDynamics(), 5_000, 100; antithetic = true)
export_euler("heston_euler_greeks", hprob, heston, HestonDynamics(), 5_000, 100;
             lenses = ("S0" => (@optic _.market_inputs.spot), "V0" => (@optic _.market_inputs.V0)))
export_euler("lognormal_euler", bprob, bs, LognormalDynamics(), 20_000, 100;
             lenses = ("S0" => (@optic _.market_inputs.spot),))

# ---- exact lognormal law: the standard normals of the one Xoshiro(seeds[1]) stream -------------------
let N = 100_000
    cfg = SimulationConfig(N; seeds = collect(UInt64, 1:N))
    method = MonteCarlo(LognormalDynamics(), BlackScholesExact(), cfg)
    sol = solve(bprob, method)
    law = marginal_law(bprob, LognormalDynamics(), bprob.payoff.expiry)      # Normal(μ̃, σ̃), montecarlo.jl:302
    z = (log.(sol.ensemble) .- mean(law)) ./ std(law)
    T = yearfrac(bprob.market_inputs.rate.reference_date, bprob.payoff.expiry)
    add_case!("exact_lognormal"; kind = "exact_lognormal", n_paths = N, model = model_json(bs, T),
              compat_sqrt_alpha = true, price = sol.price, z = wbin("exact_lognormal.z.bin", collect(Float64, z)),
              ST = wbin("exact_lognormal.ST.bin", collect(Float64, sol.ensemble)),
              law_mean = mean(law), law_std = std(law))
end

# ---- Broadie–Kaya: the three draws per trajectory, in the reference's order ----------------------------
let N = 20_000
    cfg = SimulationConfig(N; seeds = collect(UInt64, 1:N))
    method = MonteCarlo(HestonDynamics(), HestonBroadieKaya(), cfg)
    law = marginal_law(hprob, HestonDynamics(), hprob.payoff.expiry)         # LogHestonDistribution
    rng = Xoshiro(cfg.seeds[1])                                             # montecarlo.jl:456
    draws = Array{Float64}(undef, N, 3)                                     # column-major: [V_T | u | Z]
    logS = Vector{Float64}(undef, N)
    for i in 1:N
        probe = copy(rng)                          # the same stream, read ahead without consuming it
        draws[i, 1] = Hedgehog.sample_V_T(probe, law)                       # heston.jl:125-133
        draws[i, 2] = Distributions.rand(probe, Uniform(0, 1))              # sample_from_cf.jl:29
        draws[i, 3] = randn(probe)                                          # heston.jl:296
        x = Distributions.rand(rng, law)                                    # heston.jl:246-259: [log S_T, V_T]
        logS[i] = x[1]
        x[2] == draws[i, 1] || error("draw order differs from heston.jl:246-259 at trajectory $i")
    end
    sol = solve(hprob, method)
    T = yearfrac(hprob.market_inputs.rate.reference_date, hprob.payoff.expiry)
    add_case!("broadie_kaya"; kind = "bk", n_paths = N, model = model_json(heston, T), price = sol.price,
              draws = wbin("broadie_kaya.draws.bin", draws), ST = wbin("broadie_kaya.ST.bin", exp.(logS)),
              layout = "draws: [V_T | u | Z], n_paths each")
end

# ---- LSM: the spot grid the regression sees, and the reference's stopping decisions -----------------
let N = 20_000, M = 50, degree = 3
    aprob = PricingProblem(VanillaOption(100.0, expiry, American(), Put(), Spot()),
                           BlackScholesInputs(ref, bs.r, bs.S0, bs.σ))
    cfg = SimulationConfig(N; steps = M, seeds = collect(UInt64, 1:N))
    lsm = LSM(LognormalDynamics(), BlackScholesExact(), cfg, degree)
    sol = solve(aprob, lsm)                                                  # least_squares_montecarlo.jl:99-136
    grid = sol.spot_paths                                                    # (M+1) x N, Julia column-major
    T = yearfrac(ref, expiry)
    disc = df(aprob.market_inputs.rate, Hedgehog.add_yearfrac(ref, T / M))  # :107
    add_case!("lsm_put"; kind = "lsm", n_paths = N, n_steps = M, degree = degree, strike = 100.0, cp = -1.0,
              step_discount = disc, price = sol.price,
              grid = wbin("lsm_put.grid.bin", permutedims(grid)),            # -> C [step][path]
              tau = wbin("lsm_put.tau.bin", Int32[t for (t, _) in sol.stopping_info]),
              val = wbin("lsm_put.val.bin", Float64[v for (_, v) in sol.stopping_info]),
              layout = "grid: [n_steps+1][n_paths]; tau Int32, val Float64: stopping_info")
end

end  # probe only

open(joinpath(outdir, "manifest.json"), "w") do io
    print(io, "{\"generated_by\": \"julia/parity_replay.jl on Hedgehog.jl (REFERENCE output)\",\n \"cases\": [\n  ",
          join(cases, ",\n  "), "\n ]}\n")
end
println("wrote $(length(cases)) cases to $outdir")
