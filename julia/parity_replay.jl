# parity_replay.jl — exports the REFERENCE's own Wiener draws and results so that the HIP path can be
# checked per trajectory against Hedgehog.jl itself (closes the "parity unpinned" gap of DESIGN.md §2).
#
# NOT RUN in this build: there is no Julia on the build image or on the GPU box.  On a host with
# Julia + Hedgehog.jl:
#
#     julia --project julia/parity_replay.jl out_dir            # writes dW.bin, ST.bin, meta.json
#     python tools/check_reference_replay.py out_dir/meta.json  # on the MI355X box
#
# Mechanism: simulate with the reference, saving the noise — exactly what the reference itself does
# for antithetic replay (montecarlo.jl:370) — and export per-trajectory increments diff(W.W); the
# kernels consume them through HH_NOISE_REPLAY / HH_REPLAY_PATH_MAJOR.  Comparing both em_split
# settings also settles which step form StochasticDiffEq's EM() uses.
using Hedgehog, StochasticDiffEq, Dates

outdir = length(ARGS) >= 1 ? ARGS[1] : "replay_out"
mkpath(outdir)

ref = Date(2021, 1, 1); expiry = Date(2022, 1, 1)
S0, K, r, V0, κ, θ, σ, ρ = 100.0, 100.0, 0.03, 0.04, 2.0, 0.04, 0.3, -0.7
prob = PricingProblem(VanillaOption(K, expiry, European(), Call(), Spot()),
                      HestonInputs(ref, r, S0, V0, κ, θ, σ, ρ))
N, M = 20_000, 252
cfg = SimulationConfig(N; steps = M, seeds = collect(UInt64, 1:N))
method = MonteCarlo(HestonDynamics(), EulerMaruyama(), cfg)

sde = Hedgehog.sde_problem(prob, method)
ens = StochasticDiffEq.solve(Hedgehog.get_ensemble_problem(sde, cfg), EM();
                             dt = sde.tspan[2] / M, trajectories = N, save_noise = true)

# [comp, step, path] in Julia's column-major order == [path][step][comp] in C order
dW = Array{Float64}(undef, 2, M, N)
for i in 1:N, s in 1:M
    dW[:, s, i] .= ens.u[i].W.W[s + 1] .- ens.u[i].W.W[s]
end
S_ref = Hedgehog.final_sample(ens)                       # montecarlo.jl:398
payoffs = Hedgehog.reduce_payoffs(S_ref, prob.payoff, cfg.variance_reduction)
price_ref = df(prob.market_inputs.rate, prob.payoff.expiry) * sum(payoffs) / N

write(joinpath(outdir, "dW.bin"), dW)
write(joinpath(outdir, "ST.bin"), S_ref)
open(joinpath(outdir, "meta.json"), "w") do io
    print(io, """{"n_paths": $N, "n_steps": $M, "S0": $S0, "strike": $K, "r": $r, "V0": $V0,
 "kappa": $κ, "theta": $θ, "sigma": $σ, "rho": $ρ, "T": $(sde.tspan[2]), "cp": 1.0,
 "price": $price_ref, "dW": "dW.bin", "ST": "ST.bin", "layout": "path-major [path][step][comp] float64 LE"}""")
end
println("wrote $outdir: price_ref = $price_ref")
