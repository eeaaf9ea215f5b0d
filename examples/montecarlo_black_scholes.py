#!/usr/bin/env python3
"""Port of /root/reference/examples/montecarlo_black_scholes.jl to the host mirror:
European put, BlackScholesInputs, MonteCarlo(EulerMaruyama) 10^4 paths x 100 steps, then delta and
rho by ForwardAD and by finite differences (BASELINE.json configs[0] plumbing).  Needs an MI355X."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hedgehog_jl_amd as hh  # noqa: E402

strike, expiry = 1.0, hh.Date(2021, 1, 1)
payoff = hh.VanillaOption(strike, expiry, hh.European(), hh.Put(), hh.Spot())
prob = hh.PricingProblem(payoff, hh.BlackScholesInputs(hh.Date(2020, 1, 1), 0.03, 1.0, 0.04))

config = hh.SimulationConfig(10_000, steps=100, variance_reduction=hh.NoVarianceReduction())
mc = hh.MonteCarlo(hh.LognormalDynamics(), hh.EulerMaruyama(), config)

t0 = time.perf_counter()
sol = hh.solve(prob, mc)
print(f"price = {sol.price:.6f} +- {sol.std_error:.6f}   ({(time.perf_counter() - t0) * 1e3:.2f} ms, "
      f"kernel {sol.result.kernel_ms:.3f} ms; analytic 0.005166)")

spot_lens = hh.optic("market_inputs.spot")
delta_prob = hh.GreekProblem(prob, spot_lens)
print("delta  FD :", hh.solve(delta_prob, hh.FiniteDifference(1e-4, hh.FDForward()), mc).greek)
print("delta  AD :", hh.solve(delta_prob, hh.ForwardAD(), mc).greek, "(analytic -0.220337)")
rate_prob = hh.GreekProblem(prob, hh.ZeroRateSpineLens(1))
print("rho    AD :", hh.solve(rate_prob, hh.ForwardAD(), mc).greek)
print("rho    FD :", hh.solve(rate_prob, hh.FiniteDifference(1e-4, hh.FDForward()), mc).greek)
