import sys, time, dataclasses
sys.path.insert(0, '/root/repo')
import numpy as np
import hedgehog_jl_amd as hh
from datetime import date
ref = date(2021, 1, 1)
prob = hh.PricingProblem(hh.VanillaOption(100.0, date(2022, 1, 1), hh.European(), hh.Call(), hh.Spot()),
                         hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7))
n = 1_000_000
cfg = hh.SimulationConfig(n, steps=252, seeds=np.arange(1, n + 1))
mc = hh.MonteCarlo(hh.HestonDynamics(), hh.EulerMaruyama(), cfg)
for label, m in (("one ctx", mc), ("devices=(0,)", dataclasses.replace(mc, devices=(0,))), ("devices=(0,0)", dataclasses.replace(mc, devices=(0, 0)))):
    for ens in (False,):
        from hedgehog_jl_amd.montecarlo import solve_montecarlo
        t = []
        for _ in range(8):
            t0 = time.perf_counter(); s = solve_montecarlo(prob, m, ensemble=ens); t.append((time.perf_counter() - t0) * 1e3)
        print(f"{label:16s} ensemble={ens}: wall median {np.median(t[2:]):.3f} ms  kernel {s.result.kernel_ms:.3f} ms price {s.price:.6f}")
