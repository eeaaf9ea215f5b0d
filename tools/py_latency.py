import sys, time; sys.path.insert(0,'.')
import numpy as np, hedgehog_jl_amd as hh
ref=hh.Date(2021,1,1)
prob=hh.PricingProblem(hh.VanillaOption(100.0,hh.Date(2022,1,1),hh.European(),hh.Call(),hh.Spot()),hh.HestonInputs(ref,0.03,100.0,0.04,2.0,0.04,0.3,-0.7))
n=10_000
mc=hh.MonteCarlo(hh.HestonDynamics(),hh.EulerMaruyama(),hh.SimulationConfig(n,steps=100,seeds=np.arange(1,n+1)))
for ens in (False, True):
    for _ in range(20): hh.solve(prob,mc,ensemble=ens)
    t=time.perf_counter()
    for _ in range(300): s=hh.solve(prob,mc,ensemble=ens)
    print("ensemble",ens,(time.perf_counter()-t)/300*1e6,"us per hh.solve; C-ABI total_ms",s.result.total_ms*1e3,"us")
lenses=(hh.optic("market_inputs.spot"),hh.optic("market_inputs.V0"),hh.optic("market_inputs.rate.rate"))
t=time.perf_counter()
for _ in range(200): g=hh.solve(hh.BatchGreekProblem(prob,lenses),hh.ForwardAD(),mc)
print("batch greeks",(time.perf_counter()-t)/200*1e6,"us")
