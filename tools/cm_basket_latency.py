#!/usr/bin/env python3
"""The calibration objective's inner loop (calibration.jl:75-88): K Heston quotes priced by Carr–Madan —
one hh_carr_madan call per quote (the reference's basket.jl:35-38 loop), one hh_carr_madan_basket call,
one hh_carr_madan_basket_grad call (prices + the gradient a ForwardDiff objective needs).  Wall time per
objective evaluation.  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import hedgehog_jl_amd as hh
from hedgehog_jl_amd.dual import Dual

ref = hh.Date(2021, 1, 1)
expiries = [hh.Date(2021, 4, 1), hh.Date(2021, 7, 1), hh.Date(2022, 1, 1), hh.Date(2023, 1, 1)]
method = hh.CarrMadan(1.0, 32.0, hh.HestonDynamics())


def wall(f, reps=30):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e3


for per_expiry in (6, 25, 250):
    strikes = np.linspace(70.0, 140.0, per_expiry)
    payoffs = [hh.VanillaOption(float(K), e, hh.European(), hh.Call(), hh.Spot()) for e in expiries for K in strikes]
    mkt = hh.HestonInputs(ref, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7)
    e5 = lambda j: tuple(1.0 if i == j else 0.0 for i in range(5))
    mkt_d = hh.HestonInputs(ref, 0.03, 100.0, Dual(0.04, e5(0)), Dual(2.0, e5(1)), Dual(0.04, e5(2)),
                            Dual(0.3, e5(3)), Dual(-0.7, e5(4)))
    basket, basket_d = hh.BasketPricingProblem(payoffs, mkt), hh.BasketPricingProblem(payoffs, mkt_d)
    t_loop = wall(lambda: [hh.solve(hh.PricingProblem(p, mkt), method).price for p in payoffs], reps=5)
    t_one = wall(lambda: hh.solve(basket, method))
    t_grad = wall(lambda: hh.solve(basket_d, method))
    print(f"{len(payoffs):5d} quotes: one solve per quote {t_loop:8.3f} ms | basket, one launch {t_one:7.3f} ms | "
          f"basket with the 5-parameter gradient {t_grad:7.3f} ms", flush=True)
