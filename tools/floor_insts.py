#!/usr/bin/env python3
"""Algorithmic instruction floors of the VALU-bound rows of bench.py -> profiles/floor_insts.json.

A `bound: "valu"` roofline entry prices the kernel's OWN instruction count (SQ_INSTS_VALU) against the
issue peak: it says how busy the vector pipe is, not how far the kernel is from what its algorithm needs.
This file states the second number: the operations the shipped ALGORITHM requires per unit of work, each
counted as ONE VALU instruction per lane — every fp64 fma / mul / add, conversion, compare, select,
32x32->64 multiply and three-way xor that a hand-scheduled kernel could not avoid — with loop-invariant
work hoisted and nothing charged for moves, address arithmetic, waits or loop control.  Derivations: the
tables below, statement by statement against hedgehog.jl_amd/csrc/hh_rng.h, hh_math.h, hh_kernels.hip,
hh_bessel.h, hh_bk.hip, hh_lsm.hip (DESIGN.md §8 walks through them).  bench.py reports
    floor_insts_per_unit   (wave-instructions per unit = lane count / 64, the unit of valu_insts.json)
    frac_of_floor        = floor_insts_per_unit / measured valu_insts_per_unit
beside `frac` (issue-slot utilisation); frac x frac_of_floor is the distance from the floor at peak issue.

CPU only (numpy / scipy).  usage: python tools/floor_insts.py [--write]"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- Box–Muller on one Philox block: two normals (hh_rng.h) ---------------------------------------------
PHILOX = {  # per block, counter (step, 0, 0, domain), round keys k + r·W hoisted out of the step loop
    "rounds 3-10: 2 v_mad_u64_u32 + 2 v_bitop3 each": 32,
    "round 1 (counter words 1, 2 are zero, word 0 uniform): one xor": 1,
    "round 2 (one product loop-invariant): one multiply, two xors": 3,
}
UNIFORMS = {"two uniforms: 64-bit shift, or with the exponent, one exact fp64 subtraction, each": 6}
NEG2LOG = {  # neg2_log_unit
    "frexp mantissa / exponent": 2, "m < sqrt(1/2): compare, rescale m (select + ldexp), e - 1 with borrow": 4,
    "f = m - 1, d = m + 1": 2, "1/d: v_rcp_f64 + two Newton steps": 5, "s = f/d with residual correction": 3,
    "z = s^2": 1, "nine Horner steps": 9, "e -> fp64": 1, "s z, e ln2_lo, three fma of the recombination (the -2 is in the coefficients)": 5,
}
SQRT_POS = {"v_rsq_f64, g, h": 3, "two coupled Newton steps": 6, "final correction": 2}
SINCOSPI = {
    "q = rint(2t), r = t - q/2, z = r^2": 3, "sine polynomial: 7 Horner steps + r p": 8, "cosine polynomial: 8 steps": 8,
    "q -> int, swap test, four 32-bit selects": 7, "two sign fix-ups (shift, add, two v_bitop3 on the high words)": 4,
}
NORMAL_PAIR = {"Philox4x32-10": sum(PHILOX.values()), "uniforms": sum(UNIFORMS.values()),
               "-2 ln u1": sum(NEG2LOG.values()), "sqrt": sum(SQRT_POS.values()),
               "sincospi(2 u2)": sum(SINCOSPI.values()), "z1 = r c, z2 = r s": 2}

# ---- Heston Euler step (HestonModel<0, true>::step, heston.jl:7-16) -------------------------------------
HESTON_STEP = {
    "v+ = max(v, 0); theta - v+; r - v+/2; Kx; kappa (theta - v+); Kv; w = max(Kv, 0)": 7,
    "sqrt(w) clipped at 0: v_rsq_f64, v_min_f64 on the seed, one coupled Newton step, two corrections": 11,
    "sigma sqrt(w); x' and v' (two fma)": 3,
}
GENERATE = {"two normals": sum(NORMAL_PAIR.values()),
            "dW1 = sqrt(dt) z1, dW2 = sqrt(dt) (rho z1 + rho_c z2)": 4,
            "Euler step": sum(HESTON_STEP.values())}

# ---- exact lognormal law, one antithetic pair (exact_gbm_kernel) -----------------------------------------
FM_EXP = {"clamp": 2, "k = rint(x log2 e), two-term reduction": 4, "12 Horner steps": 12, "1 + r + r^2 q": 3,
          "k -> int, ldexp, NaN select": 4}

# ---- LSM, per (trajectory, date), degree D = 5 (hh_lsm.hip; pipeline of four rows) -----------------------
D = 5
LSM_INDUCTION = {
    "row t-3 statistics: payoff, in-the-money test, masked x and x^2 into the sums": 6,
    "row t-2 power sums: z = (x - mean)/std (1), z^2 .. z^2D (2D-1), masked adds (2D+1)": 1 + (2 * D - 1) + (2 * D + 1),
    "row t-1 moment sums: z (1), y = D(t) val (1), z^k y k = 1..D (D), masked adds (D+1)": 2 + D + (D + 1),
    "row t decision: z (1), Horner (D), payoff and test (3), compare (1), val / tau selects (3), discount (1)": 9 + D,
    # wave_reduce_multi (hh_lsm.hip) transposes while it reduces: P2 values cost P2/2 + P2/4 + … + 1 + log2(64/P2)
    # exchanges, each two 32-bit swaps and one add — groups of 4, 16 and 8 values: 7 + 17 + 10 = 34 exchanges
    "wave butterflies of the three groups (34 exchanges x 3 instructions), over 64 lanes x 16 trajectories":
        round(34 * 3 * 64 / 1024.0, 2),
}
GBM_GRID = {  # gbm_grid_kernel<ANTI>: one normal and ONE exponential per antithetic PAIR and date
    "normal (half a Box–Muller pair) / 2 trajectories": round(sum(NORMAL_PAIR.values()) / 2 / 2, 2),
    "a + b z, exp, S *= e, mirrored S from the shared exponential (e2a / e) / 2 trajectories": round((2 + sum(FM_EXP.values()) + 1 + 7) / 2, 2),
    "stores": 1,
}

# ---- Broadie–Kaya: one CF evaluation (evaluate_chf, heston.jl:184-212) -----------------------------------
FM_SINCOS = {"n = rint(2x/pi), two-term reduction": 4, "z, two 5-step polynomials, sin and cos assembly": 19,
             "n -> int, quadrant swap and signs": 10, "|x| <= 2^20 test": 1}
CEXP = sum(FM_EXP.values()) + sum(FM_SINCOS.values()) + 2
CSQRT = 13 + 2 + 11 + 6 + 1          # |z| (fma, mul, sqrt 11), (r + re)/2, sqrt, im / (2t) (rcp 5 + mul), product
ATAN2 = {"|.| compare, four selects, t = num rcp(den)": 11, "break points: two compares, num / den / hi / lo selects": 12,
         "u = num rcp(den)": 6, "z, w, two polynomial halves (6 + 5), assembly (5)": 18, "octant / sign fix-ups": 8}
CF_FIXED = {
    "gamma = sqrt(kappa^2 - 2 i sigma^2 a)": 1 + CSQRT, "exp(-gamma T/2) (complex exp)": 2 + CEXP,
    "e = eh^2, 1 - e, 1 + e": 6, "1/(1 - e), gamma/(1 - e), eta_gamma, nu_gamma": 9 + 4 + 4 + 6,
    "theta = atan2(nu_gamma)": sum(ATAN2.values()), "continuous unwrapping (heston.jl:198-205)": 8,
    "Bessel: reflection test, |z|, dispatch": 6 + 13 + 4,
    "Bessel series: q = z^2/4, Q = q^2, length n0 + n1 r, wave maximum": 10 + 4 + 12,
    "Bessel series: S = A + c1 q B, nu log(r/2) - lgamma, nu phi": 8 + 33 + 3,
    "exponent of phi (heston.jl:207-211)": 12, "exp of it (complex exp), x I.mul, x zeta_kappa gamma/(1 - e)": CEXP + 4 + 2 + 4,
}
CF_PER_BESSEL_TERM = 6   # two independent Horner chains (even / odd half), 12 instructions per two terms
# the two real-axis evaluations of a trajectory's set-up (round 5: besseli_logmul_re, chf_at_zero in hh_bk.hip): the
# Bessel function of a positive real argument is 2 instructions per series term, and phi(0) needs of the model only
# four numbers made once per chain (bk_tables_kernel)
BESSEL_RE_FIXED = {"|x|, x^2, rough root, dispatch": 6, "q = x^2/4, Q = q^2": 2, "length n0 + n1 r, wave maximum": 12,
                   "S = A + c1 q B": 3, "(nu/2) log(r^2/4) - lgamma": 33 + 2}
BESSEL_RE_PER_TERM = 2
CHF_ZERO = {"nu_gamma(0) = (4 sqrt(V0 VT)/sigma^2) x_re": 2, "exponent c0 + sumV c1 + log I - log I_kappa": 4,
            "exp": sum(FM_EXP.values()), "x mul x w_re": 2}
LOG_I_KAPPA = {"nu_kappa": 1, "log(mul), sum": 33 + 1}
# one CDF evaluation of the root search on the cached terms (cdf_cached): sin, cos of h x once, then per term ONE fma of
# the sum and ONE of the three-term recurrence sin((j+1)t) = 2 cos t sin(jt) - sin((j-1)t)  (round 4: a rotation, 7 per
# term with the weight's product; the weight is now folded into the register terms once per trajectory)
CDF_FIXED = sum(FM_SINCOS.values()) + 3
CDF_PER_CACHED_TERM = 2
DRAWS = {  # bk_draw_kernel: Z and u (one block + Box–Muller + uniforms), the NCchi^2 variance (d > 1: one gamma by
    # Marsaglia–Tsang: a normal, a uniform, one squeeze test; + shift normal), normal quantile of u
    "Philox blocks (3)": 3 * sum(PHILOX.values()), "Box–Muller x 2": 2 * (sum(NORMAL_PAIR.values()) - sum(PHILOX.values())),
    "gamma_mt: d, c (rcp, sqrt), v^3, squeeze": 30, "chi^2 assembly, V_T": 8, "normcdfinv(u) (rational, one log + sqrt tail)": 60,
}


def bessel_series_terms(nu, r):
    """the length rule of bessel_table(): smallest N past the largest term with T_{N+1} < 2^-57 max T_k"""
    q, term, largest, k_l = 0.25 * r * r, 1.0, 1.0, 0
    for k in range(1, 400):
        term *= q / (k * (k + nu))
        if term > largest:
            largest, k_l = term, k
        elif k > k_l and term < 2.0 ** -57 * largest:
            return k - 1
    return 63


def bk_sample(kappa=2.0, theta=0.04, sigma=0.3, V0=0.04, T=1.0, n=4000, cf_tol=1e-3, n_sigma=5.0, hm=1e-2, seed=1):
    """Config 4 on a sample of V_T: series length J of the CDF (sample_from_cf.jl:88), Bessel terms per CF
    evaluation — the two data-dependent lengths the floor of a trajectory depends on (scipy only)."""
    from scipy import special, stats
    s2 = sigma * sigma
    d = 4 * kappa * theta / s2
    em = -math.expm1(-kappa * T)
    c = s2 * em / (4 * kappa)
    lam = 4 * kappa * math.exp(-kappa * T) * V0 / (s2 * em)
    VT = c * stats.ncx2.rvs(d, lam, size=n, random_state=seed)
    nu = 0.5 * d - 1.0

    def logI(z):  # log I_nu(z), Re z >= 0 here
        return np.log(special.ive(nu, z)) + np.abs(z.real)

    def cf(a, VT):
        g = np.sqrt(kappa * kappa - 2j * s2 * a)
        e = np.exp(-g * T)
        zg = (1 - e) / g
        zk = em / kappa
        eg = g * (1 + e) / (1 - e)
        ek = kappa * (1 + math.exp(-kappa * T)) / em
        sq = np.sqrt(V0 * VT)
        nug = 4 * sq * g * np.exp(-0.5 * g * T) / (s2 * (1 - e))
        nuk = 4 * sq * kappa * math.exp(-0.5 * kappa * T) / (s2 * em)
        return (np.exp(-0.5 * (g - kappa) * T) * (zk / zg) * np.exp((V0 + VT) / s2 * (ek - eg))
                * np.exp(logI(nug) - logI(nuk + 0j))), np.abs(nug)

    pp, _ = cf(hm, VT)
    p0, _ = cf(1e-300, VT)
    mean = pp.imag / hm
    var = -(2 * (pp.real - p0.real) / (hm * hm)) - mean * mean
    sd = np.sqrt(np.maximum(var, 1e-12))
    h = math.pi / (mean + n_sigma * sd)
    J = np.zeros(n, dtype=int)
    bess = np.zeros(n)
    alive = np.ones(n, dtype=bool)
    for j in range(1, 400):
        phi, r = cf(h * j, VT)
        J[alive] = j
        bess[alive] += [bessel_series_terms(nu, x) if x < 13.0 else 32 for x in r[alive]]
        alive &= np.abs(phi) / j >= math.pi * cf_tol / 2
        if not alive.any():
            break
    return float(J.mean()), float((bess / J).mean()), nu


def main():
    J, N, nu = bk_sample()
    cf_fixed = sum(CF_FIXED.values())
    cf_eval = cf_fixed + CF_PER_BESSEL_TERM * N
    evals = 6.0  # CDF evaluations of the secant on cached terms (measured mean of hh_bk_decisions & 0xff at config 4: 5.6-6.2)
    bessel_re = sum(BESSEL_RE_FIXED.values()) + BESSEL_RE_PER_TERM * N
    setup_re = sum(CHF_ZERO.values()) + bessel_re + sum(LOG_I_KAPPA.values()) + bessel_re
    # J series terms + the moment evaluation at a = h_m (complex) | phi(0) and log I(nu_kappa) (real axis) | the root
    # search | the draws | load / premultiply the register terms, finish (log S_T, exp, payoff)
    # … and the bisection ladder of the 2.2 % of trajectories whose secant fails (BENCH: 22 197 of 10^6): its two end
    # points and ~13 midpoints, as the SEQUENTIAL loop evaluates them (what the wave-walk of the tree spends beyond
    # that — the nodes of the branches not taken — is the implementation's, not the algorithm's)
    ladder = 0.0222 * 15.0 * (CDF_FIXED + J * CDF_PER_CACHED_TERM)
    bk_path = (J + 1) * cf_eval + setup_re + evals * (CDF_FIXED + J * CDF_PER_CACHED_TERM) + ladder + sum(DRAWS.values()) + 60
    bk_path_r4 = (J + 2) * cf_eval + evals * J * 7 + sum(DRAWS.values()) + 60  # the algorithm as round 4 shipped it
    lsm = sum(LSM_INDUCTION.values()) + sum(GBM_GRID.values())
    exact = (sum(NORMAL_PAIR.values()) / 2 + 3 + sum(FM_EXP.values()) + 12) / 1.0  # one normal, mu + sd z, exp, payoff; per path
    out = {
        "_what": "operations the shipped algorithm needs per unit, one VALU instruction per lane each (tools/floor_insts.py; "
                 "DESIGN.md §8); floor_insts_per_unit is in wave-instructions (lane count / 64), the unit of valu_insts.json",
        "heston_euler_generate": {"lane_insts": sum(GENERATE.values()), "floor_insts_per_unit": sum(GENERATE.values()) / 64.0,
                                  "unit": "path-step", "phases": {**{"normals: " + k: v for k, v in NORMAL_PAIR.items()},
                                                                  **{k: v for k, v in GENERATE.items() if k != "two normals"}}},
        "lognormal_exact": {"lane_insts": exact, "floor_insts_per_unit": exact / 64.0, "unit": "path"},
        "lognormal_exact_1e8": {"lane_insts": exact, "floor_insts_per_unit": exact / 64.0,
                                "unit": "path (the same law; 64 pairs per lane amortise the workgroup's reduction)"},
        # several models on the same draws (hh_multi.hip): the normals / the loads once, one Euler step per model
        "heston_euler_generate_multi2": {"lane_insts": sum(GENERATE.values()) + GENERATE["Euler step"],
                                         "floor_insts_per_unit": (sum(GENERATE.values()) + GENERATE["Euler step"]) / 64.0,
                                         "unit": "path-step of the PASS (two models stepped on it; dW formed once: same rho, dt)"},
        "heston_euler_replay_multi2": {"lane_insts": 2 * GENERATE["Euler step"],
                                       "floor_insts_per_unit": 2 * GENERATE["Euler step"] / 64.0,
                                       "unit": "path-step of the PASS (two Euler steps; the two loads are not VALU work)"},
        "broadie_kaya": {"lane_insts": bk_path, "floor_insts_per_unit": bk_path / 64.0, "unit": "path",
                         "per_cf_evaluation": {"fixed": cf_fixed, "per_bessel_series_term": CF_PER_BESSEL_TERM,
                                               "bessel_terms_mean": N, "lane_insts": cf_eval},
                         "cf_evaluations_per_path": J + 1, "real_axis_setup": setup_re,
                         "series_length_mean": J, "bessel_order": nu,
                         "cdf_evaluations_on_cached_terms": evals, "ladder_lane_insts_mean": ladder, "per_cdf_evaluation_fixed": CDF_FIXED,
                         "per_cached_term": CDF_PER_CACHED_TERM,
                         "round4_algorithm_lane_insts": bk_path_r4,
                         "draws": sum(DRAWS.values()),
                         "sample": "4000 V_T of config 4 (scipy ncx2), reference controls cf_tol 1e-3, n_sigma 5"},
        "lsm_chain": {"lane_insts": lsm, "floor_insts_per_unit": lsm / 64.0, "unit": "(trajectory, date)",
                      "induction_lane_insts": sum(LSM_INDUCTION.values()), "grid_lane_insts": sum(GBM_GRID.values()),
                      "phases": {**LSM_INDUCTION, **{"grid: " + k: v for k, v in GBM_GRID.items()}}},
    }
    for k, v in out.items():
        if isinstance(v, dict):
            print(f"{k:24s} {v['lane_insts']:9.1f} lane-instructions per {v['unit']}  = {v['floor_insts_per_unit']:.4f} wave-instructions")
    print(f"Broadie-Kaya: J = {J:.2f} series terms, {N:.1f} Bessel terms per CF evaluation, CF evaluation = {cf_eval:.0f} (fixed {cf_fixed})")
    try:
        meas = json.load(open(os.path.join(ROOT, "profiles", "valu_insts.json")))
        for k in ("heston_euler_generate", "heston_euler_generate_multi2", "heston_euler_replay_multi2", "lognormal_exact",
                  "lognormal_exact_1e8", "broadie_kaya", "lsm_chain"):
            print(f"  frac_of_floor {k:24s} {out[k]['floor_insts_per_unit'] / meas[k]['valu_insts_per_unit']:.3f}")
    except Exception as e:  # noqa: BLE001
        print("no valu_insts.json:", e)
    if "--write" in sys.argv:
        json.dump(out, open(os.path.join(ROOT, "profiles", "floor_insts.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
