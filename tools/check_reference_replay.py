#!/usr/bin/env python3
"""Per-trajectory parity of the HIP path against draws exported from the REFERENCE itself
(julia/parity_replay.jl).  Usage, on the MI355X box:

    python tools/check_reference_replay.py out_dir/manifest.json        # every case
    python tools/check_reference_replay.py tests/golden/replay_selftest/manifest.json   # the format self-test

Each case feeds the exported draws through HH_NOISE_REPLAY (or the exported spot grid through
hh_lsm_solve_grid) and prints one line `case <name>: key=value ...`; the exit code is non-zero when a
case misses its bar.  Bars: terminal samples 1e-10 relative per trajectory (fp64 arithmetic in a
different order of fused operations), price 1e-10, AD Greeks 1e-8, Broadie–Kaya samples 1e-7 on 98 %
of the trajectories (the |F(x) - u| <= 1e-4 stopping rule may flip on rounding), LSM stopping times
identical on 99.8 %.  For the Euler cases both step forms are tried: the one that matches is
StochasticDiffEq's EM() — this settles `em_split`.  The `bk_root_probe` case (every abscissa the
reference's inverse_cdf asked of its CDF) is held against the CPU restatement in every reading of Roots.jl's two
find_zero calls and ends in `VERDICT bk_root_form = …, bk_bracket_form = …, bk_caps = …` — the values to put into
hh_config (and to make the defaults, if they are not).  Run on the reference's export this is the check that turns
"parity unpinned" into "pinned"."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from hedgehog_jl_amd import _ffi  # noqa: E402  (struct builders only; the GPU context is made in main())


def _f8(base, name):
    return np.fromfile(os.path.join(base, name), dtype="<f8")


def _model(mj, **seeds):
    keys = dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1.0,
                strike=100.0, cp=1.0)
    keys.update({k: v for k, v in mj.items() if k in keys})
    return _ffi.make_model(**keys, **seeds)


VERDICT = {}  # Heston Euler cases: {em_split: (meets the bar, max relative error of the terminal samples)}


def check_euler(ctx, base, cs):
    n, steps, anti = cs["n_paths"], cs["n_steps"], int(bool(cs.get("antithetic", False)))
    heston = cs.get("dynamics", "heston") == "heston"
    dyn = _ffi.HH_HESTON if heston else _ffi.HH_LOGNORMAL
    dW = _f8(base, cs["dW"])
    assert dW.size == n * steps * (2 if heston else 1), "dW must hold [path][step][comp] doubles"
    ST = _f8(base, cs["ST"])
    assert ST.size == n * (1 + anti)
    greeks = cs.get("greeks") or {}
    names = [g for g in ("S0", "V0", "sigma", "r_drift") if g in greeks]
    P = len(names)
    seeds = {g: [1.0 if j == k else 0.0 for j in range(P)] for k, g in enumerate(names)}
    m = _model(cs["model"], seeds=seeds, n_partials=P) if P else _model(cs["model"])
    out, ok = {}, False
    for split in (1, 0):
        c = _ffi.make_config(dyn, _ffi.HH_EULER_MARUYAMA, n, steps, antithetic=anti, em_split=split,
                             noise_mode=_ffi.HH_NOISE_REPLAY, replay=dW,
                             replay_layout=_ffi.HH_REPLAY_PATH_MAJOR, n_partials=P)
        res = _ffi.hh_result()
        term = np.zeros(n * (1 + anti))
        ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(res), term.ctypes.data))
        e_s = float(np.max(np.abs(term - ST) / ST))
        e_p = abs(res.price - cs["price"]) / abs(cs["price"])
        e_g = max([abs(res.dprice[k] - greeks[g]) / abs(greeks[g]) for k, g in enumerate(names)] or [0.0])
        out[f"em_split={split}"] = f"max_rel_S={e_s:.3e},price={e_p:.3e}" + (f",greeks={e_g:.3e}" if P else "")
        good = e_s < 1e-10 and e_p < 1e-10 and e_g < 1e-8
        ok = ok or good
        if heston:
            VERDICT.setdefault(cs["name"], {})[split] = (good, e_s)
        if heston is False:
            break  # the diffusion is constant: both forms coincide
    return ok, out


def check_exact(ctx, base, cs):
    n = cs["n_paths"]
    z, ST = _f8(base, cs["z"]), _f8(base, cs["ST"])
    m = _model(cs["model"])
    c = _ffi.make_config(_ffi.HH_LOGNORMAL, _ffi.HH_EXACT_LAW, n, noise_mode=_ffi.HH_NOISE_REPLAY,
                         replay=z, compat_sqrt_alpha=int(bool(cs.get("compat_sqrt_alpha", True))))
    res = _ffi.hh_result()
    term = np.zeros(n)
    ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(res), term.ctypes.data))
    e_s = float(np.max(np.abs(term - ST) / ST))
    e_p = abs(res.price - cs["price"]) / abs(cs["price"])
    # z was recovered from the reference's samples by (x - mean)/std: one rounding each way
    return e_s < 1e-12 and e_p < 1e-12, {"max_rel_S": f"{e_s:.3e}", "price": f"{e_p:.3e}"}


def check_bk(ctx, base, cs):
    n = cs["n_paths"]
    draws, ST = _f8(base, cs["draws"]), _f8(base, cs["ST"])
    assert draws.size == 3 * n, "draws must hold [V_T | u | Z]"
    m = _model(cs["model"])
    c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, noise_mode=_ffi.HH_NOISE_REPLAY,
                         replay=draws)
    res = _ffi.hh_result()
    term = np.zeros(n)
    ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(res), term.ctypes.data))
    rel = np.abs(term - ST) / ST
    frac = float(np.mean(rel > 1e-7))
    e_p = abs(res.price - cs["price"]) / abs(cs["price"])
    return frac <= 0.02 and e_p < 1e-4, {"frac_beyond_1e-7": f"{frac:.4f}", "median_rel_S": f"{np.median(rel):.3e}",
                                         "price": f"{e_p:.3e}", "newton_fail": int(res.bk_newton_fail)}


def check_lsm(ctx, base, cs):
    n, steps = cs["n_paths"], cs["n_steps"]
    grid = _f8(base, cs["grid"])
    assert grid.size == (steps + 1) * n, "grid must hold [n_steps+1][n_paths]"
    tau_ref = np.fromfile(os.path.join(base, cs["tau"]), dtype="<i4")
    val_ref = _f8(base, cs["val"])
    dev = C.c_void_p()
    ctx.check(ctx.lib.hh_device_malloc(ctx.handle, grid.nbytes, C.byref(dev)))
    try:
        ctx.check(ctx.lib.hh_memcpy_h2d(ctx.handle, dev, grid.ctypes.data, grid.nbytes))
        m = _ffi.make_model(strike=cs["strike"], cp=cs["cp"])
        res = _ffi.hh_lsm_result()
        tau, val = np.zeros(n, dtype=np.int32), np.zeros(n)
        ctx.check(ctx.lib.hh_lsm_solve_grid(ctx.handle, C.byref(m), dev, n, steps, cs["degree"],
                                            cs["step_discount"], C.byref(res), tau.ctypes.data,
                                            val.ctypes.data))
    finally:
        ctx.lib.hh_device_free(ctx.handle, dev)
    same = tau == tau_ref
    e_v = float(np.max(np.abs(val[same] - val_ref[same]) / np.maximum(np.abs(val_ref[same]), 1e-300))) \
        if same.any() else 0.0
    e_p = abs(res.price - cs["price"]) / abs(cs["price"])
    ok = same.mean() >= 0.998 and e_v < 1e-12 and e_p < (1e-10 if same.all() else 5e-4)
    return ok, {"same_stopping_time": f"{same.mean():.5f}", "max_rel_val": f"{e_v:.3e}", "price": f"{e_p:.3e}"}


# ---- the iterate probe: which reading of Roots.jl's two find_zero calls the reference runs ---------------------------
# (CPU only: the exported abscissae against the CPU restatement run in every reading; the kernels follow the
# restatement per trajectory in each — tests/test_gpu_bk_root_forms.py)
ROOT_VERDICT = {}


def _first_seen(xs):
    """a request sequence without re-evaluations: the reference asks for func(sol), func(0), func(max_guess) a second
    time (its own acceptance test; find_zero's bracket set-up) — an abscissa counts where it is FIRST asked for"""
    seen, out = set(), []
    for x in xs:
        if x not in seen:
            seen.add(x)
            out.append(x)
    return out


def _split_search_ladder(xs, max_guess):
    """(abscissae of the first search, abscissae of the bisection behind the ladder's two end points) — the ladder
    starts where 0.0 is asked for (no iterate of a search from a positive guess is exactly 0.0)"""
    xs = _first_seen(xs)
    if 0.0 not in xs:
        return xs, None
    k = xs.index(0.0)
    return xs[:k], [x for x in xs[k + 1:] if x != max_guess]


def _same_sequence(a, b, rtol=1e-7):
    return len(a) == len(b) and all(abs(x - y) <= rtol * max(abs(x), abs(y), 1e-3) for x, y in zip(a, b))


def root_probe_readings(base, cs):
    """-> {"search": {(root_form, caps): trajectories reproduced}, "ladder": {(bracket_form, caps): …}, "n_search": …,
    "n_ladder": …} for a `bk_root_probe` case"""
    from oracle import bk_oracle as B
    n = cs["n"]
    VT, U = _f8(base, cs["VT"]), _f8(base, cs["u"])
    GM = _f8(base, cs["max_guess"])
    counts = np.fromfile(os.path.join(base, cs["counts"]), dtype="<i4")
    xs_all = _f8(base, cs["xs"])
    assert VT.size == n and U.size == n and counts.size == n and xs_all.size == int(counts.sum())
    off = np.concatenate([[0], np.cumsum(counts)])
    mj = cs["model"]
    dist = B.LogHestonDistribution(mj["S0"], mj["V0"], mj["kappa"], mj["theta"], mj["sigma"], mj["rho"], mj["r"], mj["T"])
    search = {(rf, cp): 0 for rf in (0, 1) for cp in (0, 1)}
    ladder = {(bf, cp): 0 for bf in (0, 1) for cp in (0, 1)}
    n_ladder = 0
    for i in range(n):
        ref_s, ref_l = _split_search_ladder([float(x) for x in xs_all[off[i]:off[i + 1]]], float(GM[i]))
        n_ladder += ref_l is not None
        it = B.HestonCFIterator(float(VT[i]), dist)
        got = {}
        for rf in (0, 1):
            for bf in (0, 1):
                for cp in (0, 1):
                    xs = []
                    B.sample_from_cf(float(U[i]), it, root_form=rf, bracket_form=bf, caps=cp, xs=xs)
                    got[(rf, bf, cp)] = _split_search_ladder(xs, float(GM[i]))
        for (rf, cp) in search:
            search[(rf, cp)] += _same_sequence(got[(rf, 0, cp)][0], ref_s)
        if ref_l is not None:
            for (bf, cp) in ladder:  # (the ladder is reached after the first search: read it under every first reading)
                ladder[(bf, cp)] += any(g[1] is not None and _same_sequence(g[1], ref_l)
                                        for g in (got[(rf, bf, cp)] for rf in (0, 1)))
    return {"search": search, "ladder": ladder, "n_search": n, "n_ladder": n_ladder}


def _name_reading(scores, total, names, share=0.9):
    """the reading(s) that reproduce at least `share` of the trajectories; a fork whose two readings reproduce the same
    ones was not exercised by the sample"""
    best = max(scores.values())
    if total == 0 or best < share * total:
        return None, best
    return sorted(k for k, v in scores.items() if v == best), best


def check_root_probe(ctx, base, cs):
    r = root_probe_readings(base, cs)
    hit_s, best_s = _name_reading(r["search"], r["n_search"], None)
    hit_l, best_l = _name_reading(r["ladder"], r["n_ladder"], None)
    ROOT_VERDICT[cs["name"]] = (r, hit_s, hit_l)
    ok = hit_s is not None and (r["n_ladder"] == 0 or hit_l is not None)
    return ok, {"search": ",".join(f"root{rf}/caps{cp}:{v}" for (rf, cp), v in sorted(r["search"].items())) + f"/{r['n_search']}",
                "ladder": ",".join(f"bracket{bf}/caps{cp}:{v}" for (bf, cp), v in sorted(r["ladder"].items())) + f"/{r['n_ladder']}"}


def root_verdict_line(name):
    r, hit_s, hit_l = ROOT_VERDICT[name]
    ROOT = {0: "0 (secant)", 1: "1 (Roots' Order2: Steffensen guarded by secant)"}
    BRK = {0: "0 (arithmetic midpoint to atol)", 1: "1 (Roots' Bisection over bit patterns, to the last bit)"}
    CAP = {0: "0 (maxeval honoured)", 1: "1 (maxeval ignored)"}

    def say(hit, table, what):
        if hit is None:
            return f"{what} undecided (no reading reproduces 90 % of the trajectories)", "?"
        forms = sorted({k[0] for k in hit})
        caps = sorted({k[1] for k in hit})
        f = table[forms[0]] if len(forms) == 1 else "either (the sample does not tell them apart)"
        c = CAP[caps[0]] if len(caps) == 1 else "either (no trajectory of the sample reaches a cap)"
        return f"{what} = {f}", c
    a, ca = say(hit_s, ROOT, "bk_root_form")
    b, cb = (f"bk_bracket_form: no trajectory of the sample reached the ladder", "?") if r["n_ladder"] == 0 else \
        say(hit_l, BRK, "bk_bracket_form")
    caps = ca if ca == cb or cb == "?" else (cb if ca.startswith("either") else ca if cb.startswith("either")
                                            else f"first search {ca}, bisection {cb}")
    return f"VERDICT {a}, {b}, bk_caps = {caps} ({name}: {r['n_search']} trajectories, {r['n_ladder']} through the ladder)"


CHECKS = {"euler": check_euler, "exact_lognormal": check_exact, "bk": check_bk, "lsm": check_lsm,
          "bk_root_probe": check_root_probe}


def main(path):
    man = json.load(open(path))
    base = os.path.dirname(os.path.abspath(path))
    if "cases" not in man:  # round-1 single-case file: Heston Euler, no variance reduction
        man = {"cases": [dict(man, name="heston_euler", kind="euler", dynamics="heston",
                              model={k: man[k] for k in ("S0", "strike", "r", "V0", "kappa", "theta",
                                                         "sigma", "rho", "T", "cp")})]}
    need_gpu = any(cs["kind"] != "bk_root_probe" for cs in man["cases"])
    ctx = None
    if need_gpu:  # (a manifest that holds only iterate probes is checked on the CPU)
        import hedgehog_jl_amd as hh
        ctx = hh.get_context(0)
    bad = 0
    for cs in man["cases"]:
        ok, info = CHECKS[cs["kind"]](ctx, base, cs)
        bad += not ok
        print(f"case {cs['name']} [{cs['kind']}] {'OK' if ok else 'MISMATCH'}: " +
              " ".join(f"{k}={v}" for k, v in info.items()), flush=True)
    # the one line a maintainer with a Julia host is asked for (the first Heston Euler case decides)
    for name, v in VERDICT.items():
        hit = [sp for sp, (good, _) in v.items() if good]
        errs = ", ".join(f"em_split = {sp}: {e:.1e}" for sp, (_, e) in sorted(v.items()))
        if len(hit) == 1:
            print(f"VERDICT em_split = {hit[0]} matches the reference ({name}: max relative error of S_T — {errs})")
        else:
            print(f"VERDICT em_split undecided on {name} ({errs})")
        break
    for name in ROOT_VERDICT:
        print(root_verdict_line(name))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1]) else 0)
