"""Soak: hh_mgpu_solve over 1-5 shards (device 0 listed several times -> host ordered sum) against the
single solve, random ensembles / strategies / variants: every terminal sample equal, sums to 1e-13.
(GPU box: python tools/soak_mgpu.py [seed] [cases])"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o
from tests.test_gpu_parity import gpu_solve, HESTON_SEEDS
ctx = _ffi.get_context(0)
GBM, HES = _ffi.HH_LOGNORMAL, _ffi.HH_HESTON
EM, EXACT, BK = _ffi.HH_EULER_MARUYAMA, _ffi.HH_EXACT_LAW, _ffi.HH_BROADIE_KAYA
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mgs = {g: _ffi.MultiGpu([0] * g) for g in (1, 2, 3, 5)}
if os.environ.get("HH_SOAK_SERIAL"):
    for mg_ in mgs.values():
        mg_.set_option(_ffi.HH_MGPU_OPT_ENQUEUE, _ffi.HH_MGPU_ENQUEUE_SERIAL)
bad = 0
for it in range(N):
    kind = rng.choice(["heston_em", "gbm_em", "gbm_exact", "heston_bk", "heston_replay_pm", "heston_replay_tile",
                       "gbm_exact_replay", "gbm_em_replay_pm"])
    n = int(rng.choice([rng.integers(1, 40), rng.integers(1, 3000), rng.integers(1, 40000)]))
    steps = int(rng.integers(1, 60)); anti = int(rng.random() < 0.4); g = int(rng.choice([1, 2, 3, 5]))
    duals = 0; sd = None
    seeds = rng.integers(1, 2**63, n).astype(np.uint64)
    kw = {}
    if kind == "heston_em":
        duals = int(rng.choice([0, 3])); sd = HESTON_SEEDS if duals else None
        m = o.make_model(seeds=sd, n_partials=duals); c = o.make_config(HES, EM, n, steps, antithetic=anti, seeds=seeds, n_partials=duals)
    elif kind == "gbm_em":
        m = o.make_model(sigma=0.2); c = o.make_config(GBM, EM, n, steps, antithetic=anti, seeds=seeds)
    elif kind == "gbm_exact":
        m = o.make_model(sigma=0.2); c = o.make_config(GBM, EXACT, n, 1, antithetic=anti, seeds=seeds[:1])
    elif kind == "gbm_exact_replay":  # one normal per trajectory: shard slices start on odd elements
        m = o.make_model(sigma=0.2)
        c = o.make_config(GBM, EXACT, n, 1, antithetic=anti, noise_mode=_ffi.HH_NOISE_REPLAY, replay=rng.standard_normal(n),
                          replay_layout=int(rng.integers(0, 2)))
    elif kind == "gbm_em_replay_pm":  # odd and even rows of the reference's layout
        m = o.make_model(sigma=0.2)
        c = o.make_config(GBM, EM, n, steps, antithetic=anti, noise_mode=_ffi.HH_NOISE_REPLAY,
                          replay=rng.standard_normal((n, steps)) / np.sqrt(steps), replay_layout=_ffi.HH_REPLAY_PATH_MAJOR)
    elif kind == "heston_bk":
        m = o.make_model(); c = o.make_config(HES, BK, n, 1, antithetic=0, seeds=seeds[:1])
    else:
        pm = rng.standard_normal((n, steps, 2)) / np.sqrt(steps)
        m = o.make_model()
        if kind == "heston_replay_pm":
            c = o.make_config(HES, EM, n, steps, antithetic=anti, noise_mode=_ffi.HH_NOISE_REPLAY, replay=np.ascontiguousarray(pm), replay_layout=_ffi.HH_REPLAY_PATH_MAJOR)
        else:
            nt = (n + 255) // 256; pad = np.zeros((nt * 256, steps, 2)); pad[:n] = pm
            tiled = np.ascontiguousarray(pad.reshape(nt, 256, steps, 2).transpose(0, 2, 3, 1)).ravel()
            c = o.make_config(HES, EM, n, steps, antithetic=anti, noise_mode=_ffi.HH_NOISE_REPLAY, replay=tiled)
    r1, t1 = gpu_solve(ctx, m, c)
    t2 = np.zeros(n * (2 if c.antithetic else 1))
    try:
        r2 = mgs[g].solve(m, c, t2)
    except Exception as e:
        bad += 1; print("ERROR", kind, n, steps, anti, g, repr(e)[:200], flush=True); continue
    rel = lambda a, b: abs(a - b) <= 1e-13 * max(abs(a), abs(b), 1e-300)
    ok = np.array_equal(t1, t2) and rel(r1.price, r2.price) and rel(r1.sumsq_payoff, r2.sumsq_payoff) and \
        all(rel(r1.dprice[k], r2.dprice[k]) for k in range(duals)) and r1.n_paths_done == r2.n_paths_done
    if not ok:
        bad += 1; print("MISMATCH", kind, dict(n=n, steps=steps, anti=anti, g=g, duals=duals), r1.price, r2.price, flush=True)
print(f"{N} cases, {bad} failures")
