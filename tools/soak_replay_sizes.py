"""Soak: GENERATE == REPLAY (tile-major, hh_wiener_fill of the same seeds) == REPLAY (path-major) at
ensemble sizes between the fixed test cases (tail-tile and deep-ring thresholds, ragged last tiles), random
step counts and variants; terminal samples to 1e-12, tile- vs path-major bit for bit.
GPU box: python tools/soak_replay_sizes.py [seed] [cases]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hedgehog_jl_amd import _ffi
from tests.test_gpu_parity import HESTON_SEEDS
ctx = _ffi.get_context(0); lib, h = ctx.lib, ctx.handle
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
bad = 0
for it in range(N):
    dyn = 1 if rng.random() < 0.7 else 0
    nc = 2 if dyn == 1 else 1
    n = int(rng.choice([rng.integers(1, 3000), rng.integers(60000, 70000), rng.integers(125000, 140000), rng.integers(250000, 270000), rng.integers(1, 600000)]))
    steps = int(rng.choice([rng.integers(1, 12), rng.integers(1, 80), rng.integers(1, 300)]))
    if n * steps > 6e7: steps = max(1, int(6e7 // n))
    anti = int(rng.random() < 0.3); P = int(rng.choice([0, 0, 1, 3])) if dyn == 1 else 0
    sd = {0: None, 1: {"V0": [1.0]}, 3: HESTON_SEEDS}[P]
    m = _ffi.make_model(seeds=sd, n_partials=P) if dyn == 1 else _ffi.make_model(sigma=0.2)
    seeds = torch.from_numpy(rng.integers(1, 2**62, n).astype(np.int64)).to(dev)
    def solve(cfg):
        res = _ffi.hh_result(); nt = n * (2 if anti else 1)
        term = torch.zeros(nt, dtype=torch.float64, device=dev)
        cfg.terminal_on_device = 1
        ctx.check(lib.hh_mc_solve(h, C.byref(m), C.byref(cfg), C.byref(res), term.data_ptr()))
        return res, term
    cg = _ffi.make_config(dyn, 0, n, steps, antithetic=anti, n_partials=P); cg.seeds, cg.seeds_on_device = seeds.data_ptr(), 1
    rg, tg = solve(cg)
    dW = torch.empty(lib.hh_replay_elems(n, steps, dyn), dtype=torch.float64, device=dev)
    ctx.check(lib.hh_wiener_fill(h, dyn, m.rho, m.T, steps, n, seeds.data_ptr(), 1, dW.data_ptr()))
    lib.hh_ctx_synchronize(h)  # the fill is asynchronous on the context's stream; torch reads dW on its own
    ct = _ffi.make_config(dyn, 0, n, steps, noise_mode=1, antithetic=anti, n_partials=P); ct.replay, ct.replay_on_device = dW.data_ptr(), 1
    rt, tt = solve(ct)
    pm = dW.view(-1, steps, nc, 256).permute(0, 3, 1, 2).reshape(-1, steps, nc)[:n].contiguous()
    torch.cuda.synchronize()   # ... and the library reads pm on the context's stream
    cp_ = _ffi.make_config(dyn, 0, n, steps, noise_mode=1, antithetic=anti, n_partials=P, replay_layout=1); cp_.replay, cp_.replay_on_device = pm.data_ptr(), 1
    rp, tp = solve(cp_)
    torch.cuda.synchronize()
    ok = torch.equal(tt, tp) and rt.price == rp.price and all(rt.dprice[k] == rp.dprice[k] for k in range(P))
    ok = ok and torch.allclose(tg, tt, rtol=1e-11, atol=0) and abs(rg.price - rt.price) <= 1e-11 * abs(rt.price) + 1e-300
    if not ok:
        bad += 1
        print("MISMATCH", dict(dyn=dyn, n=n, steps=steps, anti=anti, P=P), rg.price, rt.price, rp.price, float((tg - tt).abs().max()), flush=True)
    del dW, pm
print(f"{N} cases, {bad} mismatches")
