"""Soak: path-major REPLAY == tile-major REPLAY bit for bit over random shapes, row phases and kernel variants
(GPU box: python tools/soak_path_major.py [seed] [cases]; 1500 cases take 5 s)."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hedgehog_jl_amd import _ffi
from tests import oracle_ffi as o
from tests.test_gpu_parity import gpu_solve, HESTON_SEEDS
ctx = _ffi.get_context(0)
GBM, HES, EM, REP = _ffi.HH_LOGNORMAL, _ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, _ffi.HH_NOISE_REPLAY
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
bad = 0
for it in range(N):
    dyn = HES if rng.random() < 0.6 else GBM
    nc = 2 if dyn == HES else 1
    n_paths = int(rng.choice([rng.integers(1, 70), rng.integers(1, 700), rng.integers(1, 9000)]))
    n_steps = int(rng.choice([rng.integers(1, 20), rng.integers(1, 130), rng.integers(1, 400)]))
    anti = int(rng.random() < 0.3); duals = int(rng.choice([0, 0, 1, 3])); split = int(rng.random() < 0.8)
    shift = int(rng.integers(0, 8))
    sd = {0: None, 1: {"V0": [1.0]} if dyn == HES else {"sigma": [1.0]},
          3: HESTON_SEEDS if dyn == HES else {"S0": [1, 0, 0], "sigma": [0, 1, 0], "r_drift": [0, 0, 1], "discount": [0, 0, -float(np.exp(-0.03))]}}[duals]
    m = o.make_model(sigma=0.2 if dyn == GBM else 0.3, seeds=sd, n_partials=duals)
    pm = rng.standard_normal((n_paths, n_steps, nc)) * (1.0 / np.sqrt(n_steps))
    ntile = (n_paths + 255) // 256
    padded = np.zeros((ntile * 256, n_steps, nc)); padded[:n_paths] = pm
    tiled = np.ascontiguousarray(padded.reshape(ntile, 256, n_steps, nc).transpose(0, 2, 3, 1)).ravel()
    kw = dict(antithetic=anti, em_split=split, noise_mode=REP, n_partials=duals)
    c2 = o.make_config(dyn, EM, n_paths, n_steps, replay=tiled, **kw)
    r2, t2 = gpu_solve(ctx, m, c2)
    if (n_steps * nc) % 2 == 0 or shift == 0:
        # device-resident rows starting `shift` 16-byte pieces into a line (odd row lengths only from an aligned base: they repack)
        buf = torch.full((pm.size + 2 * shift + 64,), float("nan"), dtype=torch.float64, device="cuda")
        buf[2 * shift:2 * shift + pm.size] = torch.from_numpy(pm.ravel()).cuda()
        c1 = o.make_config(dyn, EM, n_paths, n_steps, replay_layout=_ffi.HH_REPLAY_PATH_MAJOR, **kw)
        c1.replay, c1.replay_on_device, c1.replay_len = buf.data_ptr() + 16 * shift, 1, pm.size
    else:
        c1 = o.make_config(dyn, EM, n_paths, n_steps, replay=np.ascontiguousarray(pm), replay_layout=_ffi.HH_REPLAY_PATH_MAJOR, **kw)
    r1, t1 = gpu_solve(ctx, m, c1)
    ok = np.array_equal(t1, t2) and r1.price == r2.price and all(r1.dprice[k] == r2.dprice[k] for k in range(duals))
    if not ok:
        bad += 1
        print("MISMATCH", dict(dyn=dyn, n_paths=n_paths, n_steps=n_steps, anti=anti, duals=duals, split=split, shift=shift), flush=True)
print(f"{N} cases, {bad} mismatches")
