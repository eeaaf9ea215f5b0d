import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import hedgehog_jl_amd as hh
from hedgehog_jl_amd import _ffi
ctx = hh.get_context(0)
m = _ffi.make_model()
cm = 9.242521073959068
def run(seed, n=1_000_000, **kw):
    c = _ffi.make_config(1, 2, n, seeds=[seed])
    for k, v in kw.items(): setattr(c, k, v)
    r = _ffi.hh_result()
    ctx.check(ctx.lib.hh_mc_solve(ctx.handle, C.byref(m), C.byref(c), C.byref(r), None))
    return r
for label, kw in (("reference defaults", {}), ("cf_tol=1e-6", dict(bk_cf_tol=1e-6)), ("atol=1e-7, newton 30", dict(bk_atol=1e-7, bk_newton_maxiter=30)), ("n_sigma=10", dict(bk_n_sigma=10.0)), ("all tight", dict(bk_cf_tol=1e-7, bk_atol=1e-8, bk_newton_maxiter=40, bk_n_sigma=12.0))):
    z = []; ms = []
    for seed in range(1, 13):
        r = run(seed, **kw); z.append((r.price - cm) / r.std_error); ms.append(r.kernel_ms)
    z = np.array(z)
    print(f"{label:24s} mean z = {z.mean():+.2f} (se of mean {1/np.sqrt(len(z)):.2f})  z range [{z.min():+.2f}, {z.max():+.2f}]  kernel {np.median(ms):.1f} ms  cf_terms/path {r.bk_cf_terms/1e6:.1f} newton_fail {r.bk_newton_fail}")
