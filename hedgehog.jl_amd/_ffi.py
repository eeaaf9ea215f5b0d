"""ctypes binding of libhedgehog_mc.so (include/hedgehog_mc.h).

The product path has no CPU fallback: if the HIP library is missing or no HIP device is present,
every compute call raises `HedgehogMCError`.
"""
from __future__ import annotations

import ctypes as C
import os

HH_MAX_PARTIALS = 8
HH_LSM_PHASE_POW, HH_LSM_PHASE_INIT, HH_LSM_PHASE_STEP = 1, 2, 3
HH_TILE_PATHS = 256
HH_ACC_LEN = 16
HH_MAX_MODELS = 16
HH_ACC_SUM, HH_ACC_SUMSQ, HH_ACC_DSUM, HH_ACC_NPATHS = 0, 1, 2, 10

HH_LOGNORMAL, HH_HESTON = 0, 1
HH_EULER_MARUYAMA, HH_EXACT_LAW, HH_BROADIE_KAYA = 0, 1, 2
HH_NOISE_GENERATE, HH_NOISE_REPLAY = 0, 1
HH_REPLAY_TILE_MAJOR, HH_REPLAY_PATH_MAJOR = 0, 1

HH_OK, HH_ERR_INVALID, HH_ERR_UNSUPPORTED, HH_ERR_HIP, HH_ERR_NOMEM, HH_ERR_RCCL, HH_ERR_DEVICE_TIMEOUT = 0, -1, -2, -3, -4, -5, -6
HH_MGPU_AUTO, HH_MGPU_HOST_SUM, HH_MGPU_RCCL = 0, 1, 2
HH_MGPU_REDUCE_HOST, HH_MGPU_REDUCE_RCCL = 0, 1
HH_MGPU_OPT_ENQUEUE = 1
HH_MGPU_ENQUEUE_SERIAL, HH_MGPU_ENQUEUE_THREADS = 0, 1

_dp = C.POINTER(C.c_double)


class hh_model(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("S0", "V0", "kappa", "theta", "sigma", "rho", "r_drift", "discount", "T", "strike",
                 "cp")] + \
               [(n, _dp) for n in
                ("dS0", "dV0", "dkappa", "dtheta", "dsigma", "dr_drift", "ddiscount", "dstrike")]


class hh_config(C.Structure):
    _fields_ = [
        ("dynamics", C.c_int32), ("strategy", C.c_int32), ("antithetic", C.c_int32),
        ("em_split", C.c_int32), ("compat_sqrt_alpha", C.c_int32), ("noise_mode", C.c_int32),
        ("replay_layout", C.c_int32), ("seeds_on_device", C.c_int32),
        ("replay_on_device", C.c_int32), ("terminal_on_device", C.c_int32),
        ("n_steps", C.c_uint32), ("n_partials", C.c_uint32),
        ("n_paths", C.c_uint64), ("path_offset", C.c_uint64),
        ("seeds", C.c_void_p), ("replay", C.c_void_p),
        ("bk_n_sigma", C.c_double), ("bk_cf_tol", C.c_double), ("bk_atol", C.c_double),
        ("bk_moment_h", C.c_double),
        ("bk_newton_maxiter", C.c_int32), ("bk_bisect_maxiter", C.c_int32),
        ("bk_root_form", C.c_int32), ("bk_bracket_form", C.c_int32), ("bk_caps", C.c_int32), ("reserved0", C.c_int32),
        ("seeds_len", C.c_uint64), ("replay_len", C.c_uint64),
    ]


class hh_result(C.Structure):
    _fields_ = [
        ("price", C.c_double), ("std_error", C.c_double),
        ("sum_payoff", C.c_double), ("sumsq_payoff", C.c_double),
        ("dprice", C.c_double * HH_MAX_PARTIALS),
        ("n_paths_done", C.c_uint64),
        ("bk_newton_fail", C.c_uint64), ("bk_bisect_fallback", C.c_uint64),
        ("bk_maxguess_fallback", C.c_uint64), ("bk_cf_terms", C.c_uint64),
        ("kernel_ms", C.c_double), ("total_ms", C.c_double),
    ]


class hh_lsm_result(C.Structure):
    _fields_ = [("price", C.c_double), ("std_error", C.c_double), ("n_paths_total", C.c_uint64),
                ("rows_regressed", C.c_uint32), ("rows_skipped", C.c_uint32),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("form", C.c_int32), ("persistent_fallbacks", C.c_int32)]


HH_BK_ROOT_SECANT, HH_BK_ROOT_ORDER2 = 0, 1
HH_BK_BRACKET_MIDPOINT, HH_BK_BRACKET_ROOTS = 0, 1
HH_BK_CAPS_AS_WRITTEN, HH_BK_CAPS_ROOTS_DEFAULT = 0, 1
HH_OPT_LSM_FORM = 1
HH_OPT_BK_TERM_CACHE = 2
HH_OPT_GRID_FORM = 3
HH_OPT_LSM_SPIN_TICKS = 4
HH_OPT_FUSE_REDUCE = 5
HH_OPT_GRID_ORDER = 6
HH_OPT_FINISH_SPIN_TICKS = 7
HH_OPT_FINISH_TILE_FIRST = 8
HH_GRID_FORM_PER_DATE, HH_GRID_FORM_BATCHED = 0, 1
HH_CM_GRAD_LEN = 8  # enum hh_cm_grad: S0, V0, kappa, theta, sigma, rho, r_drift, discount
HH_LSM_FORM_PER_DATE, HH_LSM_FORM_PERSISTENT, HH_LSM_FORM_AUTO = 0, 1, 2


class HedgehogMCError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hedgehog_mc error {code}: {msg}")
        self.code = code


LIB_PATH = os.environ.get("HEDGEHOG_MC_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "lib", "libhedgehog_mc.so")  # env: A/B builds

# every symbol include/hedgehog_mc.h declares: (name, restype, argtypes)
_vp = C.c_void_p
SYMBOLS = [
    ("hh_abi_version", C.c_int, []),
    ("hh_ctx_create", C.c_int, [C.POINTER(_vp), C.c_int]),
    ("hh_ctx_destroy", None, [_vp]),
    ("hh_ctx_set_stream", C.c_int, [_vp, _vp]),
    ("hh_ctx_reset_stream", C.c_int, [_vp]),
    ("hh_last_error", C.c_char_p, [_vp]),
    ("hh_ctx_set_option", C.c_int, [_vp, C.c_int32, C.c_int64]),
    ("hh_ctx_check_last", C.c_int, [_vp]),
    ("hh_mc_solve", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), C.POINTER(hh_result), _vp]),
    ("hh_mc_accumulate", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), _vp, _vp]),
    ("hh_mc_solve_multi", C.c_int, [_vp, C.POINTER(hh_model), C.c_uint32, C.POINTER(hh_config), C.POINTER(hh_result), _vp]),
    ("hh_mc_accumulate_multi", C.c_int, [_vp, C.POINTER(hh_model), C.c_uint32, C.POINTER(hh_config), _vp, _vp]),
    ("hh_mc_finalize", C.c_int, [C.POINTER(hh_model), C.POINTER(hh_config), _vp, C.POINTER(hh_result)]),
    ("hh_mc_accumulate_basket", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), _vp, _vp, C.c_uint32, _vp, _vp]),
    ("hh_mc_solve_basket", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), _vp, _vp, C.c_uint32, _vp, _vp]),
    ("hh_carr_madan", C.c_int, [_vp, C.POINTER(hh_model), C.c_int32, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    ("hh_carr_madan_basket", C.c_int, [_vp, C.POINTER(hh_model), C.c_int32, C.c_int32, C.c_double, C.c_double,
                                       _vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp]),
    ("hh_carr_madan_basket_grad", C.c_int, [_vp, C.POINTER(hh_model), C.c_int32, C.c_int32, C.c_double,
                                            C.c_double, _vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp]),
    ("hh_lsm_grid_elems", C.c_size_t, [C.c_uint64, C.c_uint32, C.c_int32]),
    ("hh_lsm_solve", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), C.c_int32, C.c_double, C.POINTER(hh_lsm_result), _vp, _vp, _vp]),
    ("hh_lsm_solve_grid", C.c_int, [_vp, C.POINTER(hh_model), _vp, C.c_uint64, C.c_uint32, C.c_int32, C.c_double, C.POINTER(hh_lsm_result), _vp, _vp]),
    ("hh_heston_exact_grid", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), _vp, _vp, C.c_int32, C.POINTER(hh_result)]),
    ("hh_lsm_shard_xchg_elems", C.c_size_t, [C.c_uint32, C.c_int32]),
    ("hh_lsm_shard_begin", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), C.c_int32, C.c_double, _vp]),
    ("hh_lsm_shard_phase", C.c_int, [_vp, C.c_int32, C.c_uint32, _vp, _vp]),
    ("hh_lsm_shard_finish", C.c_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("hh_lsm_finalize", C.c_int, [_vp, C.POINTER(hh_lsm_result)]),
    ("hh_lsm_debug_read", C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_int32, _vp]),
    ("hh_replay_elems", C.c_size_t, [C.c_uint64, C.c_uint32, C.c_int32]),
    ("hh_replay_pack", C.c_int, [_vp, C.c_int32, C.c_uint64, C.c_uint32, _vp, C.c_int32, _vp]),
    ("hh_wiener_fill", C.c_int, [_vp, C.c_int32, C.c_double, C.c_double, C.c_uint32, C.c_uint64, _vp, C.c_int32, _vp]),
    ("hh_seeds_fingerprint", C.c_uint64, [_vp, C.c_uint64]),
    ("hh_seeds_cache", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.POINTER(_vp)]),
    ("hh_seeds_cache_stats", C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("hh_device_malloc", C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    ("hh_device_free", C.c_int, [_vp, _vp]),
    ("hh_memcpy_h2d", C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    ("hh_memcpy_d2h", C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    ("hh_ctx_synchronize", C.c_int, [_vp]),
    ("hh_ctx_enable_timing", C.c_int, [_vp, C.c_int32]),
    ("hh_ctx_read_timings", C.c_int, [_vp, _vp, C.c_int32, C.POINTER(C.c_int32)]),
    ("hh_bk_decisions", C.c_int, [_vp, C.c_uint64, _vp, _vp]),
    ("hh_mgpu_create", C.c_int, [C.POINTER(_vp), C.POINTER(C.c_int), C.c_int, C.c_int]),
    ("hh_mgpu_destroy", None, [_vp]),
    ("hh_mgpu_last_error", C.c_char_p, [_vp]),
    ("hh_mgpu_n_devices", C.c_int, [_vp]),
    ("hh_mgpu_reduce_mode", C.c_int, [_vp]),
    ("hh_mgpu_ctx", _vp, [_vp, C.c_int]),
    ("hh_mgpu_selftest", C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("hh_mgpu_rccl_info", C.c_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("hh_mgpu_set_option", C.c_int, [_vp, C.c_int32, C.c_int64]),
    ("hh_mgpu_enqueue_stats", C.c_int, [_vp, _vp, C.POINTER(C.c_double)]),
    ("hh_mgpu_shard_range", None, [C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("hh_mgpu_solve", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), C.POINTER(hh_result), _vp]),
    ("hh_mgpu_solve_shards", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), C.POINTER(hh_result), _vp]),
    ("hh_mgpu_solve_basket", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), _vp, _vp, C.c_uint32, _vp]),
    ("hh_mgpu_solve_multi", C.c_int, [_vp, C.POINTER(hh_model), C.c_uint32, C.POINTER(hh_config), C.POINTER(hh_result)]),
    ("hh_mgpu_lsm_solve", C.c_int, [_vp, C.POINTER(hh_model), C.POINTER(hh_config), C.c_int32, C.c_double,
                                    C.POINTER(hh_lsm_result), _vp, _vp]),
]

_lib = None


def load_library(path: str | None = None):
    """Load the C-ABI library (once). Raises HedgehogMCError when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise HedgehogMCError(HH_ERR_HIP, f"{path} not found — build it with "
                              "`python -c 'import __graft_entry__ as g; g.build()'`; "
                              "there is no CPU fallback")
    if not os.environ.get("HEDGEHOG_MC_NO_TORCH"):
        try:  # share PyTorch's HIP runtime (same libamdhip64 soname) when torch is the allocator
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the C-ABI itself
            pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Context:
    """Owns one hh_ctx (one device, one stream)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.hh_ctx_create(C.byref(h), int(device))
        if rc != HH_OK:
            raise HedgehogMCError(rc, f"hh_ctx_create(device={device}) failed — no HIP device? "
                                  "(the product path has no CPU fallback)")
        self.handle = h
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    def check(self, rc: int):
        if rc != HH_OK:
            raise HedgehogMCError(rc, self.lib.hh_last_error(self.handle).decode(errors="replace"))

    def set_stream(self, stream: int | None):
        """Launch on an external hipStream_t handle (0 = the default stream); None = own stream."""
        if stream is None:
            self.check(self.lib.hh_ctx_reset_stream(self.handle))
        else:
            self.check(self.lib.hh_ctx_set_stream(self.handle, _vp(stream)))

    def set_option(self, option: int, value: int):
        self.check(self.lib.hh_ctx_set_option(self.handle, int(option), int(value)))

    def synchronize(self):
        self.check(self.lib.hh_ctx_synchronize(self.handle))

    def check_last(self):
        """After asynchronous solves (accumulate): waits for the stream; raises HH_ERR_DEVICE_TIMEOUT when a record
        reduction inside a simulation kernel gave up since the last check (the context has been reset by then)."""
        self.check(self.lib.hh_ctx_check_last(self.handle))

    def seeds_on_device(self, seeds, fingerprint: int = 0) -> int:
        """Device address of this seed vector in the context's content-addressed cache (hh_seeds_cache): valid
        until a later call misses — ask right before every solve."""
        p = _vp()
        self.check(self.lib.hh_seeds_cache(self.handle, _vp(seeds.ctypes.data), seeds.size, int(fingerprint), C.byref(p)))
        return p.value

    def seeds_cache_stats(self):
        h, u, e = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.check(self.lib.hh_seeds_cache_stats(self.handle, C.byref(h), C.byref(u), C.byref(e)))
        return {"hits": h.value, "uploads": u.value, "evictions": e.value}

    def enable_timing(self, on: bool = True):
        self.check(self.lib.hh_ctx_enable_timing(self.handle, int(on)))

    def read_timings(self):
        """ms of each simulation-kernel launch since the last read (syncs the stream)."""
        buf = (C.c_double * 256)()
        n = C.c_int32(0)
        self.check(self.lib.hh_ctx_read_timings(self.handle, buf, 256, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    # a few recycled device buffers by exact size (the sample buffers of repeated solves)
    _POOL_MAX = 4

    def pool_take(self, nbytes: int):
        pool = self.__dict__.setdefault("_pool", [])
        for i, (p, n) in enumerate(pool):
            if n == nbytes:
                del pool[i]
                return p
        return None

    def pool_put(self, ptr: int, nbytes: int) -> bool:
        pool = self.__dict__.setdefault("_pool", [])
        if len(pool) >= self._POOL_MAX:
            old_p, _ = pool.pop(0)
            self.lib.hh_device_free(self.handle, _vp(old_p))
        pool.append((ptr, nbytes))
        return True

    def close(self):
        if getattr(self, "handle", None):
            for p, _ in self.__dict__.get("_pool", []):
                self.lib.hh_device_free(self.handle, _vp(p))
            self.__dict__["_pool"] = []
            self.lib.hh_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """Device memory of a Context (hh_device_malloc / hh_device_free), released with the object."""

    def __init__(self, ctx: Context, nbytes: int, pooled: bool = False):
        self.ctx, self.nbytes, self._pooled = ctx, int(nbytes), pooled
        self.ptr = ctx.pool_take(self.nbytes) if pooled else None
        if self.ptr is None:
            p = _vp()
            ctx.check(ctx.lib.hh_device_malloc(ctx.handle, max(self.nbytes, 8), C.byref(p)))
            self.ptr = p.value

    def upload(self, arr):
        assert arr.flags.c_contiguous and arr.nbytes <= self.nbytes
        self.ctx.check(self.ctx.lib.hh_memcpy_h2d(self.ctx.handle, _vp(self.ptr), _vp(arr.ctypes.data), arr.nbytes))
        return self

    def download(self, arr):
        assert arr.flags.c_contiguous and arr.nbytes <= self.nbytes
        self.ctx.check(self.ctx.lib.hh_memcpy_d2h(self.ctx.handle, _vp(arr.ctypes.data), _vp(self.ptr), arr.nbytes))
        return arr

    def free(self):
        if getattr(self, "ptr", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.hh_device_free(self.ctx.handle, _vp(self.ptr))
        self.ptr = None

    def recycle(self):
        """Hand the memory back to the context's pool (hipMalloc / hipFree cost 0.1-0.3 ms each and
        synchronise: a solve that allocates its sample buffer afresh pays more than the download)."""
        if getattr(self, "ptr", None) and getattr(self.ctx, "handle", None) and self.ctx.pool_put(self.ptr, self.nbytes):
            self.ptr = None
        else:
            self.free()

    def __del__(self):  # pragma: no cover
        try:
            self.recycle() if getattr(self, "_pooled", False) else self.free()
        except Exception:
            pass


class BorrowedContext(Context):
    """The hh_ctx of one device of a MultiGpu (hh_mgpu_ctx): same methods, never destroyed here."""

    def __init__(self, lib, handle, device):
        self.lib, self.handle, self.device = lib, handle, int(device)

    def close(self):
        for p, _ in self.__dict__.get("_pool", []):
            self.lib.hh_device_free(self.handle, _vp(p))
        self.__dict__["_pool"] = []
        self.handle = None


class MultiGpu:
    """Owns one hh_mgpu: one hh_ctx per listed device, driven from this host thread; the accumulator
    vectors are combined inside the library (RCCL all-reduce, or the host's ordered sum)."""

    def __init__(self, devices, flags: int = HH_MGPU_AUTO):
        self.lib = load_library()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = _vp()
        rc = self.lib.hh_mgpu_create(C.byref(h), arr, len(self.devices), int(flags))
        if rc != HH_OK:
            raise HedgehogMCError(rc, f"hh_mgpu_create(devices={self.devices}, flags={flags}) failed "
                                  "(no HIP device, a bad ordinal, or RCCL required but unavailable)")
        self.handle = h
        self._ctxs = [BorrowedContext(self.lib, _vp(self.lib.hh_mgpu_ctx(h, i)), d)
                      for i, d in enumerate(self.devices)]

    @property
    def n_devices(self) -> int:
        return self.lib.hh_mgpu_n_devices(self.handle)

    @property
    def reduce_mode(self) -> int:
        return self.lib.hh_mgpu_reduce_mode(self.handle)

    def last_error(self) -> str:
        return self.lib.hh_mgpu_last_error(self.handle).decode(errors="replace")

    def selftest(self):
        """(ranks counted by an all-reduce of ones through the solve's own exchange, reduce mode it ran in)"""
        n, mode = C.c_int32(0), C.c_int32(0)
        self.check(self.lib.hh_mgpu_selftest(self.handle, C.byref(n), C.byref(mode)))
        return n.value, mode.value

    def rccl_info(self):
        """{"library": path the collective entry points were bound from ("" = none), "version", "from_env", "usable"}"""
        buf = C.create_string_buffer(512)
        ver, env = C.c_int32(0), C.c_int32(0)
        rc = self.lib.hh_mgpu_rccl_info(self.handle, buf, 512, C.byref(ver), C.byref(env))
        return {"library": buf.value.decode(errors="replace"), "version": ver.value, "from_env": bool(env.value),
                "usable": rc == HH_OK}

    def set_option(self, option: int, value: int):
        self.check(self.lib.hh_mgpu_set_option(self.handle, int(option), int(value)))

    def enqueue_stats(self):
        """(per-shard host µs, whole-phase host µs) of the last solve's enqueue phase."""
        per = (C.c_double * len(self.devices))()
        whole = C.c_double(0.0)
        self.check(self.lib.hh_mgpu_enqueue_stats(self.handle, per, C.byref(whole)))
        return list(per), whole.value

    def ctx(self, i: int) -> Context:
        return self._ctxs[i]

    def check(self, rc: int):
        if rc != HH_OK:
            raise HedgehogMCError(rc, self.last_error())

    def shard_range(self, n_paths: int, g: int, tile_aligned: bool = False):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.lib.hh_mgpu_shard_range(int(n_paths), len(self.devices), int(g), int(tile_aligned),
                                     C.byref(a), C.byref(b))
        return a.value, b.value

    def solve(self, model, cfg, terminal=None) -> hh_result:
        res = hh_result()
        self.check(self.lib.hh_mgpu_solve(self.handle, C.byref(model), C.byref(cfg), C.byref(res),
                                          terminal.ctypes.data if terminal is not None else None))
        return res

    def solve_multi(self, models, cfg):
        arr = (hh_model * len(models))(*models)
        res = (hh_result * len(models))()
        self.check(self.lib.hh_mgpu_solve_multi(self.handle, arr, len(models), C.byref(cfg), res))
        return list(res)

    def solve_shards(self, model, cfgs, terminals=None) -> hh_result:
        arr = (hh_config * len(cfgs))(*cfgs)
        tp = None
        if terminals is not None:
            tp = (_vp * len(cfgs))(*[(t if isinstance(t, int) or t is None else t.ctypes.data) for t in terminals])
        res = hh_result()
        self.check(self.lib.hh_mgpu_solve_shards(self.handle, C.byref(model), arr, C.byref(res), tp))
        return res

    def close(self):
        if getattr(self, "handle", None):
            for c in self._ctxs:
                c.close()
            self.lib.hh_mgpu_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_contexts: dict[int, Context] = {}
_mgpus: dict[tuple, MultiGpu] = {}


def get_multi_gpu(devices, flags: int = HH_MGPU_AUTO) -> MultiGpu:
    key = (tuple(int(d) for d in devices), int(flags))
    mg = _mgpus.get(key)
    if mg is None:
        mg = _mgpus[key] = MultiGpu(key[0], flags)
    return mg


def get_context(device: int = 0) -> Context:
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = Context(device)
    return ctx


def seed_array(values, n: int):
    """P-long seed vector -> ctypes double array (or NULL when all zero)."""
    if values is None or not any(values):
        return None
    arr = (C.c_double * n)(*[float(v) for v in values])
    return arr


# ---- plain-number builders of the C structs (bench, tools, tests drive the C-ABI directly) ----------

def make_model(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1.0,
               strike=100.0, cp=1.0, discount=None, seeds=None, n_partials=0):
    """hh_model from plain numbers; defaults = benchmark problem H252.
    seeds: dict name -> list of n_partials dual seeds (names: S0, V0, kappa, theta, sigma, r_drift,
    discount, strike)."""
    import math
    m = hh_model()
    m.S0, m.V0, m.kappa, m.theta, m.sigma, m.rho = S0, V0, kappa, theta, sigma, rho
    m.r_drift, m.T, m.strike, m.cp = r, T, strike, cp
    m.discount = math.exp(-r * T) if discount is None else discount
    keep = []
    for name, vals in (seeds or {}).items():
        arr = (C.c_double * n_partials)(*vals)
        keep.append(arr)
        setattr(m, "d" + name, C.cast(arr, C.POINTER(C.c_double)))
    m._keep = keep
    return m


def make_config(dynamics, strategy, n_paths, n_steps=1, antithetic=0, em_split=1, noise_mode=0,
                seeds=None, replay=None, replay_layout=0, n_partials=0, path_offset=0,
                compat_sqrt_alpha=0):
    """hh_config from plain numbers; `seeds` / `replay` are HOST numpy arrays (kept alive on the
    returned struct) — set the pointer fields and the *_on_device flags yourself for device memory."""
    import numpy as np
    c = hh_config()
    c.dynamics, c.strategy, c.antithetic, c.em_split = dynamics, strategy, antithetic, em_split
    c.compat_sqrt_alpha = compat_sqrt_alpha
    c.noise_mode, c.replay_layout = noise_mode, replay_layout
    c.n_steps, c.n_partials, c.n_paths, c.path_offset = n_steps, n_partials, n_paths, path_offset
    keep = []
    if seeds is not None:
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        keep.append(seeds)
        c.seeds = seeds.ctypes.data
        c.seeds_len = seeds.size
    if replay is not None:
        replay = np.ascontiguousarray(replay, dtype=np.float64)
        keep.append(replay)
        c.replay = replay.ctypes.data
        c.replay_len = replay.size
    c._keep = keep
    return c
