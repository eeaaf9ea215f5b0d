"""A plain-C client of the C-ABI (tests/c/cabi_smoke.c), built with gcc against
include/hedgehog_mc.h and run with NO Python/torch in the process — what a foreign host such as
Julia's ccall sees."""
import os
import subprocess

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_client(tmp_path):
    libdir = os.path.join(ROOT, "hedgehog.jl_amd", "lib")
    exe = str(tmp_path / "cabi_smoke")
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-Werror",
                    os.path.join(ROOT, "tests", "c", "cabi_smoke.c"), "-o", exe,
                    "-L" + libdir, "-lhedgehog_mc", "-lm",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    tag, price, se = p.stdout.strip().splitlines()[-1].split()  # RCCL may print its banner before
    assert tag == "OK" and 8.5 < float(price) < 10.0 and 0 < float(se) < 0.1


def test_host_operands_of_an_asynchronous_call_are_free_when_it_returns(hhlib):
    """hedgehog_mc.h: "the caller owns every buffer it passes, for the duration of the call only" — also for
    hh_mc_accumulate, which does not wait for its kernels.  PINNED host memory is read by the copy engine only
    when the stream gets there, so the call has to wait for its staging copies: seeds and increments in pinned
    buffers are overwritten the moment the call returns, and the sums must be those of the original contents."""
    import ctypes as C

    import numpy as np
    import torch

    from hedgehog_jl_amd import _ffi
    from tests import oracle_ffi as o
    ctx = hhlib
    n, steps = 1_000_000, 4
    seeds = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    m = o.make_model()
    acc = _ffi.DeviceBuffer(ctx, 8 * _ffi.HH_ACC_LEN)

    def accumulate(c):
        ctx.check(ctx.lib.hh_mc_accumulate(ctx.handle, C.byref(m), C.byref(c), acc.ptr, None))

    want_c = o.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, n, steps, seeds=seeds)
    accumulate(want_c)
    ctx.synchronize()
    want = acc.download(np.empty(_ffi.HH_ACC_LEN))
    pinned = torch.from_numpy(seeds.view(np.int64).copy()).pin_memory()
    c = o.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, n, steps)
    c.seeds, c.seeds_len = pinned.data_ptr(), n
    for _ in range(5):  # queue work in front, so that the copy would still be waiting were it not waited for
        accumulate(want_c)
    accumulate(c)
    pinned.zero_()      # the caller's buffer is the caller's again
    ctx.synchronize()
    assert acc.download(np.empty(_ffi.HH_ACC_LEN)).tobytes() == want.tobytes()
    # the same for REPLAY increments handed over in pinned memory
    dW = np.random.default_rng(5).standard_normal(ctx.lib.hh_replay_elems(50_000, 8, _ffi.HH_HESTON)) * 0.06
    cw = o.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, 50_000, 8, noise_mode=_ffi.HH_NOISE_REPLAY, replay=dW)
    accumulate(cw)
    ctx.synchronize()
    want = acc.download(np.empty(_ffi.HH_ACC_LEN))
    pw = torch.from_numpy(dW.copy()).pin_memory()
    cp = o.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, 50_000, 8, noise_mode=_ffi.HH_NOISE_REPLAY)
    cp.replay, cp.replay_len = pw.data_ptr(), pw.numel()
    for _ in range(5):
        accumulate(want_c)
    accumulate(cp)
    pw.fill_(float("nan"))
    ctx.synchronize()
    assert acc.download(np.empty(_ffi.HH_ACC_LEN)).tobytes() == want.tobytes()
