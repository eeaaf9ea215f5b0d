#!/usr/bin/env python3
"""Throughput of hh_replay_pack (path-major dW[path][step][comp] -> tile-major) and of
hh_wiener_fill on 10^6 x 252 Heston increments, device-resident.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hedgehog_jl_amd import _ffi

ctx = _ffi.get_context(0)
lib, h = ctx.lib, ctx.handle
n, steps = 1_000_000, 252
m = _ffi.make_model()
seeds = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
src = torch.randn(n * steps * 2, dtype=torch.float64, device="cuda")
dst = torch.empty(lib.hh_replay_elems(n, steps, 1), dtype=torch.float64, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
for name, fn, nbytes in (
        ("hh_replay_pack", lambda: lib.hh_replay_pack(h, 1, n, steps, src.data_ptr(), 1, dst.data_ptr()),
         2 * 16 * n * steps),
        ("hh_wiener_fill", lambda: lib.hh_wiener_fill(h, 1, m.rho, m.T, steps, n, seeds.data_ptr(), 1,
                                                      dst.data_ptr()), 16 * n * steps)):
    ts = []
    for _ in range(8):
        ev[0].record(stream)
        ctx.check(fn())
        ev[1].record(stream)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    t = sorted(ts)[len(ts) // 2]
    print(f"{name}: {t:.3f} ms, {nbytes / t / 1e6:.0f} GB/s (read+write)" if "pack" in name else
          f"{name}: {t:.3f} ms, {nbytes / t / 1e6:.0f} GB/s written, {n * steps / t / 1e6:.1f} G path-steps/s")
