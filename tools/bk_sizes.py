#!/usr/bin/env python3
"""Broadie–Kaya chain time against the ensemble size around whole multiples of the CF kernel's resident workgroups
(1280 = 5 per CU): is the kernel's time a staircase in its number of rounds?  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hedgehog_jl_amd import _ffi

dev = torch.device("cuda", 0)
ctx = _ffi.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
acc = torch.zeros(16, dtype=torch.float64, device=dev)
seed0 = torch.tensor([99], dtype=torch.int64, device=dev)
m = _ffi.make_model()
ctx.enable_timing(True)
tiles = [int(a) for a in sys.argv[1:]] or [320, 640, 1280, 1920, 2560, 3200, 3840, 3907, 4096, 4480, 5120, 6400, 7680, 12800]
for nt in tiles:
    n = nt * 256
    c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n)
    c.seeds, c.seeds_on_device = seed0.data_ptr(), 1
    for _ in range(14):
        ctx.check(ctx.lib.hh_mc_accumulate(ctx.handle, C.byref(m), C.byref(c), acc.data_ptr(), None))
    t = np.array(ctx.read_timings())[-10:]
    print(f"{nt:6d} tiles = {nt / 1280:5.2f} rounds of 1280 workgroups, {n:9d} trajectories: chain {np.median(t):.4f} ms "
          f"= {np.median(t) / n * 1e6:.4f} ms per 10^6")
