#!/usr/bin/env python3
"""When each tile of the Broadie–Kaya CF kernel ran: start / end stamps (s_memrealtime, 10 ns) left by the diagnostic
build -DHH_BK_TILE_STAMPS=1 (tools/build_variant.py stamps -DHH_BK_TILE_STAMPS=1) in place of three series lengths per
tile.  Prints the chain's timeline: tile durations by the time they started, how many tiles run at a time, the drain.
GPU box only.  usage: bk_tile_timeline.py [n_paths [variant-tag]]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from hedgehog_jl_amd import _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tag = sys.argv[2] if len(sys.argv) > 2 else "stamps"
lib = C.CDLL(os.path.join(ROOT, "hedgehog.jl_amd", "lib", "variants", f"libhh_bk_{tag}.so"))
for name, res, args in _ffi.SYMBOLS:
    if hasattr(lib, name):
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
h = C.c_void_p()
assert lib.hh_ctx_create(C.byref(h), 0) == 0
lib.hh_ctx_enable_timing(h, 1)
seed0 = torch.tensor([99], dtype=torch.int64, device="cuda")
acc = torch.zeros(16, dtype=torch.float64, device="cuda")
m = _ffi.make_model()
c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n, 1)
c.seeds, c.seeds_on_device = seed0.data_ptr(), 1
for _ in range(8):  # warm: clocks, caches, the context's scratch
    assert lib.hh_mc_accumulate(h, C.byref(m), C.byref(c), acc.data_ptr(), None) == 0
lib.hh_ctx_synchronize(h)
buf, k = (C.c_double * 256)(), C.c_int32()
lib.hh_ctx_read_timings(h, buf, 256, C.byref(k))
print(f"n = {n}: chain by HIP events {np.median([buf[i] for i in range(k.value)]) * 1e3:.1f} us (median of {k.value})")
dec, ln = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
assert lib.hh_bk_decisions(h, n, dec.ctypes.data, ln.ctypes.data) == 0
tiles = n // 256
st = ln[:tiles * 256].reshape(tiles, 256)[:, 253:].astype(np.int64)
start, end, where = st[:, 0], st[:, 1], st[:, 2]
t0 = start.min()
start, end = (start - t0) / 100.0, (end - t0) / 100.0  # us
dur = end - start
print(f"{tiles} whole tiles; first start 0, last start {start.max():.1f} us, last end {end.max():.1f} us")
order = np.argsort(start)
print("tiles in the order they started, by groups of 640 (half a round of 1280 workgroups):")
for g in range(0, tiles, 640):
    i = order[g:g + 640]
    print(f"  tiles {g:5d}..{g + len(i) - 1:5d}: start {start[i].min():6.1f} .. {start[i].max():6.1f} us   duration mean {dur[i].mean():6.1f}  "
          f"min {dur[i].min():6.1f}  max {dur[i].max():6.1f}   end {end[i].min():6.1f} .. {end[i].max():6.1f}")
print("tiles running at time t:")
for t in np.arange(0.0, end.max() + 10.0, 10.0):
    print(f"  t = {t:6.1f} us: {int(((start <= t) & (end > t)).sum()):5d}")
xcc = where >> 16
for x in np.unique(xcc):
    i = xcc == x
    print(f"  XCD {x}: {int(i.sum()):5d} tiles, last end {end[i].max():6.1f} us, mean duration {dur[i].mean():6.1f}")
cu = where  # XCD | HW_ID
ids, cnt = np.unique(cu & 0xffffff00, return_counts=True)  # without the wave/simd bits
print(f"{len(ids)} distinct (XCD, SE, SH, CU, pipe) places; tiles per place min {cnt.min()} max {cnt.max()}")
# does tile index order = start order?
print(f"start order against tile index: {np.mean(np.abs(np.argsort(order) - np.arange(tiles))):.1f} places apart on average")
