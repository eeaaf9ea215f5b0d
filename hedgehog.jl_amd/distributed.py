"""Path-sharded solve across the GPUs of one node: one process per GPU (`torch.distributed`,
backend "nccl" = RCCL over xGMI), trajectories split into contiguous ranges, ONE all-reduce of the
16-double accumulator vector for the final estimator.  No data-path collective:
per-trajectory Philox keys (Euler) and global-index counters (exact laws) make every draw
independent of the sharding, so N-GPU and 1-GPU results differ only by the order of the final sum.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from .domain import MonteCarloSolution
from .montecarlo import MonteCarlo, _model_and_config, _price_from


def shard_range(n_paths: int, rank: int, world: int):
    """[start, stop) of `rank`: contiguous ranges of ceil(N/G) trajectories, the last ones shorter
    (possibly empty)."""
    per = -(-n_paths // world)
    start = min(n_paths, rank * per)
    return start, min(n_paths, start + per)


def _hip_accumulate(model, cfg, device):
    """Default accumulate: HIP kernels, accumulators left in HBM as a torch tensor."""
    import torch
    ctx = _ffi.get_context(device)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    acc = torch.zeros(_ffi.HH_ACC_LEN, dtype=torch.float64, device=torch.device("cuda", device))
    if cfg.n_paths > 0:
        ctx.check(ctx.lib.hh_mc_accumulate(ctx.handle, C.byref(model), C.byref(cfg), acc.data_ptr(),
                                           None))
    return acc


def solve_sharded(prob, method: MonteCarlo, group=None, accumulate=None) -> MonteCarloSolution:
    """solve(prob, method) with the trajectories of `method.config` sharded over the ranks of
    `group` (default: the world).  Every rank returns the same MonteCarloSolution (ensemble=None).

    `accumulate(model, cfg, device) -> tensor[HH_ACC_LEN]` is the per-shard kernel driver; the
    default runs the HIP path (and raises without a GPU).  Tests inject a CPU checker here to
    exercise the sharding + collective logic under gloo."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    cfg0 = method.config
    start, stop = shard_range(cfg0.trajectories, rank, world)
    model, c, keep, P, discount = _model_and_config(prob, method, n_paths=stop - start,
                                                    path_offset=start)
    seeds = cfg0.seeds
    if c.strategy == _ffi.HH_EULER_MARUYAMA:
        seeds = np.ascontiguousarray(seeds[start:stop] if stop > start else seeds[:1])
    c.seeds = seeds.ctypes.data
    c.seeds_len = seeds.size
    acc = (accumulate or _hip_accumulate)(model, c, method.device)
    if not isinstance(acc, torch.Tensor):
        acc = torch.as_tensor(np.asarray(acc, dtype=np.float64))
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)  # the path's one exchange
    acc_host = np.ascontiguousarray(acc.detach().cpu().numpy())
    res = _ffi.hh_result()
    lib = _ffi.load_library()
    rc = lib.hh_mc_finalize(C.byref(model), C.byref(c), acc_host.ctypes.data, C.byref(res))
    if rc != 0:
        raise _ffi.HedgehogMCError(rc, "hh_mc_finalize failed")
    del keep, seeds
    return MonteCarloSolution(prob, method, _price_from(res, discount, P), None,
                              std_error=res.std_error, result=res)
