"""Path-sharded solve across the GPUs of one node: one process per GPU (`torch.distributed`,
backend "nccl" = RCCL over xGMI), trajectories split into contiguous ranges, ONE all-reduce of the
16-double accumulator vector for the final estimator.  No data-path collective:
per-trajectory Philox keys (Euler) and global-index counters (exact laws) make every draw
independent of the sharding, so N-GPU and 1-GPU results differ only by the order of the final sum.
"""
from __future__ import annotations

import ctypes as C
import os

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


_checked_groups: set = set()


def rank_device(default: int, group=None, device=None) -> int:
    """HIP device ordinal this rank computes on.  An explicit `device` wins; inside an initialised
    process group of more than one rank it is LOCAL_RANK (what torch.distributed.run exports), else
    torch's current device; a single process keeps `default` (MonteCarlo.device).  An RCCL ("nccl")
    group whose ranks share a device is rejected — the all-reduce would fail with a duplicate-GPU
    error or, worse, serialise every shard on one GPU (checked once per group)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(default if device is None else device)
    if device is None:
        if "LOCAL_RANK" in os.environ:
            device = int(os.environ["LOCAL_RANK"])
        elif torch.cuda.is_available():
            device = torch.cuda.current_device()
        else:
            device = default
    device = int(device)
    if dist.get_backend(group) == "nccl" and id(group) not in _checked_groups:
        import socket
        mine = (socket.gethostname(), device)
        everyone = [None] * dist.get_world_size(group)
        dist.all_gather_object(everyone, mine, group=group)
        if len(set(everyone)) != len(everyone):
            raise ValueError(f"RCCL needs one GPU per rank, but the ranks of this group sit on "
                             f"{everyone}: start one process per GPU (torch.distributed.run sets "
                             f"LOCAL_RANK) or pass device= explicitly")
        _checked_groups.add(id(group))
    return device


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


def solve_sharded(prob, method: MonteCarlo, group=None, accumulate=None,
                  device=None) -> MonteCarloSolution:
    """solve(prob, method) with the trajectories of `method.config` sharded over the ranks of
    `group` (default: the world).  Every rank returns the same MonteCarloSolution (ensemble=None).
    The rank's GPU is `device`, else LOCAL_RANK (see rank_device).

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
    acc = (accumulate or _hip_accumulate)(model, c, rank_device(method.device, group, device))
    if not isinstance(acc, torch.Tensor):
        acc = torch.as_tensor(np.asarray(acc, dtype=np.float64))
    if world > 1:  # the path's one exchange
        if acc.is_cuda:
            _all_reduce_device(acc, group)
        else:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    acc_host = np.ascontiguousarray(acc.detach().cpu().numpy())
    res = _ffi.hh_result()
    lib = _ffi.load_library()
    rc = lib.hh_mc_finalize(C.byref(model), C.byref(c), acc_host.ctypes.data, C.byref(res))
    if rc != 0:
        # NaN sums: a record reduction inside SOME rank's kernel gave up (the all-reduce spread its NaN).  The rank
        # it happened on gets the named status from its own context; every rank raises.
        if accumulate is None:
            _ffi.get_context(rank_device(method.device, group, device)).check_last()
        raise _ffi.HedgehogMCError(rc, "hh_mc_finalize failed: the summed accumulator holds no trajectories "
                                       "(another rank's solve lost its sums?)")
    del keep, seeds
    return MonteCarloSolution(prob, method, _price_from(res, discount, P), None,
                              std_error=res.std_error, result=res)


def _all_reduce_device(t, group):
    """SUM all-reduce of a device tensor: in place over RCCL ("nccl"), through host memory when the
    group's backend cannot take device tensors (gloo — the CPU-side rehearsal of the exchange)."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    t.copy_(h)


def solve_lsm_sharded(prob, method, group=None, stopping_info: bool = False, device=None):
    """solve(prob, ::LSM) (least_squares_montecarlo.jl:99-136) with the trajectories sharded over the
    ranks of `group`.  Unlike the European solve this path HAS exchange steps: the regression of
    every exercise date needs sums over all trajectories, so the induction runs in phases
    (`hh_lsm_shard_*`, include/hedgehog_mc.h) with one small SUM all-reduce between consecutive
    phases — 2 + (steps-1) + 1 collectives of at most (steps+1)·(2·degree+1) doubles.  Every rank
    returns the same price; stopping_info (if asked) is the rank's own shard."""
    import torch
    import torch.distributed as dist

    from .lsm import LSMSolution, _lsm_structs
    from .dates import MILLISECONDS_IN_YEAR_365
    from .domain import American, VanillaOption, df
    from .montecarlo import MethodError

    payoff, m = prob.payoff, prob.market_inputs
    if not (isinstance(payoff, VanillaOption) and isinstance(payoff.exercise_style, American)):
        raise MethodError("solve(::PricingProblem, ::LSM) needs an American VanillaOption")
    mc = method.mc_method
    model, c, T = _lsm_structs(prob, mc)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, stop = shard_range(mc.config.trajectories, rank, world)
    if stop <= start:
        raise ValueError("every rank needs at least one trajectory")
    seeds = np.ascontiguousarray(mc.config.seeds[start:stop])
    c.n_paths, c.seeds, c.seeds_len = stop - start, seeds.ctypes.data, seeds.size
    steps, degree = mc.config.steps, method.degree
    step_discount = float(df(m.rate, m.referenceDate + (T / steps) * MILLISECONDS_IN_YEAR_365))

    dev_index = rank_device(mc.device, group, device)
    dev = torch.device("cuda", dev_index)
    ctx = _ffi.get_context(dev_index)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    lib, h = ctx.lib, ctx.handle
    x = torch.zeros(lib.hh_lsm_shard_xchg_elems(steps, degree), dtype=torch.float64, device=dev)
    rows, nv, nb = steps + 1, 2 * degree + 1, degree + 1

    def exchange(n):
        if world > 1:
            _all_reduce_device(x[:n], group)

    ctx.check(lib.hh_lsm_shard_begin(h, C.byref(model), C.byref(c), degree, step_discount, x.data_ptr()))
    exchange(rows * 3)
    ctx.check(lib.hh_lsm_shard_phase(h, _ffi.HH_LSM_PHASE_POW, 0, x.data_ptr(), x.data_ptr()))
    exchange(rows * nv)
    ctx.check(lib.hh_lsm_shard_phase(h, _ffi.HH_LSM_PHASE_INIT, 0, x.data_ptr(), x.data_ptr()))
    for t in range(steps - 1, 0, -1):
        exchange(nb)
        ctx.check(lib.hh_lsm_shard_phase(h, _ffi.HH_LSM_PHASE_STEP, t, x.data_ptr(), x.data_ptr()))
    acc = torch.zeros(_ffi.HH_ACC_LEN, dtype=torch.float64, device=dev)
    ntot = (stop - start) * (2 if c.antithetic else 1)
    tau = np.empty(ntot, dtype=np.int32) if stopping_info else None
    val = np.empty(ntot) if stopping_info else None
    ctx.check(lib.hh_lsm_shard_finish(h, acc.data_ptr(), tau.ctypes.data if stopping_info else None,
                                      val.ctypes.data if stopping_info else None, None, None, None))
    if world > 1:
        _all_reduce_device(acc, group)
    acc_host = np.ascontiguousarray(acc.cpu().numpy())
    res = _ffi.hh_lsm_result()
    rc = lib.hh_lsm_finalize(acc_host.ctypes.data, C.byref(res))
    if rc != 0:
        raise _ffi.HedgehogMCError(rc, "hh_lsm_finalize failed")
    del seeds
    return LSMSolution(prob, method, res.price, (tau, val) if stopping_info else None, None,
                       std_error=res.std_error, result=res)
