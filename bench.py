#!/usr/bin/env python3
"""Benchmark of the hot path: Heston Euler–Maruyama Monte Carlo, 10^6 paths x 252 steps per GPU
(BASELINE.json metric; SURVEY.md §8d benchmark problem "H252").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no RANK in the environment starts N ranks itself (one child process per
GPU, rendezvous on 127.0.0.1) BEFORE anything touches a GPU; under torch.distributed.run the ranks
are taken from the environment and must agree with --gpus.  A box with fewer than N GPUs is an
error, never a silent single-rank run.

One "step" = one complete pass of the path over one batch: every rank integrates its trajectories
x 252 Euler steps from Wiener increments already resident in HBM (REPLAY, the mode the HBM roofline
is quoted for: 16 algorithmic bytes per path-step), reduces the discounted payoff sums, and (N > 1)
all-reduces the 16-double accumulator vector over RCCL — the path's only exchange.
  default            weak scaling: --paths (10^6) trajectories per GPU
  --global-paths G   strong scaling: G trajectories split over the ranks by contiguous ranges
                     (hedgehog_jl_amd.shard_range), e.g. north_star's 10^7

`--single-process` runs the same steps from ONE process and ONE host thread over all N devices
through the library's own multi-GPU entry point (hh_mgpu_solve_shards: per-device contexts and
streams, one ncclAllReduce of the accumulator vector inside the library, host ordered sum when RCCL
is absent) — the form a Julia host calls.  The default line embeds that form's result as
`single_process` (rank 0 starts it as a child process once every rank has left the process group).

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  value_cold      the same K steps after W warm-ups with NO clock ramp before them (value has the ramp)
  value_default_mode  = generate.value: what solve() delivers by default (in-kernel Philox)
  path_major_replay   REPLAY on the reference's own noise layout dW[path][step][comp], streamed
                  directly by euler_pm_kernel, with its own HBM roofline entry (N = 1 only)
  roofline        dominant kernel (euler_kernel REPLAY): algorithmic bytes / HIP-event time vs 8 TB/s
  cpu_baseline    the CPU oracle (oracle/hh_oracle.c, a port) timed on this host on a bounded sample
  generate        the same workload with the increments drawn in-kernel from Philox — what a caller
                  of the drop-in solve() gets (VALU-bound)
  config5_greeks  BASELINE config 5 (Δ, ∂V0, ρ in one fused pass) under the same N ranks
  strong_scaling  10^7 global trajectories split over the N ranks (when run in the weak default)
  price_check     |price - CPU reference| on identical draws (bounded sample of the same buffer)
  other_configs   BASELINE configs 2 and 4, each with its own roofline entry (N = 1 only)
  widened_rows    LSM and the exact Heston grid, each with a roofline entry (N = 1 only)
Every roofline entry: {bound, achieved, peak, unit, frac}; HBM-bound kernels price algorithmic
bytes against 8 TB/s; VALU-bound kernels price SQ_INSTS_VALU x 4 SIMD-cycles (an fp64 FMA holds a
SIMD's vector pipe for 4 cycles) against 1024 SIMDs x 2.4 GHz, the instruction counts being read
from profiles/valu_insts.json (a rocprofv3 --pmc pass, file-sourced and labelled so).  A VALU entry
also says how far the kernel is from its ALGORITHM and from the clock the chip really holds:
  floor_insts_per_unit, frac_of_floor (= floor / issued), frac_vs_floor (= frac x frac_of_floor):
                  the operations the shipped algorithm needs, one instruction each (profiles/floor_insts.json)
  sustained_clock_mhz, frac_at_sustained_clock: GRBM_GUI_ACTIVE / 8 XCDs / duration of a PMC pass on
                  file (profiles/sustained_clock.json); `frac` itself stays against the 2.4 GHz maximum
The LSM row carries per_date_us (where the 10 us of a date go); cpu_baseline is timed on the metric's
own configuration (10^6 x 252, GENERATE, all host cores) and price_check.full_config_generate compares
the two prices on it.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PATH_STEP = 16.0   # two fp64 increments read once
SIMD_CYCLES_PEAK = 256 * 4 * 2.4e9  # 256 CUs x 4 SIMDs x 2.4 GHz max clock (MI355X_MICROARCH.md)
VALU_CYCLES_PER_INST = 4.0   # wave64 fp64 FMA: 4 cycles on a SIMD (78.6 TFLOP/s fp64 vector peak)

# benchmark problem H252
H252 = dict(S0=100.0, V0=0.04, kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, r=0.03, T=1.0,
            strike=100.0, cp=1.0)
H252_ANALYTIC = 9.242521073959068  # Carr–Madan restatement, SURVEY.md §8c (sanity band only)
H252_GREEKS_FOURIER = [0.65565115, 40.7248418, 56.3225943]  # ∂S0, ∂V0, ∂r


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 timed steps after 20 warm-ups (0.25 s in all) when no flags are given
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--paths", type=int, default=1_000_000, help="trajectories per GPU (weak scaling)")
    ap.add_argument("--global-paths", type=int, default=0,
                    help="strong scaling: this many trajectories in all, split over the ranks")
    ap.add_argument("--nsteps", type=int, default=252, help="Euler steps per trajectory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip everything but the headline lines")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--ramp-ms", type=float, default=30.0,
                    help="GPU time spent on the same kernel before the W warm-up steps, so that the "
                         "timed region runs at the clock a pricing service sees in steady state "
                         "(the first ~15 ms of a fresh process run 3-8 %% slower); reported as "
                         "clock_ramp_ms, 0 disables")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process, ONE host thread drives all --gpus devices through the library's "
                         "hh_mgpu_solve_shards (RCCL all-reduce inside the library, host ordered sum when "
                         "RCCL is unavailable) instead of one rank per GPU")
    ap.add_argument("--mgpu-flags", type=int, default=0, help="hh_mgpu_create flags: 0 auto, 1 host sum, 2 RCCL")
    ap.add_argument("--allow-rccl-override", action="store_true",
                    help="accept $HEDGEHOG_MC_RCCL (another library bound in place of librccl by hh_mgpu — the "
                         "tests' stand-in); without this flag a set variable is an error, so that no line of a "
                         "real run can come from a substitute unnoticed")
    ap.add_argument("--no-single-process-child", action="store_true",
                    help="do not run the one-process form as a child (profilers that preload into children)")
    # rehearsal knobs (tests): ranks on one GPU need gloo (RCCL refuses two ranks per device)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl")
    ap.add_argument("--devices", default="", help="comma list: device ordinal of each local rank")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher/rendezvous/sharding only, no GPU work (CPU test of the N-rank path)")
    return ap.parse_args(argv)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _device_list(args):
    if args.devices:
        devs = [int(x) for x in args.devices.split(",")]
        if len(devs) != args.gpus:
            sys.exit(f"bench.py: --devices lists {len(devs)} ordinals for --gpus {args.gpus}")
        return devs
    return list(range(args.gpus))


def launch(args) -> int:
    """Parent of an N-rank run: starts one child per rank, rank 0's stdout is ours.  Nothing here
    initialises a GPU (torch.cuda.device_count() does not, on this image)."""
    devs = _device_list(args)
    if not args.rehearse:
        have = torch.cuda.device_count()
        if have <= max(devs):
            print(f"bench.py: --gpus {args.gpus} needs device ordinals {devs} but this host has "
                  f"{have} GPU(s); refusing to run fewer ranks than asked", file=sys.stderr)
            return 2
        if args.backend == "nccl" and len(set(devs)) != len(devs):
            print("bench.py: RCCL needs one device per rank (use --backend gloo to rehearse ranks "
                  "on a shared GPU)", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   HH_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]],
                                      env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 1500
    pending = list(procs)
    while pending and time.time() < deadline:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:  # one rank failed: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    for p in pending:
        p.kill()
        rc = rc or 124
    return rc


class _StdoutToStderr:
    """RCCL prints a version banner on the process's stdout when its first communicator comes up;
    rank 0's stdout is reserved for the ONE JSON line, so fd 1 points at stderr until then."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


class Dist:
    """The process group of this run (or none), with the two collectives the bench needs."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = args.backend
        self.pg = None
        if self.world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: the launcher and "
                     "the flag must agree")
        devs = _device_list(args)
        self.device_index = devs[self.local_rank] if self.local_rank < len(devs) else self.local_rank
        if "RANK" in os.environ:  # launched by us or by torch.distributed.run (also with 1 rank)
            sys.stdout.flush()
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            kw = {}
            if args.backend == "nccl" and not args.rehearse:
                torch.cuda.set_device(self.device_index)
                kw["device_id"] = torch.device("cuda", self.device_index)
            with _StdoutToStderr():  # device_id= brings the communicator up here already
                dist.init_process_group("gloo" if args.rehearse else args.backend, rank=self.rank,
                                        world_size=self.world, **kw)
            self.pg = dist

    @property
    def on(self):
        return self.pg is not None

    def all_reduce_sum(self, t, async_op=False):
        """SUM all-reduce of a device tensor: in place over RCCL; through host memory under gloo."""
        if self.pg is None:
            return None
        if t.is_cuda and self.backend == "gloo":
            h = t.cpu()
            self.pg.all_reduce(h)
            t.copy_(h)
            return None
        return self.pg.all_reduce(t, async_op=async_op)

    def max_float(self, x, dev):
        if self.pg is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.backend == "gloo" else dev)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, x, dev):
        """one float of every rank, in rank order (an all-reduce of a one-hot vector: the same collective path)"""
        if self.pg is None:
            return [x]
        t = torch.zeros(self.world, dtype=torch.float64, device="cpu" if self.backend == "gloo" else dev)
        t[self.rank] = x
        self.pg.all_reduce(t)
        return [float(v) for v in t.cpu()]

    def count_ranks(self, dev):
        """ranks that really take part in the collective: an all-reduce of ones"""
        if self.pg is None:
            return 1
        t = torch.ones(1, dtype=torch.float64, device=dev)
        self.all_reduce_sum(t)
        return int(round(float(t.item())))

    def barrier(self):
        if self.pg is not None:
            self.pg.barrier()

    def close(self):
        if self.pg is not None:
            self.pg.barrier()
            self.pg.destroy_process_group()
            self.pg = None


def _pow10(n):
    """1000000 -> '1e6', 125000 -> '125000'"""
    k = len(str(n)) - 1
    return "%de%d" % (n // 10 ** k, k) if n >= 1000 and n % 10 ** k == 0 else str(n)


def metric_string(paths, n_steps, strong=False, n_global=0):
    """the `metric` of a line names the workload that line RAN: built from the arguments, not a constant"""
    what = "%s paths in all x %d steps" % (_pow10(n_global), n_steps) if strong else \
        "%s paths x %d steps per GPU" % (_pow10(paths), n_steps)
    return "MC path-steps/sec (Heston Euler-Maruyama, %s)" % what


def predict_scaling(shard_ms, overhead_ms):
    """What ONE GPU's figures say about N ranks sharing an ensemble of fixed size (strong scaling).
    shard_ms: {N: ms of one step of the shard ceil(G/N), run alone}; overhead_ms: what a solve costs beside its
    kernel and does not shrink with the shard — enqueue, the 16-double all-reduce, the synchronisation — as ONE rank
    measures it.  -> {N: {...}}: `pipelined` = back-to-back independent solves, the exchange of one behind the kernel
    of the next (how this bench's timed loop runs): a step is the longer of the two; `single_solve` = one solve from
    call to result: kernel + overhead.  Efficiency = t(1) / (N t(N)).  No xGMI hop, no second rank, no straggler is in
    these numbers: an upper bound on what a node can show."""
    t1 = shard_ms[1]
    out = {}
    for n, t in sorted(shard_ms.items()):
        pipe, single = max(t, overhead_ms), t + overhead_ms
        out[n] = {"shard_ms_per_step": t,
                  "efficiency_pipelined": max(t1, overhead_ms) / (n * pipe),
                  "efficiency_single_solve": (t1 + overhead_ms) / (n * single)}
    return out


def shard_of(n_global, rank, world):
    """contiguous ranges of ceil(N/G) trajectories (hedgehog_jl_amd.shard_range)"""
    per = -(-n_global // world)
    a = min(n_global, rank * per)
    return a, min(n_global, a + per)


def rehearse(args, d):
    """Launcher + rendezvous + collective + sharding with no GPU: what the CPU test runs."""
    n_global = args.global_paths or args.paths * d.world
    a, b = shard_of(n_global, d.rank, d.world) if args.global_paths else \
        (d.rank * args.paths, (d.rank + 1) * args.paths)
    ranks = d.count_ranks("cpu")
    cover = torch.tensor([float(b - a)], dtype=torch.float64)
    d.all_reduce_sum(cover)
    d.close()
    if d.rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": d.world, "rccl_ranks": ranks,
                          "backend": "gloo", "scaling": "strong" if args.global_paths else "weak",
                          "global_paths": n_global, "paths_covered": int(cover.item()),
                          "shard_rank0": [a, b], "value": None}), flush=True)


def lsm_traffic(n_traj, n_dates):
    """HBM bytes of the LSM chain from the PMC passes on file (same workload only), like the headline's."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if abs(t["lsm_chain_algorithmic_bytes"] - 16.0 * n_traj * (n_dates + 1)) > 1.0:
            return {"traffic": None}
        return {"traffic": t["lsm_chain_hbm_bytes_per_solve"],
                "traffic_source": "profiles/pmc_traffic.json (" + t["lsm_source"] + "; not collected in this run)"}
    except Exception:
        return {"traffic": None}


def valu_insts():
    """VALU wave-instructions per launch of the VALU-bound kernels (rocprofv3 --pmc SQ_INSTS_VALU,
    tools/valu_insts.py) — file-sourced, the time beside it is measured live."""
    p = os.path.join(ROOT, "profiles", "valu_insts.json")
    try:
        return json.load(open(p))
    except Exception:
        return {}


def floor_insts():
    """Operations the shipped algorithm needs per unit, one VALU instruction each (tools/floor_insts.py;
    DESIGN.md §8) — what `frac` of a VALU row would have to be multiplied by to grade the kernel against its
    algorithm instead of against itself."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "floor_insts.json")))
    except Exception:
        return {}


def hbm_roofline(kernel, bytes_per_launch, ms, **extra):
    ach = bytes_per_launch / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_per_launch,
         "kernel_ms": ms}
    r.update(extra)
    return r


def valu_roofline(kernel, key, units, ms, table):
    """fp64-VALU issue fraction: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x time)."""
    ent = table.get(key)
    if not ent:
        return {"bound": "valu", "kernel": kernel, "achieved": None, "peak": SIMD_CYCLES_PEAK / 1e9,
                "unit": "G SIMD-cycles/s", "frac": None, "kernel_ms": ms,
                "note": "no instruction count for this kernel in profiles/valu_insts.json"}
    insts = ent["valu_insts_per_unit"] * units
    ach = insts * VALU_CYCLES_PER_INST / (ms * 1e-3)
    r = {"bound": "valu", "kernel": kernel, "achieved": ach / 1e9, "peak": SIMD_CYCLES_PEAK / 1e9,
         "unit": "G SIMD-cycles/s", "frac": ach / SIMD_CYCLES_PEAK, "kernel_ms": ms,
         "valu_insts_per_unit": ent["valu_insts_per_unit"], "unit_of_work": ent.get("unit"),
         "insts_source": "profiles/valu_insts.json (" + ent.get("source", "rocprofv3 --pmc") + ")"}
    try:  # the clock the chip sustains under this kernel (a PMC pass on file): `frac` is against the 2.4 GHz maximum
        clk = json.load(open(os.path.join(ROOT, "profiles", "sustained_clock.json"))).get(key)
    except Exception:
        clk = None
    if clk:
        r["sustained_clock_mhz"] = clk["clock_mhz"]
        r["frac_at_sustained_clock"] = r["frac"] * 2400.0 / clk["clock_mhz"]
        r["clock_source"] = "profiles/sustained_clock.json (rocprofv3 --pmc GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; not collected in this run)"
    fl = floor_insts().get(key)
    if fl:  # `frac` grades the kernel against its OWN instruction count; these two against its algorithm's
        r["floor_insts_per_unit"] = fl["floor_insts_per_unit"]
        r["frac_of_floor"] = fl["floor_insts_per_unit"] / ent["valu_insts_per_unit"]
        r["frac_vs_floor"] = r["frac"] * r["frac_of_floor"]
        r["floor_source"] = "profiles/floor_insts.json (tools/floor_insts.py: essential operations of the shipped algorithm, DESIGN.md §8)"
    return r


def single_process(args):
    """ONE process, N devices: every step is ONE hh_mgpu_solve_shards call of this thread (the library
    enqueues the shards concurrently from its per-device worker threads)."""
    devs = _device_list(args)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= max(devs):
        print(f"bench.py: --single-process needs device ordinals {devs}; there is no CPU fallback",
              file=sys.stderr)
        return 2
    from hedgehog_jl_amd import _ffi
    with _StdoutToStderr():  # RCCL's banner
        mg = _ffi.MultiGpu(devs, args.mgpu_flags)
    n_steps, G = args.nsteps, len(devs)
    model = _ffi.make_model(**H252)
    strong = args.global_paths > 0
    n_global = args.global_paths if strong else G * args.paths
    cfgs, keep = [], []
    for g in range(G):
        a, b = mg.shard_range(n_global, g, tile_aligned=True) if strong else (g * args.paths, (g + 1) * args.paths)
        ctx = mg.ctx(g)
        n = b - a
        seeds = np.arange(a + 1, b + 1, dtype=np.uint64)
        dW = _ffi.DeviceBuffer(ctx, 8 * ctx.lib.hh_replay_elems(n, n_steps, _ffi.HH_HESTON))
        ctx.check(ctx.lib.hh_wiener_fill(ctx.handle, _ffi.HH_HESTON, model.rho, model.T, n_steps, n,
                                         seeds.ctypes.data, 0, dW.ptr))
        ctx.synchronize()
        c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, n, n_steps, noise_mode=_ffi.HH_NOISE_REPLAY)
        c.replay, c.replay_on_device = dW.ptr, 1
        cfgs.append(c)
        keep.append(dW)

    def steps(k):
        r = None
        for _ in range(k):
            r = mg.solve_shards(model, cfgs)
        return r

    def timed(k, w, ramp_ms):
        ramp = 0.0
        if ramp_ms > 0.0:
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < ramp_ms:
                steps(4)
            ramp = (time.perf_counter() - t0) * 1e3
        steps(w)
        for g in range(G):
            mg.ctx(g).enable_timing(True)
            mg.ctx(g).synchronize()
        t0 = time.perf_counter()
        r = steps(k)            # every call returns with all devices synchronised
        dt = time.perf_counter() - t0
        kern = [mg.ctx(g).read_timings() for g in range(G)]
        for g in range(G):
            mg.ctx(g).enable_timing(False)
        return dt, kern, r, ramp

    # who really carries the exchange: counted by a collective of the library's own path, and named
    ranks_counted, mode_tested = mg.selftest()
    info = mg.rccl_info()
    dt_c, _, _, _ = timed(args.steps, args.warmup, 0.0)
    dt, kern, res, ramp = timed(args.steps, args.warmup, args.ramp_ms)
    per_shard_us, phase_us = mg.enqueue_stats()
    tot = float(n_global) * n_steps
    n0 = int(cfgs[0].n_paths)
    rccl_used = mg.reduce_mode == _ffi.HH_MGPU_REDUCE_RCCL  # AFTER the run: a collective that failed fell back
    out = {
        "metric": metric_string(args.paths, n_steps, strong, n_global),
        "value": tot * args.steps / dt, "value_cold": tot * args.steps / dt_c, "unit": "path-steps/s",
        "n_gpus": G, "single_process": True,
        "reduce": "rccl" if rccl_used else "host ordered sum",
        "reduce_mode_in_selftest": "rccl" if mode_tested == _ffi.HH_MGPU_REDUCE_RCCL else "host ordered sum",
        # ranks counted by an all-reduce of ones through the solve's own exchange (hh_mgpu_selftest); 0 = no collective ran
        "rccl_ranks": ranks_counted if (rccl_used and mode_tested == _ffi.HH_MGPU_REDUCE_RCCL) else 0,
        "ranks_counted_by_the_exchange": ranks_counted,
        "rccl_library": info["library"], "rccl_version": info["version"],
        "rccl_library_from_env": info["from_env"],
        "reduce_note": mg.last_error(),
        "steps": args.steps, "warmup": args.warmup, "clock_ramp_ms": ramp,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (Philox-generated correlated Wiener increments, resident in each GPU's HBM)",
        "config": {"workload": "HestonDynamics EulerMaruyama European call H252 (configs[2]): %d paths x %d "
                               "steps %s, NoVarianceReduction, noise REPLAY in the library's tile-major "
                               "layout (hh_wiener_fill)" % (n_global if strong else args.paths, n_steps,
                                                            "in all" if strong else "per GPU"),
                   "paths_per_gpu": n0, "n_steps": n_steps, "global_paths": n_global,
                   "parallelism": "one process, ONE call per step (hh_mgpu_solve_shards): the %d shards enqueued "
                                  "concurrently by the library's per-device threads, one 16-double all-reduce "
                                  "inside the library" % G},
        "price": res.price, "std_error": res.std_error, "analytic_carr_madan": H252_ANALYTIC,
        "per_rank_kernel_ms": [float(np.mean(k)) if len(k) else None for k in kern],
        "enqueue_host_us": {"per_shard": per_shard_us, "phase": phase_us},
        "roofline": dict(hbm_roofline("euler_kernel<HestonModel,REPLAY> (device %d's launches)" % devs[0],
                                      BYTES_PER_PATH_STEP * n0 * n_steps, float(np.mean(kern[0]))),
                         traffic=None, launches_timed=len(kern[0])),
    }
    print(json.dumps(out), flush=True)
    mg.close()
    return 0


def run_single_process_child(args, world, timeout=240):
    """The one-process form of the same run, as a child process (it owns its devices' contexts and its
    RCCL communicators; a hang or failure there costs this block, not the line)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--single-process",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--paths", str(args.paths),
           "--nsteps", str(args.nsteps), "--ramp-ms", str(args.ramp_ms)]
    if args.devices:
        cmd += ["--devices", args.devices]
    if args.allow_rccl_override:
        cmd += ["--allow-rccl-override"]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HH_BENCH_CHILD",
                        "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        for line in reversed(p.stdout.splitlines()):
            if line.startswith("{"):
                j = json.loads(line)
                return {k: j.get(k) for k in ("value", "value_cold", "unit", "n_gpus", "reduce", "reduce_note",
                                              "rccl_ranks", "ranks_counted_by_the_exchange", "rccl_library",
                                              "rccl_version", "rccl_library_from_env", "per_rank_kernel_ms",
                                              "enqueue_host_us", "ms_per_step", "price", "steps", "warmup", "roofline")}
        return {"error": "no JSON line", "rc": p.returncode, "stderr_tail": p.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %d s" % timeout}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def main():
    args = parse()
    if os.environ.get("HEDGEHOG_MC_RCCL") and not args.allow_rccl_override:
        print("bench.py: $HEDGEHOG_MC_RCCL is set (%s): hh_mgpu would bind that library in place of librccl. "
              "Unset it, or pass --allow-rccl-override to run with it knowingly (the line then names it)."
              % os.environ["HEDGEHOG_MC_RCCL"], file=sys.stderr)
        sys.exit(2)
    if args.single_process:
        sys.exit(single_process(args))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch(args))
    d = Dist(args)
    if args.rehearse:
        rehearse(args, d)
        return
    rank, world = d.rank, d.world
    if not torch.cuda.is_available() or torch.cuda.device_count() <= d.device_index:
        sys.exit(f"bench.py: rank {rank} needs cuda:{d.device_index}; there is no CPU fallback")
    torch.cuda.set_device(d.device_index)
    dev = torch.device("cuda", d.device_index)

    import hedgehog_jl_amd as hh
    from hedgehog_jl_amd import _ffi

    ctx = hh.Context(dev.index)  # raises without a HIP device — there is no CPU fallback
    lib, h = ctx.lib, ctx.handle
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    with _StdoutToStderr():  # the first collective brings RCCL's communicator (and its banner) up
        rccl_ranks = d.count_ranks(dev)
        torch.cuda.synchronize(dev)
    if rccl_ranks != world:
        sys.exit(f"bench.py: all-reduce of ones gave {rccl_ranks}, expected {world}")

    n_steps = args.nsteps
    model = _ffi.make_model(**H252)

    class Shard:
        """This rank's trajectories [g0, g0 + n) and their increments, resident in HBM."""

        def __init__(self, g0, n):
            self.g0, self.n = g0, n
            # seeds[i] = global 1-based trajectory index
            self.seeds = torch.arange(g0 + 1, g0 + n + 1, dtype=torch.int64, device=dev)
            n_el = lib.hh_replay_elems(n, n_steps, _ffi.HH_HESTON)
            self.dW = torch.empty(n_el, dtype=torch.float64, device=dev)
            ctx.check(lib.hh_wiener_fill(h, _ffi.HH_HESTON, model.rho, model.T, n_steps, n,
                                         self.seeds.data_ptr(), 1, self.dW.data_ptr()))

        def config(self, noise, n_partials=0):
            c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, self.n, n_steps,
                                 noise_mode=noise, n_partials=n_partials)
            c.seeds, c.seeds_on_device, c.seeds_len = self.seeds.data_ptr(), 1, self.n
            c.replay, c.replay_on_device, c.replay_len = self.dW.data_ptr(), 1, self.dW.numel()
            return c

    # Steps are independent pricing jobs: the (latency-bound, 128-byte) all-reduce of step k is
    # issued asynchronously on RCCL's stream and overlaps the simulation kernel of step k+1; two
    # accumulator buffers alternate, and every all-reduce has completed before the clock stops.
    accums = [torch.zeros(_ffi.HH_ACC_LEN, dtype=torch.float64, device=dev) for _ in range(2)]
    pending = [None, None]

    def step(mdl, cfg, i):
        b = i & 1
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(cfg), accums[b].data_ptr(), None))
        if d.on:  # the path's one exchange: 16 doubles, SUM
            pending[b] = d.all_reduce_sum(accums[b], async_op=True)
        return b

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def timed(mdl, cfg, k, w, ramp_ms=0.0):
        """W untimed warm-up steps, then EXACTLY k steps between barrier + synchronize on both
        sides; MAX over ranks.  -> (seconds, per-launch kernel ms, accumulator, ramp ms spent)"""
        ramp = 0.0
        if ramp_ms > 0.0:
            # bring the device to its steady clock on the same kernel (disclosed as clock_ramp_ms).
            # Kernel launches only, NO collective: every rank loops on its own clock, so the ranks
            # run different numbers of iterations
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < ramp_ms:
                for _ in range(8):
                    ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(cfg),
                                                   accums[0].data_ptr(), None))
                torch.cuda.synchronize(dev)
            ramp = (time.perf_counter() - t0) * 1e3
        last = 0
        for i in range(w):
            step(mdl, cfg, i)
        drain()
        ctx.enable_timing(True)
        d.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(k):
            last = step(mdl, cfg, i)
        drain()
        torch.cuda.synchronize(dev)
        d.barrier()
        dt = time.perf_counter() - t0
        kern_ms = ctx.read_timings()
        ctx.enable_timing(False)
        dt = d.max_float(dt, dev)
        return dt, kern_ms, accums[last].cpu().numpy().copy(), d.max_float(ramp, dev)

    def finalize(mdl, cfg, acc):
        r = _ffi.hh_result()
        lib.hh_mc_finalize(C.byref(mdl), C.byref(cfg), acc.ctypes.data, C.byref(r))
        return r

    # ---- the headline workload ------------------------------------------------------------------
    strong = args.global_paths > 0
    if strong:
        g0, g1 = shard_of(args.global_paths, rank, world)
        if g1 <= g0:
            sys.exit("bench.py: --global-paths leaves a rank without trajectories")
        n_global = args.global_paths
    else:
        g0, g1 = rank * args.paths, (rank + 1) * args.paths
        n_global = world * args.paths
    sh = Shard(g0, g1 - g0)
    n_paths = sh.n
    cfg_rep, cfg_gen = sh.config(_ffi.HH_NOISE_REPLAY), sh.config(_ffi.HH_NOISE_GENERATE)

    dt_cold, _, _, _ = timed(model, cfg_rep, args.steps, args.warmup, 0.0)  # no ramp: first GPU work after the fill
    dt_rep, kern_rep, acc_rep, ramp_ms = timed(model, cfg_rep, args.steps, args.warmup, args.ramp_ms)
    dt_gen, kern_gen, acc_gen, _ = timed(model, cfg_gen, args.steps, args.warmup, args.ramp_ms)
    res, res_gen = finalize(model, cfg_rep, acc_rep), finalize(model, cfg_gen, acc_gen)
    total_path_steps = float(n_global) * n_steps
    value = total_path_steps * args.steps / dt_rep
    vt = valu_insts() if rank == 0 else {}

    kern_ms = float(np.mean(kern_rep))
    per_rank_kernel_ms = d.gather_floats(kern_ms, dev)
    per_rank_step_ms = d.gather_floats(dt_rep / args.steps * 1e3, dev)
    # which collective library carried the run's all-reduces (torch.distributed's backend "nccl" IS RCCL on ROCm)
    coll = {"backend": args.backend, "ranks_counted_by_all_reduce_of_ones": rccl_ranks}
    if args.backend == "nccl" and d.on:
        try:
            coll["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            coll["rccl_version"] = repr(e)
        try:  # the file the process mapped it from
            libs = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln or "libnccl" in ln})
            coll["rccl_library"] = libs
        except Exception:  # noqa: BLE001
            coll["rccl_library"] = None
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath) and n_paths == 1_000_000 and n_steps == 252:
        try:  # PMC counters cannot be read from inside this process: a committed rocprofv3 pass
            tj = json.load(open(tpath))
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, " \
                             "separate passes, same workload; not measured in this run)"
        except Exception:
            traffic = None

    out = {
        "metric": metric_string(args.paths, n_steps, strong, n_global),
        "value": value,
        "value_cold": total_path_steps * args.steps / dt_cold,
        "value_default_mode": total_path_steps * args.steps / dt_gen,
        "unit": "path-steps/s",
        "n_gpus": world,
        "rccl_ranks": rccl_ranks,
        "collective": coll,
        "per_rank_kernel_ms": per_rank_kernel_ms,
        "per_rank_ms_per_step": per_rank_step_ms,
        "steps": args.steps,
        "warmup": args.warmup,
        "clock_ramp_ms": ramp_ms,
        "ms_per_step": dt_rep / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (Philox-generated correlated Wiener increments, resident in HBM)",
        "config": {"workload": "HestonDynamics EulerMaruyama European call H252 "
                               "(configs[2]): %d paths x %d steps %s, NoVarianceReduction, "
                               "noise REPLAY in the library's tile-major layout dW[tile][step][comp][256] "
                               "(hh_wiener_fill); the reference's own layout dW[path][step][comp] is the "
                               "path_major_replay block" % (n_paths if not strong else n_global, n_steps,
                                                            "in all" if strong else "per GPU"),
                   "paths_per_gpu": n_paths, "n_steps": n_steps, "global_paths": n_global,
                   "backend": "rccl" if args.backend == "nccl" else args.backend,
                   "parallelism": "path-sharded x%d, one 16-double all-reduce" % world},
        "price": res.price,
        "std_error": res.std_error,
        "analytic_carr_madan": H252_ANALYTIC,
        "roofline": dict(hbm_roofline("euler_kernel<HestonModel,REPLAY> (rank 0's launches)",
                                      BYTES_PER_PATH_STEP * n_paths * n_steps, kern_ms),
                         traffic=traffic, traffic_source=traffic_source,
                         kernel_ms_avg=kern_ms, kernel_ms_min=float(np.min(kern_rep)),
                         launches_timed=len(kern_rep)),
        "generate": {
            "value": total_path_steps * args.steps / dt_gen, "unit": "path-steps/s",
            "what": "the same workload with in-kernel Philox — the mode hh.solve() / the Julia "
                    "solve_hip() use by default (= value_default_mode); REPLAY needs the caller's increments",
            "ms_per_step": dt_gen / args.steps * 1e3,
            "price": res_gen.price,
            "rel_diff_vs_replay": abs(res_gen.price - res.price) / abs(res.price),
            "roofline": valu_roofline("euler_kernel<HestonModel,GENERATE>", "heston_euler_generate",
                                      float(n_paths) * n_steps, float(np.mean(kern_gen)), vt)},
    }

    if not args.no_extra:
        # ---- BASELINE config 5 under the same N ranks: Δ, ∂V0, ρ in one fused pass ---------------
        sd = {"S0": [1, 0, 0], "V0": [0, 1, 0], "r_drift": [0, 0, 1],
              "discount": [0, 0, -float(np.exp(-H252["r"] * H252["T"]))]}
        m5 = _ffi.make_model(**H252, seeds=sd, n_partials=3)
        c5 = sh.config(_ffi.HH_NOISE_REPLAY, n_partials=3)
        k5 = min(args.steps, 50)
        dt5, kern5, acc5, _ = timed(m5, c5, k5, min(args.warmup, 5), args.ramp_ms)
        r5 = finalize(m5, c5, acc5)
        t5 = float(np.mean(kern5))
        out["config5_greeks"] = {
            "what": "BatchGreekProblem (Δ, ∂V0, ρ) as 3 dual partials through Heston Euler, REPLAY, "
                    "%d ranks, one all-reduce of the same 16 doubles" % world,
            "value": total_path_steps * k5 / dt5, "unit": "path-steps/s", "steps": k5,
            "ms_per_step": dt5 / k5 * 1e3, "price": r5.price,
            "greeks": [r5.dprice[k] for k in range(3)], "fourier": H252_GREEKS_FOURIER,
            "roofline": hbm_roofline("euler_kernel<HestonModel,P=1,REPLAY> (one carried derivative)",
                                     BYTES_PER_PATH_STEP * n_paths * n_steps, t5)}

        # ---- strong scaling on north_star's 10^7 trajectories (when the headline ran weak) ---------
        if not strong:
            del cfg_rep, cfg_gen, c5
            G = 10_000_000
            a, b = shard_of(G, rank, world)
            sh = None
            torch.cuda.empty_cache()
            sh = Shard(a, b - a)
            cs = sh.config(_ffi.HH_NOISE_REPLAY)
            ks = min(args.steps, 40 if world == 1 else 100)
            dts, kerns, accs, _ = timed(model, cs, ks, min(args.warmup, 5), args.ramp_ms)
            rs = finalize(model, cs, accs)
            out["strong_scaling"] = {
                "global_paths": G, "paths_this_rank": sh.n, "scaling": "strong", "steps": ks,
                "value": float(G) * n_steps * ks / dts, "unit": "path-steps/s",
                "ms_per_step": dts / ks * 1e3, "price": rs.price, "std_error": rs.std_error,
                "roofline": hbm_roofline("euler_kernel<HestonModel,REPLAY> (rank 0's shard)",
                                         BYTES_PER_PATH_STEP * sh.n * n_steps,
                                         float(np.mean(kerns)))}
            del cs
            sh = None
            torch.cuda.empty_cache()
            sh = Shard(g0, g1 - g0)  # the headline shard again, for the single-GPU extras below

    # every collective of the run is behind us: all ranks leave the process group together, here;
    # rank 0 then measures the single-GPU extras (other configs, CPU baseline) on its own
    d.close()
    if rank != 0:
        return

    if not args.no_extra and not args.no_single_process_child:
        if world > 1:
            time.sleep(2.0)  # the other ranks are leaving their GPUs
        out["single_process"] = run_single_process_child(args, world)

    # ---- what one GPU can say about the 1 -> 8 curve: each shard of ceil(G/N) alone ---------------
    if world == 1 and not args.no_extra:
        sp = out.get("single_process") or {}
        try:  # a solve's cost beside its kernel, as the one-process form measured it with ONE rank
            overhead_ms = max(0.0, float(sp["ms_per_step"]) - float(sp["per_rank_kernel_ms"][0]))
        except Exception:  # noqa: BLE001
            overhead_ms = None
        sh = None
        torch.cuda.empty_cache()
        pred = {"label": "one GPU's figures, not a node's: every shard of ceil(G/N) trajectories was run ALONE on this "
                         "GPU; the per-solve overhead is the one-process form's step time minus its kernel time with "
                         "ONE rank (enqueue + all-reduce of 16 doubles + synchronisation).  No xGMI hop, no second "
                         "rank, no straggler is in these numbers — an upper bound to read a SCALE record against",
                "per_solve_overhead_ms_one_rank": overhead_ms,
                "enqueue_host_us": sp.get("enqueue_host_us"), "reduce_of_that_run": sp.get("reduce"),
                "strong": {}}
        for G in (1_000_000, 10_000_000):
            big = Shard(0, G)  # a prefix of its tiles is the shard of a smaller rank count
            shard_ms = {}
            for N in (1, 2, 4, 8):
                n = -(-G // N)
                cN = big.config(_ffi.HH_NOISE_REPLAY)
                cN.n_paths = n
                cN.seeds_len = n
                cN.replay_len = lib.hh_replay_elems(n, n_steps, _ffi.HH_HESTON)
                k = min(args.steps, 40)
                dtN, _, _, _ = timed(model, cN, k, min(args.warmup, 5), args.ramp_ms)
                shard_ms[N] = dtN / k * 1e3
            rows = predict_scaling(shard_ms, overhead_ms or 0.0)
            for N, row in rows.items():
                row["paths_per_rank"] = -(-G // N)
                row["predicted_value_pipelined"] = float(G) * n_steps / (max(row["shard_ms_per_step"], overhead_ms or 0.0) * 1e-3)
            pred["strong"][str(G)] = {str(N): row for N, row in rows.items()}
            big = None
            torch.cuda.empty_cache()
        # weak scaling (the driver's SCALE runs: --paths per GPU fixed): every rank repeats the N = 1 step; what is
        # added is the exchange, hidden behind the next step's kernel as long as it is shorter than the kernel
        t1 = dt_rep / args.steps * 1e3
        pred["weak_%s_per_gpu" % _pow10(n_paths)] = {
            str(N): {"ms_per_step": max(t1, overhead_ms or 0.0),
                     "efficiency_pipelined": t1 / max(t1, overhead_ms or 0.0)} for N in (1, 2, 4, 8)}
        out["predicted_scaling"] = pred
        sh = Shard(g0, g1 - g0)  # the headline shard again, for the extras below

    accum = accums[0]
    if world == 1 and not args.no_extra:
        def kernel_ms(mdl, cfg, reps=10, warm_ms=15.0):
            # each kernel is timed at ITS steady clock: the chip's clock follows the load of the last
            # milliseconds, and a VALU-heavy kernel measured right behind a memory-bound one (or the
            # reverse) reads 10-25 % off for its first ~10 ms
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < warm_ms:
                for _ in range(4):
                    ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(cfg), accum.data_ptr(), None))
                torch.cuda.synchronize(dev)
            ctx.enable_timing(True)
            for _ in range(reps):
                ctx.check(lib.hh_mc_accumulate(h, C.byref(mdl), C.byref(cfg), accum.data_ptr(), None))
            t = float(np.median(ctx.read_timings()))
            ctx.enable_timing(False)
            return t, finalize(mdl, cfg, accum.cpu().numpy().copy())

        # `solve_ms` of the rows below is the HIP-event time of EVERYTHING one hh_mc_accumulate enqueues (since
        # round 5 the timing hook brackets the call's last kernel too: a record reduction that is a kernel of its
        # own — Broadie–Kaya, baskets — is inside)
        def multi_ms(models, cfg, reps=20, warm_ms=40.0, slots_per_call=1):
            K = len(models)
            arr = (_ffi.hh_model * K)(*models)
            acc_k = torch.zeros(K * _ffi.HH_ACC_LEN, dtype=torch.float64, device=dev)
            call = lambda: ctx.check(lib.hh_mc_accumulate_multi(h, arr, K, C.byref(cfg), acc_k.data_ptr(), None))  # noqa: E731
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < warm_ms:
                for _ in range(4):
                    call()
                torch.cuda.synchronize(dev)
            ctx.enable_timing(True)
            for _ in range(reps):
                call()
            # (a Broadie–Kaya call closes one timing slot per model: the chain's, then each finish pass's)
            t = float(np.median(np.asarray(ctx.read_timings()).reshape(reps, slots_per_call).sum(axis=1)))
            ctx.enable_timing(False)
            a = acc_k.cpu().numpy().copy()
            return t, [finalize(models[k], cfg, a[k * _ffi.HH_ACC_LEN:(k + 1) * _ffi.HH_ACC_LEN].copy()) for k in range(K)]

        # ---- FiniteDifference(1e-3, FDCentral) delta: the two bumped solves of greeks_problem.jl:296-303 on
        #      the same draws in ONE pass (hh_mc_accumulate_multi), both noise modes
        eps = 1e-3
        up, dn = dict(H252, S0=H252["S0"] * (1 + eps)), dict(H252, S0=H252["S0"] * (1 - eps))
        fd_models = [_ffi.make_model(**up), _ffi.make_model(**dn)]
        fd = {"what": "solve(GreekProblem(prob, spot), FiniteDifference(1e-3), MonteCarlo): the reference runs two full "
                      "solves on the same seeds (greeks_problem.jl:296-303); here both models are stepped on each draw "
                      "in one pass — bit-identical prices (tests/test_gpu_multi.py)", "bump": eps}
        for key, cfg_n in (("generate", sh.config(_ffi.HH_NOISE_GENERATE)), ("replay", sh.config(_ffi.HH_NOISE_REPLAY))):
            t_one, _ = kernel_ms(fd_models[0], cfg_n, reps=20, warm_ms=40.0)
            t_two, rr = multi_ms(fd_models, cfg_n)
            delta = (rr[0].price - rr[1].price) / (2 * eps * H252["S0"])
            ent = {"solve_ms": t_two, "two_separate_solves_ms": 2 * t_one, "ratio": t_two / (2 * t_one),
                   "delta": delta, "delta_fourier": H252_GREEKS_FOURIER[0],
                   "model_path_steps_per_s": 2.0 * n_paths * n_steps / (t_two * 1e-3)}
            if key == "replay":  # the 16 bytes of a pair of increments now serve TWO path-steps
                ent["roofline"] = hbm_roofline("euler_multi_kernel<HestonModel,REPLAY,K=2>",
                                               BYTES_PER_PATH_STEP * n_paths * n_steps, t_two,
                                               bytes_per_model_path_step=BYTES_PER_PATH_STEP / 2)
                ent["roofline_valu"] = valu_roofline("euler_multi_kernel<HestonModel,REPLAY,K=2>", "heston_euler_replay_multi2",
                                                     float(n_paths) * n_steps, t_two, vt)
            else:
                ent["roofline"] = valu_roofline("euler_multi_kernel<HestonModel,GENERATE,K=2>", "heston_euler_generate_multi2",
                                                float(n_paths) * n_steps, t_two, vt)
            fd[key] = ent
        out["fd_central_delta_H252"] = fd
        # … and of the same delta on the exact (Broadie–Kaya) law: a bumped spot is invisible to the variance process, so
        # ONE chain serves both models (bk_refinish_kernel) — the reference has no other way to a Broadie–Kaya Greek
        # than these bumps (no dual numbers through rand!, heston.jl:261-276)
        cbk = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n_paths)
        cbk.seeds, cbk.seeds_on_device = sh.seeds.data_ptr(), 1
        t_bk1, _ = kernel_ms(fd_models[0], cbk, reps=5)
        t_bk2, rbk = multi_ms(fd_models, cbk, reps=10, warm_ms=20.0, slots_per_call=2)
        out["fd_central_delta_broadie_kaya"] = {
            "what": "solve(GreekProblem(prob, spot), FiniteDifference(1e-3), MonteCarlo(HestonBroadieKaya)): two full solves "
                    "in the reference; here one chain and a finish pass — each price bit-identical to its own solve "
                    "(tests/test_gpu_multi.py)",
            "solve_ms": t_bk2, "two_separate_solves_ms": 2 * t_bk1, "ratio": t_bk2 / (2 * t_bk1),
            "delta": (rbk[0].price - rbk[1].price) / (2 * eps * H252["S0"]), "delta_fourier": H252_GREEKS_FOURIER[0]}

        c4 = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n_paths)
        c4.seeds, c4.seeds_on_device = sh.seeds.data_ptr(), 1
        t4, r4 = kernel_ms(model, c4, reps=5)
        # the same law at ten times the size: what does not scale with the ensemble (the tail kernel, the CF kernel's
        # drain: ~40 us, profiles/r06_ae_bk_sizes.txt) is 13 % of the 10^6 row and 1.5 % of this one
        n4b = 10_000_000
        c4b = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n4b)
        c4b.seeds, c4b.seeds_on_device = sh.seeds.data_ptr(), 1
        t4b, r4b = kernel_ms(model, c4b, reps=5)
        m2 = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0)
        c2 = _ffi.make_config(_ffi.HH_LOGNORMAL, _ffi.HH_EXACT_LAW, n_paths)
        c2.seeds, c2.seeds_on_device = sh.seeds.data_ptr(), 1
        t2, r2 = kernel_ms(m2, c2)
        n2b = 100_000_000  # the same law at a size where the kernel, not its launch, is what is timed
        c2b = _ffi.make_config(_ffi.HH_LOGNORMAL, _ffi.HH_EXACT_LAW, n2b)
        c2b.seeds, c2b.seeds_on_device = sh.seeds.data_ptr(), 1
        t2b, r2b = kernel_ms(m2, c2b, reps=10)
        ca = sh.config(_ffi.HH_NOISE_REPLAY)
        ca.antithetic = 1
        ta, ra = kernel_ms(model, ca)
        out["other_configs"] = {
            "config2_lognormal_exact": {
                "paths_per_s": n_paths / (t2 * 1e-3), "solve_ms": t2, "kernel_ms": t2, "price": r2.price,
                "analytic": 10.450583572185565,
                "note": "BASELINE's size (10^6): ONE launch (489 workgroups, four pairs of trajectories per lane, the last "
                        "workgroup adds the records), ~11 us by the HIP events around a single call, 8.6 us per solve back to "
                        "back (profiles/r06_c_exact_pairs_ab.txt) — launch-bound, its fraction says little; the 10^8 row "
                        "beside it is the kernel",
                "roofline": valu_roofline("exact_gbm_kernel<4 pairs per lane>", "lognormal_exact", float(n_paths), t2, vt)},
            "config2_lognormal_exact_1e8": {
                "paths": n2b, "paths_per_s": n2b / (t2b * 1e-3), "solve_ms": t2b, "price": r2b.price,
                "std_error": r2b.std_error, "analytic": 10.450583572185565,
                "roofline": valu_roofline("exact_gbm_kernel<64 pairs per lane>", "lognormal_exact_1e8", float(n2b), t2b, vt)},
            "config4_broadie_kaya": {
                "paths_per_s": n_paths / (t4 * 1e-3), "solve_ms": t4, "kernel_ms": t4, "price": r4.price,
                "std_error": r4.std_error, "cf_terms_per_path": r4.bk_cf_terms / n_paths,
                "bisect_fallbacks": int(r4.bk_bisect_fallback),
                "roofline": valu_roofline("bk_cf_kernel (draws, series, inversion, the ladder walked by one wave per tile) + bk_tail_kernel (long series: none; the record sums)",
                                          "broadie_kaya", float(n_paths), t4, vt)},
            "config4_broadie_kaya_1e7": {
                "paths": n4b, "paths_per_s": n4b / (t4b * 1e-3), "solve_ms": t4b, "price": r4b.price,
                "std_error": r4b.std_error, "cf_terms_per_path": r4b.bk_cf_terms / n4b,
                "note": "the 10^6 row is BASELINE's size; this one shows the chain where its fixed ~40 us (the drain of a grid whose waves issue oldest-first, the tail kernel: profiles/r06_ab_bk_last_round_priority.txt) no longer count",
                "roofline": valu_roofline("the same chain of kernels", "broadie_kaya", float(n4b), t4b, vt)},
            "config3_antithetic_replay": {
                "integrated_path_steps_per_s": 2.0 * n_paths * n_steps / (ta * 1e-3),
                "solve_ms": ta, "kernel_ms": ta, "price": ra.price, "std_error": ra.std_error,
                "roofline": hbm_roofline("euler_kernel<HestonModel,REPLAY,ANTI>",
                                         BYTES_PER_PATH_STEP * n_paths * n_steps, ta)},
        }

        # REPLAY on the reference's own noise layout (montecarlo.jl:258,370: W.W per trajectory):
        # dW[path][step][comp], consumed as it stands by euler_pm_kernel — no repack pass
        pm = sh.dW.view(-1, n_steps, 2, _ffi.HH_TILE_PATHS).permute(0, 3, 1, 2).reshape(-1, n_steps, 2)[:n_paths].contiguous()
        torch.cuda.synchronize(dev)
        cp = sh.config(_ffi.HH_NOISE_REPLAY)
        cp.replay, cp.replay_len, cp.replay_layout = pm.data_ptr(), pm.numel(), _ffi.HH_REPLAY_PATH_MAJOR
        tp, rp = kernel_ms(model, cp, reps=40)
        pm_traffic = None
        try:
            pm_traffic = json.load(open(tpath)).get("path_major_hbm_bytes_per_launch")
        except Exception:
            pass
        out["path_major_replay"] = {
            "what": "the same increments in the reference's layout dW[path][step][comp], device-resident, "
                    "streamed directly (128-byte lines by LDS-DMA into a swizzled wave-private LDS image)",
            "value": float(n_paths) * n_steps / (tp * 1e-3), "unit": "path-steps/s", "kernel_ms": tp,
            "price": rp.price, "same_price_as_tile_major": rp.price == res.price,
            "roofline": dict(hbm_roofline("euler_pm_kernel<HestonModel>", BYTES_PER_PATH_STEP * n_paths * n_steps, tp),
                             traffic=pm_traffic,
                             traffic_source="profiles/pmc_traffic.json (rocprofv3 --pmc, separate passes)"
                             if pm_traffic else None)}
        del pm, cp
        torch.cuda.empty_cache()

        # the rows widened beyond the headline path: one timing each, same C-ABI
        import math
        n_l, st_l = 1_000_000, 100
        m_l = _ffi.make_model(S0=100.0, sigma=0.2, r=0.05, T=1.0, strike=100.0, cp=-1.0)
        c_l = _ffi.make_config(_ffi.HH_LOGNORMAL, _ffi.HH_EXACT_LAW, n_l, st_l, antithetic=1)
        c_l.seeds, c_l.seeds_on_device, c_l.seeds_len = sh.seeds.data_ptr(), 1, n_paths
        r_l = _ffi.hh_lsm_result()
        t_l = []
        for _ in range(4):
            ctx.check(lib.hh_lsm_solve(h, C.byref(m_l), C.byref(c_l), 5, math.exp(-0.05 / st_l),
                                       C.byref(r_l), None, None, None))
            t_l.append(r_l.kernel_ms)
        t_lsm = float(np.min(t_l[1:]))
        n_g, st_g = 200_000, 12
        c_g = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_BROADIE_KAYA, n_g, st_g)
        c_g.seeds, c_g.seeds_on_device, c_g.seeds_len = sh.seeds.data_ptr(), 1, n_paths
        r_g = _ffi.hh_result()
        # timed like every other row (kernel_ms above): the same call for 15 ms first — this chain is VALU-bound
        # and follows the memory-bound LSM solves — then the median of five
        t0, t_g = time.perf_counter(), []
        while (time.perf_counter() - t0) * 1e3 < 15.0 or len(t_g) < 5:
            ctx.check(lib.hh_heston_exact_grid(h, C.byref(model), C.byref(c_g), None, None, 0,
                                               C.byref(r_g)))
            t_g.append(r_g.kernel_ms)
        t_grid = float(np.median(t_g[-5:]))
        # the calibration objective's inner loop (calibration.jl:75-88): 100 Heston quotes by
        # Carr–Madan in ONE launch, plain and with the 5-parameter gradient; wall time per evaluation
        import hedgehog_jl_amd as hh
        from hedgehog_jl_amd.dual import Dual
        ref_d = hh.Date(2021, 1, 1)
        quotes = [hh.VanillaOption(float(K), e, hh.European(), hh.Call(), hh.Spot())
                  for e in (hh.Date(2021, 4, 1), hh.Date(2021, 7, 1), hh.Date(2022, 1, 1), hh.Date(2023, 1, 1))
                  for K in np.linspace(70.0, 140.0, 25)]
        e5 = lambda j: tuple(1.0 if i == j else 0.0 for i in range(5))  # noqa: E731
        mkts = (hh.HestonInputs(ref_d, 0.03, 100.0, 0.04, 2.0, 0.04, 0.3, -0.7),
                hh.HestonInputs(ref_d, 0.03, 100.0, Dual(0.04, e5(0)), Dual(2.0, e5(1)), Dual(0.04, e5(2)),
                                Dual(0.3, e5(3)), Dual(-0.7, e5(4))))
        cm, t_cm = hh.CarrMadan(1.0, 32.0, hh.HestonDynamics()), []
        for mkt in mkts:
            bp = hh.BasketPricingProblem(quotes, mkt)
            hh.solve(bp, cm)
            t0 = time.perf_counter()
            for _ in range(20):
                hh.solve(bp, cm)
            t_cm.append((time.perf_counter() - t0) / 20 * 1e3)

        out["widened_rows"] = {
            "carr_madan_basket_100_heston_quotes": {
                "wall_ms_prices": t_cm[0], "wall_ms_prices_and_5_parameter_gradient": t_cm[1],
                "what": "hh.solve(BasketPricingProblem, CarrMadan): one kernel launch per call, host "
                        "wall time including the result copy"},
            "lsm_american_put_2e6_paths_x_100_dates": {
                "kernel_ms": t_lsm, "price": r_l.price, "std_error": r_l.std_error,
                # the spot grid written once by the path kernel and read once by the backward
                # induction: 2 x 8 B per (trajectory, date)
                "roofline": hbm_roofline("gbm_grid_kernel + LSM backward induction (whole chain)",
                                         16.0 * 2 * n_l * (st_l + 1), t_lsm, **lsm_traffic(2 * n_l, st_l)),
                # what actually bounds a date of the induction is the SIMDs' issue rate (and ~2.5 µs of
                # synchronisation): the same chain against the fp64-VALU issue roofline
                "roofline_valu": valu_roofline("gbm_grid_kernel + lsm_persistent_kernel + lsm_final_kernel",
                                               "lsm_chain", 2.0 * n_l * st_l, t_lsm, vt),
                # why 0.28 of the HBM floor is not wasted traffic (PMC: 1.035 x algorithmic): the induction is
                # a chain of 100 dates, each waiting for the one before
                "per_date_us": {
                    "dates": st_l, "grid_kernel_ms_under_profiler": 0.405,
                    "induction_us_per_date": (t_lsm - 0.405) * 1e3 / st_l,
                    "budget_us": {"moment sums, statistics, power sums, decisions, totals of both record halves "
                                  "(arithmetic of the two waves a SIMD holds)": 7.4,
                                  "normal equations, one wave (LDL^T, rows over its lanes)": 1.35,
                                  "all-gather of the 256 workgroups' records (from the last wave's arrival)": 1.1,
                                  "loop overhead": 0.3},
                    "budget_source": "in-kernel s_memrealtime stamps of a -DHH_LSM_STAMPS build, profiles/r03_g_lsm_cuts.txt "
                                     "(box 4), r03_b_lsm_phase_stamps.txt; not collected in this run",
                    "reading": "a date costs ~10 µs whatever the row's 16 MB cost to stream (2 µs at 8 TB/s): the row is "
                               "read at 1.6 TB/s because the next date cannot start before this one's regression is solved"}},
            "heston_exact_grid_2e5_paths_x_12_dates": {
                "kernel_ms": t_grid, "transitions_per_s": n_g * st_g / (t_grid * 1e-3),
                "cf_terms_per_transition": r_g.bk_cf_terms / (n_g * st_g),
                "roofline": valu_roofline("bk_draw_grid_kernel + counting sort of the pairs by one 8-bit key (three kernels) + ONE bk chain over all (date, trajectory) pairs, read through the order + bk_grid_spots_kernel",
                                          "heston_exact_grid", float(n_g) * st_g, t_grid, vt)},
        }

    # ---- bounded-sample checks against the CPU oracle (rank 0, N = 1 only) ------------------
    if world == 1 and not args.no_cpu_baseline:
        from tests import oracle_ffi  # the ONLY use of the oracle here: checker + timed CPU baseline
        orc = oracle_ffi.load()
        tiles = 200                                   # 51,200 trajectories of the SAME buffer
        ns = min(n_paths, tiles * _ffi.HH_TILE_PATHS)
        n_s_el = lib.hh_replay_elems(ns, n_steps, _ffi.HH_HESTON)
        dW_s = sh.dW[:n_s_el].cpu().numpy()
        c_s = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, ns, n_steps,
                               noise_mode=_ffi.HH_NOISE_REPLAY, replay=dW_s)
        r_gpu = _ffi.hh_result()
        ctx.check(lib.hh_mc_solve(h, C.byref(model), C.byref(c_s), C.byref(r_gpu), None))
        threads = orc.num_threads()
        r_cpu, _, _ = orc.mc_solve(model, c_s, want_terminal=False)  # also warms the CPU
        out["price_check"] = {
            "sample_paths": ns, "gpu": r_gpu.price, "cpu_ref": r_cpu.price,
            "rel_err": abs(r_gpu.price - r_cpu.price) / abs(r_cpu.price),
            "what": "same Wiener increments through the HIP kernel and the CPU oracle"}
        # the CPU baseline on the metric's OWN configuration: 10^6 x 252, increments drawn per trajectory from
        # its seed (GENERATE, what solve() runs by default) — the REPLAY sample above stays the price check
        seeds_h = np.arange(sh.g0 + 1, sh.g0 + n_paths + 1, dtype=np.uint64) if hasattr(sh, "g0") else \
            np.arange(1, n_paths + 1, dtype=np.uint64)
        c_f = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, n_paths, n_steps,
                               noise_mode=_ffi.HH_NOISE_GENERATE, seeds=seeds_h)
        r_full, _, _ = orc.mc_solve(model, c_f, want_terminal=False)
        reps, t0 = 0, time.perf_counter()
        while True:
            r_full, _, _ = orc.mc_solve(model, c_f, want_terminal=False)
            reps += 1
            el = time.perf_counter() - t0
            if el >= args.cpu_seconds or reps >= 50:
                break
        r_gen = _ffi.hh_result()
        ctx.check(lib.hh_mc_solve(h, C.byref(model), C.byref(c_f), C.byref(r_gen), None))
        out["price_check"]["full_config_generate"] = {
            "paths": n_paths, "gpu": r_gen.price, "cpu_ref": r_full.price,
            "rel_err": abs(r_gen.price - r_full.price) / abs(r_full.price),
            "what": "the whole 10^6 x 252 configuration, same seeds through both Philox / Box-Muller implementations"}
        out["cpu_baseline"] = {
            "value": reps * float(n_paths) * n_steps / el, "unit": "path-steps/s", "cores": threads,
            "kind": "port",
            "sample": "%d x (the full configuration: %d paths x %d steps, GENERATE from per-trajectory seeds), %.1f s, "
                      "oracle/hh_oracle.c with OpenMP over paths" % (reps, n_paths, n_steps, el)}
    for c_ in list(_ffi._contexts.values()):  # nothing of the library is left for the interpreter's exit to find
        c_.synchronize()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
