"""bench.py --gpus N must really run N ranks (the driver's scaling run depends on it).

CPU part: the launcher, the 127.0.0.1 rendezvous, the all-reduce that counts the ranks and the
shard ranges, rehearsed without a GPU (`--rehearse`, gloo).  GPU part: N > the box's GPUs fails
loudly; two ranks on the one GPU of the test box (gloo standing in for RCCL, which refuses two
ranks per device) run the real kernels and price exactly what one process prices."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(*flags, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items()
         if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH, *flags], env=e, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


@pytest.mark.timeout(300)
def test_launcher_starts_n_ranks_weak():
    p, out = run_bench("--gpus", "2", "--rehearse")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len([ln for ln in p.stdout.splitlines() if ln.startswith("{")]) == 1  # rank 0 only
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "weak"
    assert out["global_paths"] == 2_000_000 and out["paths_covered"] == 2_000_000


def test_the_metric_string_names_the_workload_that_ran():
    import bench
    assert bench.metric_string(1_000_000, 252) == "MC path-steps/sec (Heston Euler-Maruyama, 1e6 paths x 252 steps per GPU)"
    assert "125000 paths x 252 steps per GPU" in bench.metric_string(125_000, 252)
    assert "1e7 paths in all x 100 steps" in bench.metric_string(0, 100, True, 10_000_000)


def test_predicted_scaling_arithmetic():
    """predicted_scaling of the N = 1 line: from each shard's time ALONE and one rank's per-solve overhead."""
    import bench
    rows = bench.predict_scaling({1: 8.0, 2: 4.0, 4: 2.0, 8: 1.0}, 0.0)  # perfectly divisible work, free exchange
    assert all(abs(r["efficiency_pipelined"] - 1.0) < 1e-15 and abs(r["efficiency_single_solve"] - 1.0) < 1e-15
               for r in rows.values())
    rows = bench.predict_scaling({1: 8.0, 2: 4.0, 4: 2.0, 8: 1.0}, 0.5)
    # one solve, call to result: (8 + .5) / (8 (1 + .5)); back-to-back solves hide the exchange while it is the shorter
    assert rows[8]["efficiency_single_solve"] == pytest.approx(8.5 / 12.0)
    assert rows[8]["efficiency_pipelined"] == pytest.approx(1.0) and rows[1]["efficiency_single_solve"] == 1.0
    rows = bench.predict_scaling({1: 8.0, 8: 1.0}, 2.0)  # the exchange longer than the shard: it sets the step
    assert rows[8]["efficiency_pipelined"] == pytest.approx(8.0 / (8 * 2.0))
    rows = bench.predict_scaling({1: 0.58, 2: 0.31, 4: 0.18, 8: 0.11}, 0.02)  # a shard that does not fill the chip
    assert rows[2]["efficiency_pipelined"] == pytest.approx(0.58 / 0.62)
    assert rows[8]["efficiency_pipelined"] < rows[4]["efficiency_pipelined"] < rows[2]["efficiency_pipelined"] < 1.0


def test_a_substitute_collective_library_is_refused_unless_asked_for():
    """$HEDGEHOG_MC_RCCL makes hh_mgpu bind another library in place of librccl (the tests' stand-in): bench.py
    must not produce a line under it silently — exit code 2 before anything runs, whatever the mode."""
    for flags in (("--gpus", "2", "--rehearse"), ("--gpus", "1", "--single-process"), ()):
        p, out = run_bench(*flags, env={"HEDGEHOG_MC_RCCL": "/nonexistent/libfake_rccl.so"}, timeout=120)
        assert p.returncode == 2 and out is None and "--allow-rccl-override" in p.stderr


@pytest.mark.timeout(300)
def test_launcher_strong_scaling_ranges_cover_the_ensemble():
    p, out = run_bench("--gpus", "3", "--rehearse", "--global-paths", "10000000")
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["n_gpus"] == 3 and out["rccl_ranks"] == 3 and out["scaling"] == "strong"
    assert out["paths_covered"] == 10_000_000 and out["shard_rank0"] == [0, 3_333_334]


@pytest.mark.timeout(300)
def test_rank_count_must_agree_with_the_flag():
    # as torch.distributed.run would start it, but with a --gpus that disagrees: loud, non-zero
    p, out = run_bench("--gpus", "2", "--rehearse",
                       env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                                MASTER_PORT="29611"))
    assert p.returncode != 0 and out is None
    assert "WORLD_SIZE" in p.stderr


@pytest.mark.timeout(300)
def test_more_ranks_than_gpus_is_refused():
    import torch
    have = torch.cuda.device_count()
    p, out = run_bench("--gpus", str(max(have, 1) + 1), "--steps", "2", "--warmup", "1")
    assert p.returncode == 2 and out is None
    assert "refusing" in p.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_on_this_gpu_price_what_one_process_prices(hhlib):
    from hedgehog_jl_amd import _ffi
    n = 100_000
    p, out = run_bench("--gpus", "2", "--backend", "gloo", "--devices", "0,0", "--paths", str(n),
                       "--steps", "3", "--warmup", "1", "--ramp-ms", "0", "--no-cpu-baseline")
    assert p.returncode == 0, p.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_paths"] == 2 * n
    # one process, the whole ensemble, increments drawn in-kernel from the same seeds
    m = _ffi.make_model()
    c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, 2 * n, 252,
                         seeds=np.arange(1, 2 * n + 1, dtype=np.uint64))
    r = _ffi.hh_result()
    hhlib.check(hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(m), C.byref(c), C.byref(r), None))
    assert out["price"] == pytest.approx(r.price, rel=1e-12)
    assert out["generate"]["price"] == pytest.approx(r.price, rel=1e-12)
    g = out["config5_greeks"]
    assert g["price"] == pytest.approx(r.price, rel=1e-12)
    assert g["greeks"][0] == pytest.approx(0.6557, abs=0.02)  # Δ of the H252 call
    s = out["strong_scaling"]
    assert s["global_paths"] == 10_000_000 and s["paths_this_rank"] == 5_000_000
    assert s["price"] == pytest.approx(9.2425, abs=0.05)
    # the same run from ONE process through hh_mgpu_solve_shards (device 0 listed twice: RCCL refuses
    # the duplicate, the library sums the two accumulator vectors on the host)
    sp = out["single_process"]
    assert "error" not in sp, sp
    assert sp["n_gpus"] == 2 and sp["reduce"] == "host ordered sum"
    assert sp["price"] == pytest.approx(r.price, rel=1e-12)
    assert sp["value"] > 0 and sp["value_cold"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_single_process_line_keeps_the_contract(hhlib):
    p, out = run_bench("--gpus", "1", "--single-process", "--steps", "5", "--warmup", "2", "--paths", "200000")
    assert p.returncode == 0, p.stderr[-3000:]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out
    assert out["single_process"] is True and out["n_gpus"] == 1 and out["steps"] == 5
    assert out["roofline"]["launches_timed"] == 5
    # one device: AUTO keeps the host path (no collective to run): the line must not claim RCCL ranks
    assert out["ranks_counted_by_the_exchange"] == 1 and out["rccl_library_from_env"] is False
    assert (out["reduce"] == "rccl") == (out["rccl_ranks"] == 1)
    assert len(out["per_rank_kernel_ms"]) == 1 and out["per_rank_kernel_ms"][0] > 0
    p, _ = run_bench("--gpus", "2", "--single-process", "--devices", "0,7", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0  # fewer GPUs than asked: refused, never a smaller run


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_one_rank_line_keeps_the_contract(hhlib):
    p, out = run_bench("--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                       "--no-extra")
    assert p.returncode == 0, p.stderr[-3000:]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["steps"] == 5
    assert len(out["per_rank_kernel_ms"]) == 1 and out["collective"]["ranks_counted_by_all_reduce_of_ones"] == 1
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert rf["launches_timed"] == 5


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
@pytest.mark.timeout(900)
def test_two_ranks_over_rccl_price_what_one_process_prices(hhlib):
    """The real thing (runs wherever the box has >= 2 GPUs; the single-GPU test box skips it): two
    ranks, one GPU each, the 16-double all-reduce over RCCL — weak and strong scaling lines."""
    from hedgehog_jl_amd import _ffi
    n = 100_000
    p, out = run_bench("--gpus", "2", "--paths", str(n), "--steps", "3", "--warmup", "1",
                       "--ramp-ms", "0", "--no-cpu-baseline")
    assert p.returncode == 0, p.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["backend"] == "rccl"
    m = _ffi.make_model()
    c = _ffi.make_config(_ffi.HH_HESTON, _ffi.HH_EULER_MARUYAMA, 2 * n, 252,
                         seeds=np.arange(1, 2 * n + 1, dtype=np.uint64))
    r = _ffi.hh_result()
    hhlib.check(hhlib.lib.hh_mc_solve(hhlib.handle, C.byref(m), C.byref(c), C.byref(r), None))
    assert out["price"] == pytest.approx(r.price, rel=1e-12)
    assert out["strong_scaling"]["paths_this_rank"] == 5_000_000
    sp = out["single_process"]  # one process, two devices, the library's own ncclAllReduce
    assert "error" not in sp, sp
    assert sp["reduce"] == "rccl" and sp["price"] == pytest.approx(r.price, rel=1e-12)
    p, out = run_bench("--gpus", "2", "--global-paths", str(2 * n), "--steps", "3", "--warmup", "1",
                       "--ramp-ms", "0", "--no-cpu-baseline", "--no-extra")
    assert p.returncode == 0 and out["scaling"] == "strong" and out["config"]["paths_per_gpu"] == n
    assert out["price"] == pytest.approx(r.price, rel=1e-12)
