"""hh_mgpu's RCCL branch with G = 3 ranks, its AUTO-mode fall-back after a collective that fails for
one rank, and the status HH_MGPU_RCCL returns instead — on ONE GPU, through a test-only stand-in for
librccl (tests/c/stub_rccl.hip; see tests/rccl_stub_worker.py for the scenarios).  What the driver's
8-GPU node runs with the real RCCL (montecarlo.jl:478-493 stays one call)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tests.c.build_stub import STUB, build_stub  # noqa: E402  (pytest-free: the product's build() uses it too)


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    build_stub()
    out = str(tmp_path_factory.mktemp("rccl_stub") / "out.json")
    env = dict(os.environ, HEDGEHOG_MC_RCCL=STUB)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_stub_worker.py"), out], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.load(open(out))


def test_three_ranks_through_the_rccl_branch_equal_the_host_ordered_sum(run):
    assert run["auto_mode_is_rccl"]
    assert run["european_bit_equal"]  # Σ, Σ², three partials, every terminal sample
    assert run["basket_bit_equal"]
    assert run["lsm_bit_equal"] and run["lsm_heston_bit_equal"]  # price, std error, stopping times and values
    assert run["groups"] == run["groups_expected"]  # ONE all-reduce per solve; 2 + (steps - 1) + 1 per LSM solve
    assert run["calls"] == 3 * run["groups"]


def test_rank_count_comes_from_a_collective_and_the_library_is_named(run):
    """What bench.py prints next to "reduce: rccl": ranks counted by an all-reduce of ones through the solve's own
    exchange, the file the entry points were bound from, its version, and that the environment chose it."""
    assert run["selftest"] == [3, 1]            # three ranks, RCCL branch
    assert run["selftest_host_context"] == [3, 0]
    info = run["rccl_info"]
    assert info["library"].endswith("libstub_rccl.so") and info["version"] == 9990000
    assert info["from_env"] is True and info["usable"] is True


def test_several_models_through_the_rccl_branch(run):
    assert run["multi_bit_equal"] and run["multi_groups"] == 1


def test_a_shard_that_cannot_leave_its_stream_makes_the_context_refuse(run):
    assert run["stuck_code"] == -5 and run["stuck_second_code"] == -5
    assert "create a new context" in run["stuck_text"] or "create a new one" in run["stuck_text"]
    assert run["stuck_seconds"] < 2.5  # the stand-in's orphan would take three seconds


def test_serial_and_threaded_enqueue_agree(run):
    assert run["serial_enqueue_bit_equal"] and run["enqueue_stats_ok"]


def test_auto_mode_finishes_on_the_host_after_a_rank_failure_without_waiting_for_the_orphan(run):
    assert run["auto_failure_result_bit_equal"]
    assert run["auto_failure_orphans"] == 1  # rank 0 was enqueued before rank 1 refused
    assert run["auto_failure_aborts"] == 3
    # the orphan leaves after 3 s by itself: a library that synchronised its stream first would take that long
    assert run["auto_failure_seconds"] < 1.5
    assert run["auto_failure_mode_is_host"]
    assert "aborted" in run["auto_failure_text"]
    assert run["auto_after_failure_bit_equal"]  # the context keeps working (fresh streams, host sum)
    assert run["basket_auto_failure_bit_equal"]


def test_required_rccl_returns_a_status_not_a_hang(run):
    assert run["strict_first_solve_bit_equal"]
    assert run["strict_failure_code"] == run["HH_ERR_RCCL"]
    assert run["strict_failure_seconds"] < 1.5
    assert run["strict_second_code"] == run["HH_ERR_RCCL"]  # HH_MGPU_RCCL does not fall back, ever
    assert run["strict_lsm_code"] == run["HH_ERR_RCCL"]


def test_eight_ranks(run):
    """The node's shape: eight communicators, seven worker threads; idle shards (7 trajectories over 8 ranks)."""
    assert run["eight_ranks_mode_is_rccl"]
    assert run["eight_ranks_bit_equal"] and run["eight_ranks_lsm_bit_equal"]
    assert run["eight_ranks_failure_bit_equal"] and run["eight_ranks_failure_seconds"] < 1.5


def test_lsm_induction_through_a_failing_exchange(run):
    assert run["lsm_auto_failure_bit_equal"]  # run again on the host's ordered sum
    assert run["lsm_auto_failure_orphans"] == 2
    assert run["lsm_auto_failure_seconds"] < 2.0
    assert run["lsm_auto_failure_mode_is_host"]
    assert run["lsm_strict_failure_code"] == run["HH_ERR_RCCL"]
    assert run["lsm_strict_failure_seconds"] < 1.5


def test_bench_single_process_form_through_the_rccl_branch():
    """What rank 0 of the driver's multi-GPU bench starts as its `single_process` block, rehearsed on one GPU:
    bench.py --single-process over three shards with the stand-in bound — the line must say RCCL carried them."""
    build_stub()
    env = dict(os.environ, HEDGEHOG_MC_RCCL=STUB)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--single-process",
                        "--devices", "0,0,0", "--steps", "3", "--warmup", "1", "--paths", "30000", "--ramp-ms", "0",
                        "--allow-rccl-override"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(next(ln for ln in p.stdout.splitlines() if ln.startswith("{")))
    assert line["single_process"] is True and line["n_gpus"] == 3
    assert line["reduce"] == "rccl" and line["rccl_ranks"] == 3 and line["ranks_counted_by_the_exchange"] == 3
    # … and WHO carried them: the line names the stand-in, so it cannot be taken for a real-RCCL record
    assert line["rccl_library"].endswith("libstub_rccl.so") and line["rccl_library_from_env"] is True
    assert line["rccl_version"] == 9990000
    assert len(line["per_rank_kernel_ms"]) == 3 and all(t and t > 0 for t in line["per_rank_kernel_ms"])
    assert len(line["enqueue_host_us"]["per_shard"]) == 3 and line["enqueue_host_us"]["phase"] > 0
    assert line["config"]["global_paths"] == 90000 and line["scaling"] == "weak"
    # without the explicit flag a set $HEDGEHOG_MC_RCCL is refused before anything runs
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--single-process",
                        "--devices", "0,0,0", "--steps", "1", "--warmup", "0", "--paths", "3000"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "allow-rccl-override" in p.stderr and not p.stdout.strip()
    assert abs(line["price"] - line["analytic_carr_madan"]) < 6 * line["std_error"] + 0.05  # Euler bias at dt = 1/252
    assert line["value"] > 0 and line["roofline"]["launches_timed"] == 3
