"""Profiling hygiene (tools/run_steps.sh): a step that dies on a signal is waved through ONLY when its log carries the
one exit-time fault this image is known for (profiles/r04_b_coop_exit_sigsegv.txt: hsa_shut_down after a cooperative
launch under rocprofv3, nine fixed return addresses) — any other fault stops the list instead of hiding behind it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOWN = ["59e", "e63", "31d", "d67", "01d", "cee", "c5a", "fce", "097"]


def frames(suffixes):
    return "".join("    @     0x769213b%s%s (unknown)\\n" % ("%02x" % i, s) for i, s in enumerate(suffixes))


def run(tmp_path, steps):
    (tmp_path / "steps.txt").write_text("\n".join(steps) + "\n")
    with open(tmp_path / "steps.txt") as f:
        p = subprocess.run([os.path.join(ROOT, "tools", "run_steps.sh"), str(tmp_path), "20"], stdin=f,
                           capture_output=True, text=True, timeout=120)
    return p.returncode, (tmp_path / "steps.log").read_text()


def test_known_exit_fault_is_accepted_and_the_list_goes_on(tmp_path):
    rc, log = run(tmp_path, ["a|echo fine", "b|printf '%s'; kill -SEGV $$" % frames(KNOWN), "c|echo reached"])
    assert rc == 0 and "known exit-time signature" in log and "== c:" in log


def test_any_other_fault_stops_the_list(tmp_path):
    other = list(KNOWN)
    other[4] = "abc"  # one frame differs: another stack
    rc, log = run(tmp_path, ["a|printf '%s'; kill -SEGV $$" % frames(other), "b|echo not reached"])
    assert rc == 1 and "WITHOUT the known exit-time signature" in log and "== b:" not in log
    rc, log = run(tmp_path, ["a|kill -SEGV $$", "b|echo not reached"])  # no stack at all
    assert rc == 1 and "== b:" not in log
    rc, log = run(tmp_path, ["a|kill -ABRT $$", "b|echo not reached"])
    assert rc == 1 and "== b:" not in log


def test_a_timeout_stops_the_list(tmp_path):
    (tmp_path / "steps.txt").write_text("a|sleep 30\nb|echo not reached\n")
    with open(tmp_path / "steps.txt") as f:
        p = subprocess.run([os.path.join(ROOT, "tools", "run_steps.sh"), str(tmp_path), "1"], stdin=f,
                           capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "killed by its timeout" in (tmp_path / "steps.log").read_text()
