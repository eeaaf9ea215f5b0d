"""The iterate probe of julia/parity_replay.jl (`bk_root_probe`: every abscissa the reference's inverse_cdf,
sample_from_cf.jl:105-135, asks of its CDF) and the logic that reads it (tools/check_reference_replay.py): on the
committed self-test file sets — written by the CPU restatement in EACH of its eight readings of Roots.jl's two
find_zero calls, tests/golden/make_root_probe_selftest.py — the checker must name the reading that wrote a set, for
each of the three forks the sample exercises, and say "either" where it does not.  CPU only: this is the part of the
exchange a Julia host's ONE run will go through."""
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = os.path.join(ROOT, "tests", "golden", "root_probe_selftest")


@pytest.fixture(scope="module")
def checker():
    spec = importlib.util.spec_from_file_location("check_reference_replay", os.path.join(ROOT, "tools", "check_reference_replay.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cases():
    return json.load(open(os.path.join(BASE, "manifest.json")))["cases"]


def test_the_exchange_format_is_what_the_julia_script_writes():
    src = open(os.path.join(ROOT, "julia", "parity_replay.jl")).read()
    for cs in cases():
        n = cs["n"]
        counts = np.fromfile(os.path.join(BASE, cs["counts"]), dtype="<i4")
        assert counts.size == n and (counts >= 2).all()
        for key in ("VT", "u", "initial_guess", "max_guess", "h", "sol"):
            assert np.fromfile(os.path.join(BASE, cs[key]), dtype="<f8").size == n, key
            assert f"{key} = wbin(" in src, key   # the same keys in the Julia export
        assert np.fromfile(os.path.join(BASE, cs["xs"]), dtype="<f8").size == counts.sum()
        assert "counts = wbin(" in src and "xs = wbin(" in src and 'kind = "bk_root_probe"' in src
        # every first search starts from Roots' two points: x0 + dx, then x0
        xs = np.fromfile(os.path.join(BASE, cs["xs"]), dtype="<f8")
        g0 = np.fromfile(os.path.join(BASE, cs["initial_guess"]), dtype="<f8")
        off = np.concatenate([[0], np.cumsum(counts)])
        assert np.array_equal(xs[off[:-1] + 1], g0) and (xs[off[:-1]] > g0).all()


@pytest.mark.parametrize("cs", cases(), ids=lambda c: c["name"])
def test_the_checker_names_the_reading_that_wrote_the_file_set(checker, cs):
    w = cs["written_with"]
    ok, info = checker.check_root_probe(None, BASE, cs)
    assert ok, info
    r, hit_s, hit_l = checker.ROOT_VERDICT[cs["name"]]
    assert r["n_ladder"] >= 4  # the tail uniforms reach the ladder
    # the first search: the root form always; the caps where a trajectory runs into them (Order2 does, the secant not)
    assert {k[0] for k in hit_s} == {w["root_form"]}
    if w["root_form"] == 1:
        assert {k[1] for k in hit_s} == {w["caps"]}
    else:
        assert {k[1] for k in hit_s} == {0, 1}
    # the bisection: its form always; neither cap is reached (12 / 62 iterations against 100 / none)
    assert {k[0] for k in hit_l} == {w["bracket_form"]} and {k[1] for k in hit_l} == {0, 1}
    line = checker.root_verdict_line(cs["name"])
    assert line.startswith("VERDICT bk_root_form = %d" % w["root_form"]) and "bk_bracket_form = %d" % w["bracket_form"] in line


def test_bit_pattern_midpoint():
    from oracle import bk_oracle as B
    assert B.roots_middle(0.0, 1.0) == np.uint64(0x3FF0000000000000 >> 1).view(np.float64)
    assert B.roots_middle(1.0, 2.0) == 1.5 and B.roots_middle(1.0, 4.0) == 2.0  # halfway in the exponent, then the mantissa
    a = 0.25
    assert B.roots_middle(a, np.nextafter(a, 1.0)) == a  # adjacent floats: nothing between them
    assert B.roots_middle(-1.0, 1.0) == 0.0
