"""hedgehog.jl_amd/csrc/hh_bessel.h — the complex I_ν(z) of the Broadie–Kaya characteristic function
(the reference calls SpecialFunctions.besseli = AMOS, heston.jl:184-212) — compiled for the HOST with
g++ and checked against 40-digit mpmath on a sweep of orders and arguments covering the three
evaluation regimes (ascending series, Hankel expansion, base order + ratio recurrence), both half
planes, and the seams between them.  The same source compiles for the device; there only the
reciprocal differs (hardware rcp + two Newton steps) and a wave's lanes share the longest loop."""
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")
from tests.conftest import host_cxxflags, host_env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = tmp_path / "bessel_check"
    subprocess.run(["g++", *host_cxxflags(), "-std=c++17", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "hedgehog.jl_amd", "csrc"),
                    os.path.join(ROOT, "tests", "c", "bessel_check.cpp"), "-o", str(exe)], check=True)
    return str(exe)


def _cases():
    rng = np.random.default_rng(7)
    out = []
    # ν of the parameter sets in use (H252: 0.777…; Feller-satisfied sets: 1 … 20) and the edges
    for nu in (-0.95, -0.5, -0.2, 0.0, 0.5, 7.0 / 9.0, 0.99, 1.0, 1.5, 3.0, 7.3, 15.0, 19.0, 40.0, 63.0):
        rh = max(13.0, nu * nu / 6.0 + 13.0)                  # Hankel at order ν from here (hh_bessel.h)
        radii = np.concatenate([10.0 ** rng.uniform(-2, np.log10(300.0), 36), rng.uniform(13.0, rh + 1.0, 14),
                                [12.9, 12.999, 13.0, 13.001, 13.4, 15.4, 15.6, 18.4, 18.6, 19.5,
                                 rh - 0.01, rh + 0.01, 1.5 * rh]])
        for r in radii:
            # angles on both sides of the series criterion |z| - Re z <= 14 of the in-between range
            edge = math.acos(max(-1.0, 1.0 - 14.0 / r)) if r > 7.0 else 3.0
            for ang in (0.0, 0.3, 1.2, math.pi / 2 - 1e-3, math.pi / 2 + 1e-3, 2.5, -0.7, -2.9,
                        min(edge, 1.5) - 0.01, min(edge, 1.5) + 0.01):
                out.append((float(nu), float(r * math.cos(ang)), float(r * math.sin(ang))))
    return out


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_besseli_against_mpmath(tmp_path):
    exe = _build(tmp_path)
    cases = _cases()
    text = "".join(f"{nu!r} {re!r} {im!r}\n" for nu, re, im in cases)
    out = subprocess.run([exe], input=text, check=True, capture_output=True, text=True, env=host_env()).stdout.split("\n")
    assert "table-bound-violated" not in out
    mp.mp.dps = 40
    worst = 0.0
    for (nu, re, im), line in zip(cases, out):
        lre, lim = map(float, line.split())
        want = mp.besseli(nu, mp.mpc(re, im))
        got = mp.exp(mp.mpc(lre, lim))
        if abs(want) > mp.mpf(10) ** 300 or abs(want) < mp.mpf(10) ** -300:
            want_l = mp.log(want)
            err = float(abs(mp.exp(mp.mpc(lre, lim) - want_l) - 1))
        else:
            err = float(abs(got - want) / abs(want))
        # The ascending series is accurate to rounding of its LARGEST terms, I_ν(|z|) in size: relative to
        # the result that is the cancellation I_ν(|z|)/|I_ν(z)|, which the dispatch keeps below e^14 (and
        # which is what the characteristic function needs: it divides by I_ν(ν_κ) >= I_ν(|ν_γ|)).  The
        # Hankel sums lose what their two exponentials cancel next to the imaginary axis (J-like zeros).
        r = math.hypot(re, im)
        near_axis = abs(abs(math.atan2(im, re)) - math.pi / 2) < 0.5
        rh = max(13.0, nu * nu / 6.0 + 13.0)
        if r < 13.0 or (nu >= 1.0 and r < rh and (r - abs(re) <= 14.0 or im * im <= 28.0 * (nu + 1.0))):
            loss = float(mp.besseli(nu, r) / abs(want)) if abs(want) > 0 else 1.0
            assert loss < 3e6 or near_axis, (nu, re, im, loss)  # e^14.9: the estimate holds away from the J-like zeros
            bar = max(2e-11, 4e-15 * loss, 1e-15 * abs(lre))
        else:
            bar = 5e-10 if near_axis else 2e-11
        assert err < bar, (nu, re, im, err)
        worst = max(worst, err)
    assert worst > 0.0


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_series_term_bound_holds_for_every_order(tmp_path):
    """bessel_table() finds, for its ν, a series length n0 + n1·|z| that leaves out only terms below
    2^-57 of the largest one on the whole series range, within the table size; swept over ν in (-1, 300]."""
    exe = _build(tmp_path)
    nus = np.concatenate([-1.0 + 10.0 ** np.linspace(-4, 0, 40), np.linspace(0.0, 60.0, 121), [100.0, 200.0, 300.0]])
    text = "".join(f"{float(nu)!r} 1.0 0.5\n" for nu in nus)
    out = subprocess.run([exe], input=text, capture_output=True, text=True)
    assert out.returncode == 0 and "table-bound-violated" not in out.stdout
