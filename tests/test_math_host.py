"""hedgehog.jl_amd/csrc/hh_math.h (the range-specialised sincos / log / atan2 of the Broadie–Kaya
kernels) compiled for the HOST with g++ and checked against 80-bit libm on 2·10^6 random arguments
per function: the approximations themselves (reduction constants, polynomial coefficients, quadrant
logic) are the same source the device compiles; only the reciprocal differs (division here, hardware
reciprocal + two Newton steps there)."""
import os
import shutil
import subprocess

import pytest

from tests.conftest import host_cxxflags, host_env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_hh_math_against_long_double_libm(tmp_path):
    exe = tmp_path / "math_check"
    subprocess.run(["g++", *host_cxxflags(), "-std=c++17", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "hedgehog.jl_amd", "csrc"),
                    os.path.join(ROOT, "tests", "c", "math_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, env=host_env()).stdout
    err = {ln.split()[0]: float(ln.split()[2]) for ln in out.strip().splitlines()}
    assert set(err) == {"sin", "cos", "log", "atan2", "exp", "wsin", "wcos", "nquant"}
    # exp <= 1.5 ulp; the wide sincos (|x| <= 2^45, three-term reduction) as good as the narrow one
    assert err["exp"] < 1.5 and err["wsin"] < 2.0 and err["wcos"] < 2.0, err
    # ulp of the fp64 result (sin/cos: absolute 2^-73 where the value is below 1e-6)
    assert err["sin"] < 2.0 and err["cos"] < 2.0 and err["log"] < 2.5 and err["atan2"] < 2.5, err
    # the normal quantile (AS 241): a rational approximation good to 1e-16 before rounding; it seeds a root search
    assert err["nquant"] < 8.0, err
