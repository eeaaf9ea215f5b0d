"""fm::sqrt_lean on the DEVICE (hedgehog.jl_amd/csrc/hh_math.h): the contract its header states, pinned.
The routine replaces sqrt() in cabs / csqrt of the Broadie–Kaya CF arithmetic (heston.jl:184-212) and drops the
library routine's range scaling and class test; what that changes at the edges must be what the header says."""
import os
import shutil
import struct
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def f(bits):
    return struct.unpack("<d", struct.pack("<Q", bits))[0]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_sqrt_lean_contract(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "math_device_check")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "hedgehog.jl_amd", "csrc"),
                    os.path.join(ROOT, "tests", "c", "math_device_check.hip"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    rows = {}
    worst = None
    for ln in out.splitlines():
        p = ln.split()
        if p[0] == "arg":
            rows[int(p[1], 16)] = (int(p[3], 16), int(p[5], 16))
        elif p[0] == "random_worst_ulp":
            worst = int(p[1])
    import math
    # the domain the callers use — zero, and everything from 2^-767 up to the largest finite double: sqrt()'s bits
    assert worst == 0
    for w in (0.0, 2.0**-767, 1.5 * 2.0**-767, 1e-200, 0.25, 1.0, 2.0, 3.0, 1e300, 1.7976931348623157e308):
        lean, ref = rows[struct.unpack("<Q", struct.pack("<d", w))[0]]
        assert lean == ref, w
    # below 2^-767 (the header: "merely less accurate, never NaN"): finite, non-negative, within 1e-3 relative
    for w in (4.9406564584124654e-324, 1e-310, 2.0**-1022, 2.0**-768):
        lean, ref = rows[struct.unpack("<Q", struct.pack("<d", w))[0]]
        assert math.isfinite(f(lean)) and f(lean) >= 0.0 and abs(f(lean) - f(ref)) <= 1e-3 * f(ref) + 1e-200, w
    # outside the contract (w finite, >= 0), stated in the header: +inf -> NaN (0 x inf), NaN -> NaN, and a negative
    # argument gives no finite positive number a caller could mistake for a modulus
    lean_inf, _ = rows[0x7FF0000000000000]
    lean_nan, _ = rows[0x7FF8000000000000]
    assert math.isnan(f(lean_inf)) and math.isnan(f(lean_nan))
    assert math.isnan(f(rows[0x8000000000000000][0]))  # -0.0: rsq = -inf passes the cap, -0 x -inf (sqrt gives -0.0)
    for neg in (-1.0, -1e-300):
        v = f(rows[struct.unpack("<Q", struct.pack("<d", neg))[0]][0])
        assert not (math.isfinite(v) and v > 0.0), (neg, v)
