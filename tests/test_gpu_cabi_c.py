"""A plain-C client of the C-ABI (tests/c/cabi_smoke.c), built with gcc against
include/hedgehog_mc.h and run with NO Python/torch in the process — what a foreign host such as
Julia's ccall sees."""
import os
import subprocess

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_client(tmp_path):
    libdir = os.path.join(ROOT, "hedgehog.jl_amd", "lib")
    exe = str(tmp_path / "cabi_smoke")
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-Werror",
                    os.path.join(ROOT, "tests", "c", "cabi_smoke.c"), "-o", exe,
                    "-L" + libdir, "-lhedgehog_mc", "-lm",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    tag, price, se = p.stdout.strip().splitlines()[-1].split()  # RCCL may print its banner before
    assert tag == "OK" and 8.5 < float(price) < 10.0 and 0 < float(se) < 0.1
