"""hedgehog.jl_amd/csrc/hh_rng.h compiled for the HOST (hipcc, host side only — no HIP call is made, no GPU needed):
the Philox4x32-10 the kernels draw with against the Random123 known-answer vectors, and the ends of
the uniform built from its bits.  The oracle carries a Philox of its own (oracle/hh_oracle.c): two independent
implementations held to the same published vectors."""
import os
import shutil
import subprocess

import pytest

from tests.conftest import SANITIZE, host_env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_philox_known_answers_on_the_host(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "rng_check"
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize"] if SANITIZE else ["-O2"]
    subprocess.run([hipcc, *san, "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "hedgehog.jl_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "rng_check.hip"), "-o", str(exe)],
                   check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True, env=host_env())
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count(" ok") == 4 and "MISMATCH" not in p.stdout
