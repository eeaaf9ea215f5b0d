"""Builds tests/c/libstub_rccl.so, the test-only stand-in for librccl that the multi-rank tests of hh_mgpu
bind on a one-GPU box ($HEDGEHOG_MC_RCCL).  No pytest here: __graft_entry__.build() calls this too, and a
product build must not depend on the test runner."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
STUB_SRC = os.path.join(HERE, "stub_rccl.hip")
STUB = os.path.join(HERE, "libstub_rccl.so")


def build_stub():
    if os.path.exists(STUB) and os.path.getmtime(STUB) >= os.path.getmtime(STUB_SRC):
        return STUB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-std=c++17", "--offload-arch=gfx950", STUB_SRC, "-o", STUB],
                   check=True)
    return STUB
