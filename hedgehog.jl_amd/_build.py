"""Builds lib/libhedgehog_mc.so from csrc/ with hipcc for gfx950 (in-tree; no JIT cache)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libhedgehog_mc.so")
SOURCES = ["hh_api.hip", "hh_kernels.hip", "hh_bk.hip", "hh_lsm.hip", "hh_fourier.hip"]
FLAGS = ["-shared", "-fPIC", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "hedgehog_mc.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, extra_flags=()) -> str:
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [_hipcc(), *FLAGS, *extra_flags, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True))
