"""Builds lib/libhedgehog_mc.so from csrc/ with hipcc for gfx950 (in-tree; no JIT cache).

One object per translation unit (compiled in parallel, re-made only when its source or a header is
newer), then one link."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libhedgehog_mc.so")
OBJ = os.path.join(HERE, "lib", "obj")
SOURCES = ["hh_api.hip", "hh_mgpu.hip", "hh_kernels.hip", "hh_multi.hip", "hh_bk.hip", "hh_lsm.hip", "hh_fourier.hip"]
# (source, object, extra flags): hh_bk.hip is built without the machine-code LICM pass — see the head of that file
UNITS = [(s, s.replace(".hip", ".o"), ["-mllvm", "-disable-machine-licm"] if s == "hh_bk.hip" else []) for s in SOURCES]
CFLAGS = ["-fPIC", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wall",
          "-Wno-unused-function"]
LDFLAGS = ["-shared", "-fPIC", "--offload-arch=gfx950", "-ldl"]  # RCCL is bound with dlopen (hh_mgpu.hip)


def _hipcc() -> str:
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _headers():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "hedgehog_mc.h"))
    return deps


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
    return proc.stdout + proc.stderr


def build_library(force: bool = False, extra_flags=(), out: str | None = None) -> str:
    """extra_flags (-D… variants for A/B builds) go to every compile; they get their own object
    directory so that the default build's objects stay valid."""
    lib = out or LIB
    if not force and not extra_flags and out is None and not is_stale():
        return lib
    tag = "default" if not extra_flags else "v_" + "_".join(
        f.replace("-D", "").replace("=", "-") for f in extra_flags)
    objdir = os.path.join(OBJ, tag)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    jobs, objs = [], []
    for s, o, unit_flags in UNITS:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, o)
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([_hipcc(), "-c", *CFLAGS, *unit_flags, *extra_flags, src, "-o", obj])
    warnings = ""
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 7)) as ex:
            warnings = "".join(ex.map(_run, jobs))
    _run([_hipcc(), *LDFLAGS, *objs, "-o", lib])
    if warnings.strip():
        print(warnings)
    return lib


if __name__ == "__main__":
    print(build_library(force=True))
