#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in the shipped gfx950 code objects, read from the
AMDGPU metadata notes of lib/obj/<tag>/*.o (no GPU needed).
usage: python tools/kernel_regs.py [tag] [name-filter …]   -> one line per kernel"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "a.co"), os.path.join(td, "a.fat")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True)
        if r.returncode or not os.path.exists(fat):
            return []
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                           capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            return []
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s*- (?=\.agpr_count:)", notes):  # a kernel's map starts with its first key
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or ".vgpr_count" not in blk:
            continue
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) if re.search(rf"\.{k}:\s+(\d+)", blk) else 0  # noqa: E731
        out.append(dict(name=name.group(1), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"),
                        spill=g("vgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                        lds=g("group_segment_fixed_size")))
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    filt = sys.argv[2:]
    objdir = os.path.join(ROOT, "hedgehog.jl_amd", "lib", "obj", tag)
    rows = []
    for f in sorted(os.listdir(objdir)):
        if f.endswith(".o"):
            for k in kernels_of(os.path.join(objdir, f)):
                rows.append((f, k))
    names = demangle([k["name"] for _, k in rows])
    print(f"{'object':14s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'spill':>5s} {'scratch':>7s} {'lds':>6s}  kernel")
    for (f, k), nm in zip(rows, names):
        nm = re.sub(r"^void ", "", nm)
        if filt and not any(s in nm for s in filt):
            continue
        print(f"{f:14s} {k['vgpr']:4d} {k['agpr']:4d} {k['sgpr']:4d} {k['spill']:5d} {k['scratch']:7d} {k['lds']:6d}  {nm[:150]}")


if __name__ == "__main__":
    main()
