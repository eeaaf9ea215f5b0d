#!/usr/bin/env python3
"""Where a date of the persistent LSM induction spends its time: builds diagnostic variants of the
library (HH_LSM_DEBUG bits: 1 no waiting in the all-gather, 2 no solve, 4 no partial sums /
reductions / publish — their RESULTS ARE WRONG, only the clock is read) and times each on the same
ensembles in a child process.  GPU box only (hipcc builds here, ~40 s per variant)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "hedgehog.jl_amd", "_build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
sizes = sys.argv[1:] or ["10000", "1000000"]
vdir = os.path.join(ROOT, "hedgehog.jl_amd", "lib", "variants")
os.makedirs(vdir, exist_ok=True)
for bits in (0, 1, 3, 7, 2, 4):
    lib = os.path.join(vdir, f"libhh_lsmdbg{bits}.so")
    cmd = [b._hipcc(), *b.FLAGS, f"-DHH_LSM_DEBUG={bits}", *[os.path.join(b.CSRC, s) for s in b.SOURCES],
           "-o", lib]
    subprocess.run(cmd, check=True, capture_output=True)
    env = dict(os.environ, HEDGEHOG_MC_LIB=lib)
    code = ("import sys; sys.argv=['x',%s]; exec(open('%s').read())"
            % (",".join(repr(s) for s in sizes), os.path.join(ROOT, "tools", "lsm_latency.py")))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout
    print(f"== HH_LSM_DEBUG={bits}")
    print("\n".join(ln.split(" | per date")[0] for ln in out.splitlines() if ln.startswith("n=")), flush=True)
