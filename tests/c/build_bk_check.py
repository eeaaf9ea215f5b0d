"""Builds tests/c/libhh_bk_check.so: the product library with two Broadie–Kaya code paths switched back to the forms
they replaced — the trajectory's real-axis evaluations through the complex code (HH_BK_COMPLEX_SETUP), the ladder
kernel searching the prefix sums in place (HH_BK_LADDER_SPAN = 1).  tests/test_gpu_bk_forms.py holds the shipped build
to this one bit for bit.  No pytest here: __graft_entry__.build() calls this too (non-fatally), so that the file
travels to the GPU box with the snapshot."""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libhh_bk_check.so")
FLAGS = ("-DHH_BK_COMPLEX_SETUP=1", "-DHH_BK_LADDER_SPAN=1")


def build_bk_check():
    spec = importlib.util.spec_from_file_location("_hh_build", os.path.join(ROOT, "hedgehog.jl_amd", "_build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    deps = [os.path.join(mod.CSRC, s) for s in mod.SOURCES] + mod._headers()
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    return mod.build_library(extra_flags=FLAGS, out=OUT)


if __name__ == "__main__":
    print(build_bk_check())
