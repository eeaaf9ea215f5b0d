"""Builds tests/c/libhh_bk_check.so: the product library with two Broadie–Kaya code paths in their plain forms — the
trajectory's real-axis evaluations through the complex code (HH_BK_COMPLEX_SETUP), and the bisection ladder of a
trajectory whose secant failed run by its own lane, statement for statement as sample_from_cf.jl:123-133 writes it
(HH_BK_SERIAL_LADDER; the shipped build walks the bisection tree with the whole wave).  tests/test_gpu_bk_forms.py
holds the shipped build to this one bit for bit; its session fixture calls this (a minute of hipcc where the file
is missing or older than the sources — `python tests/c/build_bk_check.py` makes it ahead of a GPU run, and the file
travels with the snapshot)."""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libhh_bk_check.so")
FLAGS = ("-DHH_BK_COMPLEX_SETUP=1", "-DHH_BK_SERIAL_LADDER=1")


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
