#!/usr/bin/env python3
"""Builds an A/B variant of the library: tools/build_variant.py <tag> [-DFLAG=V ...] ->
hedgehog.jl_amd/lib/variants/libhh_bk_<tag>.so (what tools/bk_ab.py loads beside the shipped build)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "hedgehog.jl_amd", "_build.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
tag, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "hedgehog.jl_amd", "lib", "variants", f"libhh_bk_{tag}.so")
print(m.build_library(extra_flags=tuple(flags), out=out))
