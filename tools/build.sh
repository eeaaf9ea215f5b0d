#!/bin/bash
# rebuild lib/libhedgehog_mc.so (stale objects only) — what __graft_entry__.build() does for the product
cd "$(dirname "$0")/.." && python - <<'PY'
import importlib.util
spec = importlib.util.spec_from_file_location('b', 'hedgehog.jl_amd/_build.py')
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); print(m.build_library())
PY
