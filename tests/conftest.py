import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# HH_SANITIZE=1 (make test-cpu-asan): the host-compiled checks are built with AddressSanitizer + UBSan.  Their
# executables run WITHOUT the preloaded libasan of the python that started them (they link their own).
SANITIZE = os.environ.get("HH_SANITIZE") == "1"


def host_cxxflags():
    if SANITIZE:
        return ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    return ["-O2"]


def host_env():
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    return env


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Make sure the in-tree HIP library exists (hipcc cross-compiles gfx950 without a GPU); a
    stale or missing .so is rebuilt here exactly as __graft_entry__.build() does."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_hh_build", os.path.join(ROOT, "hedgehog.jl_amd", "_build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        mod.build_library()
    except Exception as e:  # no hipcc on this host: keep whatever prebuilt .so travelled here
        if not os.path.exists(mod.LIB):
            pytest.exit(f"libhedgehog_mc.so missing and cannot be built: {e}", returncode=3)


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_ffi
    return oracle_ffi.load()


@pytest.fixture(scope="session")
def hhlib():
    """The product C-ABI (HIP). GPU tests only."""
    import hedgehog_jl_amd as hh
    return hh.get_context(0)
