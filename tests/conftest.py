import importlib.util
import os
import sys

import pytest

try:  # torch bundles its own HIP runtime: it has to be the first one loaded in the process
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: asserts round counts or seconds, not bytes -- collected after every parity test")


# The two full-size configurations run first: 32 GiB of input + a 34 GB image (and 16 GiB + its decode) want the
# process's memory as a fresh process has it -- behind 250 other tests the same 32 GiB test took 116 s instead of 74.
FULL_SIZE_FIRST = ("test_cfg5_full_32gib_random", "test_roundtrip_full_size_cfg3_headline", "test_cfg4_full_10gib_zstd_round_trip")


def pytest_collection_modifyitems(config, items):
    first = [it for name in FULL_SIZE_FIRST for it in items if it.name == name]
    if first:
        rest = [it for it in items if it not in first]
        items[:] = first + rest
    # assertions about counts and seconds go last: under -x they can never hide a parity test
    perf = [it for it in items if it.get_closest_marker("perf")]
    if perf:
        items[:] = [it for it in items if it not in perf] + perf


def load_bindings():
    """The product package directory is `lrzip-next_amd/` (not an importable identifier)."""
    name = "lrzip_next_amd_bindings"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "lrzip-next_amd", "bindings.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def B():
    return load_bindings()


@pytest.fixture(scope="session")
def O():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib
