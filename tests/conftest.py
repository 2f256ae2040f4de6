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
