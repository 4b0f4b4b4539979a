import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def load_pkg():
    """The product package lives in a directory whose name is not a Python identifier."""
    import importlib.util
    name = "rust_brotli_decompressor_amd"
    if name in sys.modules:
        return sys.modules[name]
    # torch ships its own copy of the HIP runtime; when it is going to be used in this process it has to be the
    # first one to initialise (a second runtime initialised later reports "No HIP GPUs are available")
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    path = os.path.join(ROOT, "rust-brotli-decompressor_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()
