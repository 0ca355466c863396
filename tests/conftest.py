import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("raytracing-rust_amd")


@pytest.fixture(scope="session")
def abi(pkg):
    return pkg.abi


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hb():
    """The HIP back end binding; the library must exist (built by __graft_entry__.build())."""
    mod = importlib.import_module("raytracing-rust_amd.hip_backend")
    if not os.path.exists(mod.LIB_PATH):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "raytracing-rust_amd", "csrc"), "-s"], check=True)
    return mod


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
