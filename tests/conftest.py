import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    import raytracer_glsl_amd
    return raytracer_glsl_amd


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement (test infrastructure only)."""
    from oracle.oracle import CpuOracle
    return CpuOracle()


@pytest.fixture(scope="session")
def gpu_lib(rt):
    """The product library on a real device; fails loudly when it is missing."""
    lib = rt.host.load_library()
    return lib
