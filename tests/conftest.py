import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/."""
    from tests import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def hip_lib():
    """The product's C-ABI library; GPU tests must run the HIP path, never a fallback."""
    from chameleonrt_amd import core
    L = core.load()
    assert L.crt_hip_device_count() > 0, "GPU test without a HIP device"
    return L
