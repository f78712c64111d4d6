"""pytest configuration: `gpu` marker, import paths for the product binding and the oracle."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "grace-devel_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O  # oracle/oracle.py -- the checker, test infrastructure only
    O.build()
    return O


@pytest.fixture(scope="session")
def gh():
    """The product binding (ctypes over libgrace_hip.so).  Fails loudly if not built."""
    import grace_hip
    return grace_hip


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a GPU")
    return torch.device("cuda:0")
