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
    # The sort's overflow memory (a context remembers that its last large sort overflowed and skips
    # the bucket sort for a while) would make which sort a test exercises depend on the tests before
    # it: off for the suite, on in the test that is about it.
    grace_hip.set_sort_overflow_hint(False)
    return grace_hip


@pytest.fixture(params=["fast", "exact"])
def integral_mode(request, gh):
    """Both evaluations of the per-hit kernel integral in the column-density trace: "fast"
    (default: hardware sqrt, fp32 lerp) and "exact" (the reference's arithmetic bit for bit)."""
    gh.set_exact_integrals(request.param == "exact")
    yield request.param
    gh.set_exact_integrals(False)


def check_column_densities(got, ref32, ref64, mode, max_term=None):
    """got: traced column densities; ref32 / ref64: the oracle's class-ordered fp32 sum and its
    fp64 sum.  Stated tolerance 1e-5 relative in both modes; "exact" is
    bit-identical to the oracle, "fast" stays within 3e-6 (measured: 2.4e-7 from exact at 10^7
    particles).  max_term: for rays that may consist of a few grazing hits only (short
    segments), the largest single term F(0)/h_min^2 -- the fast evaluation's error is a few
    ulp of the TABLE POSITION, i.e. absolute ~1e-6 of the table's scale per hit, which no
    relative bound on a near-zero sum can express."""
    import numpy as np
    atol = 0.0 if max_term is None else 2e-6 * max_term
    err = np.abs(got - ref64)
    bad = np.nonzero(err > 1e-5 * np.abs(ref64) + atol)[0]
    assert len(bad) == 0, (bad[:5], got[bad[:5]], ref64[bad[:5]])
    if mode == "exact":
        assert np.array_equal(got.view(np.uint32), ref32.view(np.uint32))
    else:
        bad = np.nonzero(err > 3e-6 * np.abs(ref64) + atol)[0]
        assert len(bad) == 0, (bad[:5], got[bad[:5]], ref64[bad[:5]])


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a GPU")
    return torch.device("cuda:0")
