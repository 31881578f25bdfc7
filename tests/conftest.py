import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build every native piece once (HIP library cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import oracle_py as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hip(built):
    """HipRenderer on device 0; GPU tests only."""
    from cpuraytracer_amd import HipRenderer
    r = HipRenderer(0)
    yield r
    r.close()
