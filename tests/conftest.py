import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Make sure the in-tree native pieces exist (oracle + cross-compiled libhcedge.so) BEFORE the test modules are
    # collected: they import haploconduct_amd, which refuses to load without its library.  `make` makes this a no-op
    # when everything is up to date.
    if not hasattr(config, "workerinput"):  # not in pytest-xdist workers
        import __graft_entry__ as g

        g.build(quiet=True)


@pytest.fixture(scope="session")
def oracle():
    from tests import _oracle

    return _oracle


@pytest.fixture(autouse=True)
def _flush_c_stdio():
    """The library prints the reference's own messages ("incorrect overlap; skipping", src/EdgeCalculator.cpp:600) through C stdio, which —
    stdout being a pipe — buffers them until the process ends, i.e. until pytest's capture is gone: they then land BEHIND the summary line and
    push it out of a log's tail.  Flushed here, inside the test's capture, they stay with their test (and are dropped when it passes)."""
    yield
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
