import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are deselected by the -m expression the driver passes; nothing to do here.
    pass


@pytest.fixture(scope="session")
def built():
    """Everything compiled: product library (hipcc), oracle restatement (gcc), synthetic generator (g++)."""
    import fcntl
    import __graft_entry__ as g
    # (pytest-xdist: every worker process has a session of its own - one of them builds, the others wait and then find everything up to date)
    with open(os.path.join(ROOT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            g.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return True
