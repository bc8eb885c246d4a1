import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def engine():
    import pailliercryptolib_amd as pa
    from pailliercryptolib_amd import build
    build.build_pgpu()          # no-ops when the in-tree libraries are up to date (a fresh checkout has none)
    build.build_ipcl()
    pa.initialize()
    yield pa
    pa.terminate()
