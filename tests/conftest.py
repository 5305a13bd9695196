import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The parity tests compare per-pixel step counters with the oracle's: every step is marched unless a test asks otherwise
# (sdfr_set_step_shortcuts; tests/test_gpu_shortcuts.py covers the default of the library, shortcuts on).
os.environ.setdefault("SDFR_STEP_SHORTCUTS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); built on demand with gcc."""
    from oracle import pyoracle

    pyoracle.build(census=False, ref=os.path.isdir("/root/reference/Engine"))
    return pyoracle
