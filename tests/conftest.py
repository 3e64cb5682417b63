import os
import sys

import pytest

# the library is deaf to its tuning / test knobs (JSDR_TAIL8, JSDR_F32_AS_I16, JSDR_FAST_*_SCALE ...) unless JSDR_KNOBS=1: the tests
# force kernels through them, so the test processes (and the children they start) switch them on -- before the first library call
os.environ.setdefault("JSDR_KNOBS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
