import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def restated():
    import oracle

    return oracle.restated()


@pytest.fixture(scope="session")
def reference():
    import oracle

    if not oracle.have_reference():
        pytest.skip("oracle/_ref/libgsplat_ref.so not built (needs /root/reference)")
    return oracle.reference()
