import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference build (oracle/_ref). Present in the dev container and, as a prebuilt
    .so, on the GPU box; tests that need it are skipped when it is absent."""
    from oracle import pyoracle
    if not pyoracle.have_ref():
        pytest.skip("oracle/_ref not built")
    return pyoracle.Ref()
