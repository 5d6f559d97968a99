import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pedp_oracle

    pedp_oracle.build()
    return pedp_oracle


@pytest.fixture(scope="session")
def ctx():
    """One library context for the whole GPU session (fails loudly if the HIP library or
    the GPU is missing: there is no fallback to test)."""
    from pedp_hip import _lib

    c = _lib.Context(0)
    yield c
    c.close()
