import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build them once, the way
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU; gcc for the oracle).
    On the GPU box the prebuilt in-tree files travel with the snapshot and nothing is compiled."""
    import importlib.util

    pkg = os.path.join(ROOT, "6dof-pose-estimation-and-defect-projection_amd")
    if not os.path.exists(os.path.join(pkg, "libpedp_hip.so")):
        spec = importlib.util.spec_from_file_location("pedp_build", os.path.join(pkg, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(force=False)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libpedp_oracle.so")):
        import pedp_oracle

        pedp_oracle.build()


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
