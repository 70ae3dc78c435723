import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
RESOURCES = os.path.join(GOLDEN, "resources")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_builds_are_current():
    """The shared libraries are git-ignored build products: (re)build them in-tree when missing or older than their sources
    (a no-op after `python __graft_entry__.py`; hipcc / gcc exist in the build container and on the GPU box)."""
    import __graft_entry__ as ge
    ge.build_hip()
    ge.build_oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def resources_dir():
    return RESOURCES
