import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def orc():
    import orclib
    return orclib.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own sources compiled by oracle/build_ref.sh (oracle/_ref)."""
    import orclib
    if not orclib.Reference.available():
        pytest.skip("oracle/_ref/libmsdr_ref.so not built (needs /root/reference: `make -C oracle ref`)")
    return orclib.Reference()


@pytest.fixture(scope="session")
def golden():
    import goldenlib
    return goldenlib.load()
