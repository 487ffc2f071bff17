"""pytest configuration: registers the `gpu` marker and builds the CPU checker once."""
import pathlib
import sys

import pytest

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    import qpelib

    so = qpelib.ORACLE_DIR / "libqpe_oracle.so"
    src = qpelib.ORACLE_DIR / "qpe_oracle.c"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        qpelib.build_oracle()
    yield
