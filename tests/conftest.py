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
    # the product library (hipcc cross-compiles gfx950 without a GPU); `make` is a no-op when fresh
    lib = qpelib.pq.LIB_PATH
    srcs = [p for pat in ("csrc/*", "engine/hip/*.c", "host/*.c") for p in qpelib.PKG.glob(pat)]
    srcs += list((qpelib.ROOT / "include").glob("*.h"))
    if not lib.exists() or lib.stat().st_mtime < max(p.stat().st_mtime for p in srcs):
        qpelib.pq.build_library()
    yield
