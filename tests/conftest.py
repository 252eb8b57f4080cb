import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One device context per test session.  Fails (does not skip) when the HIP library or the
    device is missing: -m gpu tests must never pass on a fallback."""
    from polr_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
