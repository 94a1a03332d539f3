import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def close_checked(ctx):
    """Closes a context; against a -DUNET_DEBUG build of the library (UNET_OFFLOAD_LIB=.../variants/debug.so) it first
    reads the count of LDS stores that fell outside their wave's area: any is a failure of the tests that ran."""
    bad = ctx.debug_lds_violations()
    ctx.close()
    if bad is not None:
        print(f"\n-DUNET_DEBUG library: {bad} out-of-area LDS stores counted")
        assert bad == 0, f"{bad} LDS stores outside their area (debug build)"


@pytest.fixture(scope="session")
def offload():
    """One offload context on GPU 0 for the whole GPU test session."""
    import unet_amd

    ctx = unet_amd.Offload(0)
    yield ctx
    close_checked(ctx)
