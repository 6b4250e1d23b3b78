import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        from libdogleg_amd import capi
        return capi.lib().dlg_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail loudly (not skip) when the HIP library or device is missing."""
    from libdogleg_amd import capi
    L = capi.lib()
    assert L.dlg_device_count() > 0, "no HIP device visible: -m gpu tests need an MI355X"
    return L
