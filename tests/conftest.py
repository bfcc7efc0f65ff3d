import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The tests drive the library's test hooks (gk_ctx_set_option, the loopback transport): they load the TEST build — the product
# library's own objects plus csrc/gk_testhooks.o — unless the caller chose a build (a variant under genome_amd/variants/).
# smoke() and bench.py load the product library.
os.environ.setdefault("GK_LIB_PATH", os.path.join(ROOT, "genome_amd", "libgenome_amd_test.so"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def has_gpu() -> bool:
    """True when a HIP device is usable (no torch involved: asks the C-ABI library)."""
    try:
        from genome_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return has_gpu()
