import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_box():
    """a ROCm device node and the built library are present (no HIP call: nothing is initialised at collection time)"""
    return os.path.exists("/dev/kfd") and os.path.exists(os.path.join(ROOT, "mdrp_amd", "libmdrp_hip.so"))


def pytest_collection_modifyitems(config, items):
    """a plain `pytest tests` on the CPU build box skips the GPU parity tests instead of failing them; an explicit
    `-m gpu` selection is never skipped (on a box without a GPU it must fail loudly, not pass vacuously)"""
    if _gpu_box() or "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="needs an MI355X and libmdrp_hip.so (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load
