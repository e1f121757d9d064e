import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GGL_TEST_OPTIONS="name=value,...": ctx options every engine of the session starts with (gglasso_amd._lib.OPTIONS) -- for
    # running the whole suite with a feature switched off when a failure has to be pinned on it
    # GGL_DEBUG_POISON=1: every ctx of the session starts from 0xFF-filled buffers instead of zeros (ggl_debug_poison)
    # (GGL_DEBUG_POISON=127 / 71: 0x7F / 0x47 bytes -- FINITE garbage, 1.4e306 / 1.5e35, which a max reduction keeps where it
    # drops a NaN)
    if os.environ.get("GGL_DEBUG_POISON", "") not in ("", "0"):
        from gglasso_amd import _lib
        _lib.load().ggl_debug_poison(int(os.environ["GGL_DEBUG_POISON"]))
    extra = os.environ.get("GGL_TEST_OPTIONS", "")
    if extra:
        from gglasso_amd import solver
        for item in extra.split(","):
            name, value = item.split("=")
            solver.ENGINE_OPTIONS[name.strip()] = float(value)


@pytest.fixture()
def dev_library(monkeypatch):
    """Tests of the measured-and-rejected alternatives (GGL_OPT_CHAIN, GGL_OPT_BOUND_SIDE, ...: options of the development
    library only since round 6): every engine of the test is created in libggl_hip_dev.so (python -m gglasso_amd.build --dev),
    which carries everything the product library does.  Skipped when that library has not been built."""
    from gglasso_amd import _lib
    if not os.path.exists(_lib.DEV_LIB_PATH):
        pytest.skip("libggl_hip_dev.so is not built (python -m gglasso_amd.build --dev)")
    dev = _lib.load_dev()
    monkeypatch.setattr(_lib, "_lib", dev)
    return dev


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden
