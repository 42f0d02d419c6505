import importlib
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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session", autouse=True)
def _library_built():
    """The tests bind libnunet.so through ctypes; a fresh checkout has only sources (the .so is git-ignored), so it is
    compiled here when absent (hipcc cross-compiles gfx950 without a GPU; __graft_entry__.build() does the same)."""
    L = importlib.import_module("pytorch_nested-unet_amd._lib")
    if not os.path.exists(L.LIB_PATH):
        L.build()


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("pytorch_nested-unet_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("pytorch_nested-unet_amd.synth")


GOLDEN_CASES = {
    # name: (N, H, W, cin, ncls, ds, train, fresh_bn)   -- mirrors tests/golden/make_golden.py
    "a_n2_32x32_k1": (2, 32, 32, 3, 1, False, True, True),
    "b_n2_32x32_k1_ds": (2, 32, 32, 3, 1, True, True, True),
    "c_n2_32x32_k4": (2, 32, 32, 3, 4, False, True, True),
    "d_n2_16x16_k1": (2, 16, 16, 3, 1, False, True, True),
    "e_n2_32x32_k1_eval": (2, 32, 32, 3, 1, False, False, False),
    "f_n3_48x32_k1_ds": (3, 48, 32, 3, 1, True, True, False),
    "g_n1_64x64_k2_c1": (1, 64, 64, 1, 2, False, True, True),
}
