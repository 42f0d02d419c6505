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


class _GuardBands:
    """Guard bands around every device buffer a test hands to the library: while active, the torch
    factory functions the op tests allocate with (zeros / empty / full / ones / randn and their *_like forms, for a
    CUDA device) return views into a larger byte buffer whose first and last BAND bytes carry a canary pattern; the
    trailing band starts at the first byte after the tensor. check() verifies every band after the test's kernels ran:
    a kernel that writes one element past (or before) a caller-provided buffer fails the test instead of corrupting a
    neighbour - or faulting only once an allocation happens to end a mapping."""
    BAND = 4096
    PATTERN = 0xA5

    def __init__(self, torch):
        self.torch = torch
        self.bufs = []
        self.saved = {}

    def _wrap(self, name):
        torch, orig = self.torch, getattr(self.torch, name)

        def factory(*a, **kw):
            t = orig(*a, **kw)
            if not t.is_cuda or t.numel() == 0 or not t.is_contiguous():
                return t
            nb = t.numel() * t.element_size()
            raw = self.saved["empty"](2 * self.BAND + (nb + 255) // 256 * 256, dtype=torch.uint8, device=t.device)
            raw.fill_(self.PATTERN)
            v = raw[self.BAND:self.BAND + nb].view(t.dtype).view(t.shape)
            v.copy_(t)
            self.bufs.append((raw, nb, name, tuple(t.shape)))
            return v
        return factory

    NAMES = ("zeros", "empty", "full", "ones", "randn", "zeros_like", "empty_like", "full_like", "ones_like")

    def __enter__(self):
        for n in self.NAMES:
            self.saved[n] = getattr(self.torch, n)
        for n in self.NAMES:
            setattr(self.torch, n, self._wrap(n))
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(self.torch, n, f)

    def check(self):
        self.torch.cuda.synchronize()
        bad = []
        for raw, nb, name, shape in self.bufs:
            head = raw[:self.BAND]
            tail = raw[self.BAND + nb:]
            if not bool((head == self.PATTERN).all()):
                bad.append("write BEFORE a torch.%s%s buffer" % (name, shape))
            if not bool((tail == self.PATTERN).all()):
                first = int((tail != self.PATTERN).nonzero()[0])
                bad.append("write %d bytes PAST a torch.%s%s buffer of %d bytes" % (first, name, shape, nb))
        self.bufs = []
        return bad


@pytest.fixture
def guard_bands():
    """Used (autouse) by the per-kernel GPU tests: see _GuardBands."""
    import torch
    with _GuardBands(torch) as g:
        yield g
        bad = g.check()
    assert not bad, "guard band violated: " + "; ".join(bad)
