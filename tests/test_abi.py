"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports
every symbol include/nunet.h declares; host-only entry points behave."""
import ctypes as C
import os
import re

import pytest

import nunet_amd
from nunet_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        L.build()
    return L.lib()


def _declared():
    """every function include/nunet.h (the product boundary) and include/nunet_diag.h (test / diagnostic hooks) declare"""
    out = set()
    for h in ("nunet.h", "nunet_diag.h"):
        hdr = open(os.path.join(ROOT, "include", h)).read()
        out |= set(re.findall(r"\b(nunet_[a-z0-9_]+)\s*\(", hdr))
    return out - {"nunet_plan"}


def test_header_symbols_all_exported_and_bound(lib):
    declared = _declared()
    assert declared, "no declarations parsed"
    raw = C.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "libnunet.so does not export %s" % name
        assert name in L._SIG, "%s is not bound in _lib.py" % name
    assert set(L._SIG) == declared
    assert lib.nunet_version() >= 100


def test_product_library_carries_no_diagnostic_build(lib):
    """The conv kernel source has two compile-time diagnostic modes (-DNUNET_KSTAMP: in-kernel phase stamps,
    -DNUNET_ABLATE=bits: parts of the kernel compiled out; tools/kstamp_build.sh, tools/ablate_build.sh write their
    libraries under tools/_diag/). The product library must be built with neither: it exports nothing but what
    include/nunet.h declares (the stamp build adds nunet_kstamp_set), and the Makefile passes no such define."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (nunet_[a-z0-9_]+)", out))
    declared = _declared()
    prod = open(os.path.join(ROOT, "include", "nunet.h")).read()
    for hook in ("nunet_debug_spin", "nunet_plan_stamps_read", "nunet_plan_set_lanes"):
        assert hook not in prod, "%s is a diagnostic hook: it belongs in include/nunet_diag.h" % hook
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    mk = open(os.path.join(os.path.dirname(L.LIB_PATH), "csrc", "Makefile")).read()
    assert "NUNET_KSTAMP" not in mk and "NUNET_ABLATE" not in mk
    assert os.environ.get("NUNET_LIB_PATH") is None or "_diag" not in os.environ["NUNET_LIB_PATH"]


def test_plan_layout_matches_reference_state_dict(lib):
    """Host-only: the plan's flat parameter layout equals the reference parameters() order/size
    (9,163,329 params w/o DS, 9,163,428 with)."""
    for ds, ncls, expect in ((0, 1, 9163329), (1, 1, 9163428), (0, 4, 9163428)):
        cfg = L.PlanCfg(2, 32, 32, 3, ncls, ds, L.F32, 0)
        p = lib.nunet_plan_create(C.byref(cfg))
        assert p
        assert lib.nunet_plan_param_count(p) == expect
        assert lib.nunet_plan_bn_layers(p) == 30
        assert lib.nunet_plan_bnbuf_count(p) == 2 * 2 * (32 * 5 + 64 * 4 + 128 * 3 + 256 * 2 + 512)
        assert lib.nunet_plan_num_heads(p) == (4 if ds else 1)
        assert lib.nunet_plan_arena_bytes(p) > 0
        m = nunet_amd.archs.NestedUNet(ncls, 3, bool(ds))
        assert sum(q.numel() for q in m.parameters()) == expect
        lib.nunet_plan_destroy(p)
    cfg = L.PlanCfg(2, 32, 32, 3, 1, 0, L.BF16, 1)
    p = lib.nunet_plan_create(C.byref(cfg))
    assert lib.nunet_plan_param_count(p) == sum(q.numel() for q in nunet_amd.archs.UNet(1).parameters())
    lib.nunet_plan_destroy(p)


def test_plan_rejects_bad_shapes(lib):
    for bad in (L.PlanCfg(2, 40, 32, 3, 1, 0, L.F32, 0), L.PlanCfg(0, 32, 32, 3, 1, 0, L.F32, 0),
                L.PlanCfg(2, 32, 32, 3, 9, 0, L.F32, 0), L.PlanCfg(2, 32, 32, 3, 1, 0, 7, 0)):
        assert not lib.nunet_plan_create(C.byref(bad))
        assert lib.nunet_last_error()


def test_module_surface_matches_reference():
    m = nunet_amd.archs.NestedUNet(1, 3, True)
    keys = list(m.state_dict().keys())
    spec = [k for k, _, _ in nunet_amd.synth.state_dict_spec(1, 3, True)]
    assert keys == spec and len(keys) == 218
    assert nunet_amd.archs.__all__ == ['UNet', 'NestedUNet']
    assert nunet_amd.losses.__all__ == ['BCEDiceLoss', 'LovaszHingeLoss']      # reference losses.py:100
    assert nunet_amd.utils.count_params(m) == 9163428
    with pytest.raises(L.NunetError):
        import torch
        m(torch.zeros(1, 3, 32, 32))     # CPU tensors: no silent fallback
