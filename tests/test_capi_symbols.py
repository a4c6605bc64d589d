"""The C-ABI library loads on a CPU-only machine and exports every symbol include/bourse_amd.h declares;
without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bourse_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bk_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    import bourse_amd

    names = _declared()
    assert len(names) >= 35
    L = bourse_amd._lib.load()
    for n in names:
        assert hasattr(L, n), f"{n} declared in bourse_amd.h but not exported by libbourse_amd.so"
        assert n in bourse_amd._lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(bourse_amd._lib.SIGNATURES) == set(names)
    raw = ctypes.CDLL(bourse_amd._lib.lib_path())
    assert raw.bk_l2_width(None) == 0  # pure host function, callable without a GPU


def test_struct_layouts_match_header():
    import bourse_amd
    from bourse_amd import _lib

    assert ctypes.sizeof(_lib.Config) == 72
    assert ctypes.sizeof(_lib.RandomAgentsCfg) == 28
    assert ctypes.sizeof(_lib.Stats) == 64
    assert _lib.TRADE_DTYPE.itemsize == 40 and _lib.ORDER_DTYPE.itemsize == 48


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import bourse_amd

    n = ctypes.c_int(-1)
    bourse_amd._lib.load().bk_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(bourse_amd.NoDeviceError):
        bourse_amd.ManyBookEnv(1, 101, 0, 1, 1000)
    with pytest.raises(bourse_amd.NoDeviceError):
        bourse_amd.core.StepEnv(101, 0, 1, 1000)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bourse_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert "pyoracle" not in txt and "bourse_oracle" not in txt, os.path.join(d, f)
