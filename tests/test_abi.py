"""The C-ABI library loads and exports every symbol include/gzpx.h declares (no compute calls:
there is no GPU here) and refuses to work without a device.  No GPU."""
import ctypes
import os
import re

import pytest

from gzp_amd import _native

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def built_lib():
    from gzp_amd import build
    return build.build()


def test_header_symbols_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "gzpx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # prose in comments is not a declaration
    declared = set(re.findall(r"\b(gzpx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = ctypes.CDLL(built_lib)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(_native.EXPORTS) <= declared


def test_release_build_has_no_experiment_hooks(built_lib):
    """The measurement-only switches (-DGZPX_EXPERIMENT: parts of kernels skipped, cycle counters)
    live in a second library that tools/exp_*.py build; the product library must not carry them."""
    L = ctypes.CDLL(built_lib)
    assert not hasattr(L, "gzpx_exp_cycles")
    blob = open(built_lib, "rb").read()
    assert b"g_exp_cycles" not in blob


def test_no_cpu_fallback_without_device(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _native.GzpxLib(built_lib)
    with pytest.raises(_native.GzpxError) as ei:
        _native.Context(level=1, lib=lib)
    assert ei.value.code in (_native.ERR_NO_DEVICE, _native.ERR_DEVICE)
    assert lib.L.gzpx_alloc_compressor(99) is None


def test_product_package_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gzp_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("binary oracle", ""), os.path.join(dirpath, f)
