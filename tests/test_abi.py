"""The C-ABI library builds, loads on a GPU-less host and exports every symbol
include/smmregrid_amd.h declares.  No compute calls here."""
import ctypes
import os
import re

import pytest

from smmregrid_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smmregrid_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for name in ("smm_operator_create", "smm_apply", "smm_group_apply", "smm_operator_mask_apply",
                 "smm_operator_export_csr", "smm_last_error"):
        assert name in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_ctypes_table_matches_header():
    table = set(_lib.SIGNATURES) | set(_lib.SPECIAL)
    assert table == set(declared_symbols())


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.smm_abi_version() == 1
    assert isinstance(lib.smm_last_error(), (bytes, type(None)))


def test_no_cpu_fallback_without_device():
    """On a host without a GPU every compute entry fails loudly with NO_DEVICE."""
    if _lib.device_count() > 0:
        pytest.skip("a device is present")
    import numpy as np
    from smmregrid_amd import SparseOperator
    with pytest.raises(_lib.SmmNoDeviceError):
        SparseOperator(4, 4, np.array([1], np.int32), np.array([1], np.int32), np.array([1.0]),
                       device=0)
