"""CPU-only: the C-ABI library builds for gfx950, loads, and exports exactly what include/cvcs_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from cvcs_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.lib()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "cvcs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cvcs_[a-zA-Z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    from cvcs_amd import _lib
    names = declared_functions()
    assert names, "no declarations parsed"
    assert sorted(_lib.SIGNATURES) == names
    for n in names:
        assert hasattr(lib, n), f"libcvcs_hip.so does not export {n}"


def test_version_and_error_channel(lib):
    import ctypes as C
    from cvcs_amd import _lib
    assert lib.cvcs_abi_version() == _lib.ABI_VERSION and lib.cvcs_sizeof_conv8_desc() == C.sizeof(_lib.Conv8Desc)
    assert lib.cvcs_conv3x3_fp8(None, None) == -1 and b"null descriptor" in lib.cvcs_last_error()
    assert lib.cvcs_sizeof_conv_desc() == C.sizeof(_lib.ConvDesc) and lib.cvcs_sizeof_wgrad_desc() == C.sizeof(_lib.WgradDesc)
    # argument validation happens on the host, before any HIP call: usable without a GPU
    assert lib.cvcs_conv2d(None, None) == -1
    assert b"null descriptor" in lib.cvcs_last_error()
    assert lib.cvcs_conv_stat_rows(None) == -1
    assert lib.cvcs_bn_bwd_rows(1) == 1
    assert lib.cvcs_ce_workspace_floats(5000) == 2 + 2 * 5


def test_no_oracle_import_in_product():
    """the product package must never route through the oracle (tier rule 3)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cvcs_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dirpath, f)).read(), f
