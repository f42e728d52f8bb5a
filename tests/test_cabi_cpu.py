"""CPU-only checks of the C-ABI shared library: it builds (hipcc cross-compiles gfx950 without a GPU),
loads, and exports exactly the entry points include/dasv_hip.h declares.  No compute launches here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from doubleattentionspeakerverification_amd import build
    return build.build(verbose=False)


def _declared():
    text = open(os.path.join(ROOT, "include", "dasv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dasv_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("dasv_dmha_fwd", "dasv_dmha_fwd_f32", "dasv_dmha_fwd_bf16", "dasv_dmha_fwd_f16", "dasv_dmha_bwd",
                 "dasv_dmha_workspace_bytes", "dasv_dmha_workspace_init", "dasv_dmha_bwd_workspace_bytes"):
        assert must in names


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(lib, name), "libdasv_hip.so does not export %s" % name


def test_binding_table_covers_the_header(lib_path):
    from doubleattentionspeakerverification_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.dasv_abi_version() == 1
    assert lib.dasv_error_string(-3) == b"workspace missing or too small"
    assert lib.dasv_dmha_workspace_bytes(64, 300, 8, 64) >= 64 * 4
    assert lib.dasv_dmha_bwd_workspace_bytes(64, 300, 8, 64) >= 64 * 512 * 4
    assert lib.dasv_dmha_workspace_bytes(0, 1, 1, 1) == 0


def test_argument_validation_returns_codes_without_touching_the_gpu(lib_path):
    from doubleattentionspeakerverification_amd import _lib
    lib = _lib.load()
    nul = None
    # x NULL -> DASV_E_NULL ; bad dtype -> DASV_E_DTYPE ; bad shape -> DASV_E_SHAPE
    assert lib.dasv_dmha_fwd(0, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, 0, 1, 1, 1, 4, 4, 4, nul) == -1
    assert lib.dasv_dmha_fwd(7, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, 0, 1, 1, 1, 4, 4, 4, nul) == -5
    assert lib.dasv_dmha_fwd(0, 16, 16, nul, nul, nul, nul, nul, nul, 16, nul, nul, 0, 0, 1, 1, 4, 4, 4, nul) == -2
    assert lib.dasv_dmha_fwd(0, 16, 16, nul, nul, nul, nul, nul, nul, 16, nul, nul, 0, 1, 1, 1, 4, 4, 2, nul) == -4
    assert lib.dasv_dmha_bwd(0, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, 0, 1, 1, 1, 4, 4, 4, nul) == -1


def test_kernels_are_built_for_gfx950_only(lib_path):
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_80", b"sm_90"):
        assert other not in blob
