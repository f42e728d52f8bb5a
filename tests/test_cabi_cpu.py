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


def _declared(debug=False):
    """Entry points include/dasv_hip.h declares: the product surface, or (debug=True) the developer surface inside
    `#ifdef DASV_DEBUG` that only libdasv_hip_dbg.so exports."""
    text = open(os.path.join(ROOT, "include", "dasv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    m = re.search(r"#ifdef DASV_DEBUG(.*?)#endif", text, flags=re.S)
    text = m.group(1) if debug else text.replace(m.group(0), "")
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


def test_developer_surface_is_only_in_the_debug_library(lib_path):
    """no debug exports and no environment knobs in the product library.  The tuning knobs and the read-floor probe
    are compiled only under -DDASV_DEBUG into libdasv_hip_dbg.so (same sources); the product library exports none of them and
    does not import getenv."""
    import subprocess
    from doubleattentionspeakerverification_amd import _lib, build
    prod, dbg = ctypes.CDLL(lib_path), ctypes.CDLL(build.debug_lib_path())
    assert _declared(debug=True) == sorted(_lib.DEBUG_SIGNATURES) and _declared(debug=True)
    for name in _declared(debug=True):
        assert not hasattr(prod, name), name
        assert hasattr(dbg, name), name
    for name in _declared():
        assert hasattr(dbg, name), name
    syms = subprocess.run(["nm", "-D", lib_path], capture_output=True, text=True).stdout
    assert "dasv_debug" not in syms and "getenv" not in syms


def test_binding_table_covers_the_header(lib_path):
    from doubleattentionspeakerverification_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.dasv_abi_version() == 3
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


def test_split_k_plan_and_the_packed_chain_gate(lib_path):
    """dasv_conv3x3_workspace_bytes is host arithmetic: zero means one pass fills the chip — the condition under which the
    inference chain keeps its activations packed (a packed output has no split-K form).  Full batches of the default model
    need no workspace in any layer; a single utterance does in the narrow-map layers; the new packed entry points validate
    their arguments without a GPU."""
    from doubleattentionspeakerverification_amd import _lib
    lib = _lib.load()
    layers = [(128, 128, 300, 80), (128, 256, 150, 40), (256, 256, 150, 40), (256, 512, 75, 20), (512, 512, 75, 20),
              (512, 1024, 38, 10), (1024, 1024, 38, 10)]
    for cin, cout, H, W in layers:
        assert lib.dasv_conv3x3_workspace_bytes(64, H, W, cin, cout) == 0
        assert lib.dasv_conv3x3_workspace_bytes(1, 64 * H, W, cin, cout) == 0      # a ragged group of the same footprint
    assert lib.dasv_conv3x3_workspace_bytes(1, 19, 10, 1024, 1024) > 0
    assert lib.dasv_conv3x3_workspace_bytes(1, 19, 10, 1024, 1024) % (19 * 10 * 1024 * 4) == 0   # S partial outputs
    # e4m3 planes | [Cout][4] scale bytes | [Cout] per-channel inverse scales | {L1 bound, max |bias|, pad}
    assert lib.dasv_conv3x3_packed_weight_bytes_x8(1024, 1024) == 1024 * 9 * 1024 * 2 + 1024 * 8 + 16
    assert lib.dasv_conv3x3_bias_relu_ragged_x8_nhwc(None, 1, None, None, None, None, None, 0, None, None, None, 0, 100, 40, 128, 256, 1, None) < 0
    assert lib.dasv_conv3x3_c1_bias_relu_ragged_x8_nhwc(None, None, None, None, None, None, None, 100, 80, 128, 1, None) < 0
    assert lib.dasv_maxpool2x2_ceil_ragged_x8_nhwc(None, None, None, None, None, 100, 50, 80, 128, None) < 0
    # round-3 entry points refuse missing tables / records / index bytes before touching the device
    assert lib.dasv_conv3x3_wgrad_x8_nhwc(16, None, 16, 16, 16, 16, 1 << 30, 2, 8, 8, 64, 64, None) == -1
    assert lib.dasv_conv3x3_wgrad_x8_nhwc(16, 16, 16, None, 16, 16, 1 << 30, 2, 8, 8, 64, 64, None) == -1
    assert lib.dasv_maxpool2x2_ceil_idx_nhwc(16, None, 16, None, None, 2, 8, 8, 64, None) == -1
    assert lib.dasv_maxpool2x2_ceil_bwd_idx_nhwc(None, 16, 16, None, 1, 2, 8, 8, 64, None) == -1
    assert lib.dasv_ragged_tables(None, None, 3, 4, None, None, None) == -1
    import ctypes
    three = (ctypes.c_int * 3)(5, 0, 7)
    assert lib.dasv_ragged_tables(16, ctypes.addressof(three), 3, 4, 16, None, None) == -2      # a zero-length utterance
    assert lib.dasv_ragged_tables(16, ctypes.addressof(three), 5000, 4, 16, None, None) == -2   # more utterances than a group may hold
    # the half-precision ragged chain: tables and records are mandatory
    assert lib.dasv_conv3x3_bias_relu_ragged_f16_nhwc(16, 1, 16, 16, None, 16, 1, 16, None, None, 0, 100, 40, 128, 256, 1, None) == -1
    assert lib.dasv_conv3x3_c1_bias_relu_ragged_f16_nhwc(16, 16, 16, None, 16, 16, None, 100, 80, 128, 1, None) == -1
    assert lib.dasv_maxpool2x2_ceil_ragged_f16_nhwc(16, 16, 16, 1, 16, None, 100, 50, 80, 128, None) == -1
    # the split8 mode scales its operands by range records: without one it refuses (no silent unscaled path), also through the
    # generic entry points, which carry none
    assert lib.dasv_conv3x3_bias_relu_x8_nhwc(16, 0, None, 16, 16, None, 16, 0, None, None, 0, 1, 8, 8, 64, 128, 1, None) == -1
    assert lib.dasv_conv3x3_bias_relu_x8_nhwc(16, 0, 16, 16, 16, None, 16, 1, None, None, 0, 1, 8, 8, 64, 128, 1, None) == -1   # packed y
    assert lib.dasv_conv3x3_bias_relu_nhwc(3, 16, 16, 16, None, 16, 1, 8, 8, 64, 128, 1, None) == -1
    assert lib.dasv_conv3x3_c1_bias_relu_x8_nhwc(16, None, 16, None, 16, None, 1, 8, 8, 64, 1, None) == -1
    assert lib.dasv_maxpool2x2_ceil_x8_nhwc(16, None, 16, None, 1, 8, 8, 64, None) == -1
    assert lib.dasv_range_absmax_f32(None, 16, None, None) == -1
    # the pooling workspace: a fixed-size counter region first, so its layout does not depend on the batch size
    per_utt = 8 * 4 + 8 * (4 + 64) * 4           # head scores + one partial {M, S, 0, 0 | V[dh]} per head at most
    assert lib.dasv_dmha_workspace_bytes(1, 10, 8, 64) - per_utt == lib.dasv_dmha_workspace_bytes(130, 10, 8, 64) - 130 * per_utt == 16384 * 4


def test_kernels_are_built_for_gfx950_only(lib_path):
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_80", b"sm_90"):
        assert other not in blob
