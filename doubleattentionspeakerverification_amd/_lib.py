"""ctypes binding of libdasv_hip.so (include/dasv_hip.h).

There is no fallback: if the library is missing or a symbol cannot be bound,
importing the compute path raises.  Build it with
``python -m doubleattentionspeakerverification_amd.build`` (or
``__graft_entry__.build()``).
"""
import ctypes
import os
import threading

from . import build as _build

_lock = threading.Lock()
_lib = None

c_void_p, c_int, c_int64, c_size_t, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float

_FWD_ARGS = [c_void_p] * 10 + [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p]

# name -> (restype, argtypes); must list every function include/dasv_hip.h declares
SIGNATURES = {
    "dasv_abi_version": (c_int, []),
    "dasv_error_string": (ctypes.c_char_p, [c_int]),
    "dasv_dmha_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dasv_dmha_workspace_init": (c_int, [c_void_p, c_size_t, c_void_p]),
    "dasv_dmha_fwd": (c_int, [c_int] + _FWD_ARGS),
    "dasv_dmha_fwd_f32": (c_int, _FWD_ARGS),
    "dasv_dmha_fwd_bf16": (c_int, _FWD_ARGS),
    "dasv_dmha_fwd_f16": (c_int, _FWD_ARGS),
    "dasv_dmha_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dasv_dmha_bwd": (c_int, [c_int] + [c_void_p] * 13 + [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "dasv_conv3x3_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "dasv_conv3x3_packed_weight_bytes_x8": (c_size_t, [c_int, c_int]),
    "dasv_conv3x3_pack_weights_x8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dasv_conv3x3_pack_weights_x8_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dasv_conv3x3_pack_weights_x8_dgrad_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dasv_range_absmax_f32": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "dasv_conv3x3_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dasv_conv3x3_packed_weight_bytes_f16": (c_size_t, [c_int, c_int]),
    "dasv_conv3x3_pack_weights_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dasv_conv3x3_bias_relu_f16_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                                c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bias_relu_f16_nhwc": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_f16_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_bias_relu_ragged_f16_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                       c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bias_relu_ragged_f16_nhwc": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_ragged_f16_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_bias_relu_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dasv_conv3x3_bias_relu_nhwc_ws": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_bias_relu_ragged_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bias_relu_ragged_nhwc": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_ragged_nhwc": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_bias_relu_ragged_x8_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                                      c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bias_relu_ragged_x8_nhwc": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_ragged_x8_nhwc": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_masked_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_colsum_rows": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "dasv_conv3x3_masked_x8_nhwc": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_conv3x3_bias_relu_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dasv_conv3x3_wgrad_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_wgrad_x8_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bias_relu_nhwc": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_nhwc": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_bwd_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_relu_bwd_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_bwd_colsum_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "dasv_ragged_tables": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dasv_maxpool2x2_ceil_idx_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_bwd_idx_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_idx_ragged_nhwc": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_bwd_idx_ragged_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_bwd_colsum_rows_ragged": (c_int, [c_int, c_int, c_int]),
    "dasv_conv3x3_masked_ragged_x8_nhwc": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bwd_ragged_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dasv_conv3x3_c1_bwd_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_bias_relu_x8_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                               c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bias_relu_x8_nhwc": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_pool_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dasv_conv3x3_bias_relu_pool_x8_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                                    c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_bias_relu_pool_idx_x8_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                        c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_masked_x8p_nhwc": (c_int, [c_void_p, c_int] + [c_void_p] * 8 + [c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_bwd_idx_x8_nhwc": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_masked_c1_x8p_nhwc": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_adam_step_f32": (c_int, [c_void_p] * 5 + [ctypes.c_longlong, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                   ctypes.c_float, ctypes.c_longlong, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_idx_ragged_x8_nhwc": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_bwd_idx_ragged_x8_nhwc": (c_int, [c_void_p] * 6 + [c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_masked_ragged_x8p_nhwc": (c_int, [c_void_p, c_int] + [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_bias_relu_bf16_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int, c_int, c_int, c_int,
                                                 c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bias_relu_bf16_nhwc": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_bf16_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_colsum_reduce": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dasv_conv3x3_masked_c1_ragged_x8p_nhwc": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_conv3x3_c1_bwd_reduce_parts": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "dasv_conv3x3_wgrad_x8p_nhwc": (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 3 + [c_size_t] + [c_int] * 5 + [c_void_p]),
    "dasv_conv3x3_bias_relu_pool_f16_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                                     c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_maxpool2x2_ceil_x8_nhwc": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_dmha_nhwc_supported": (c_int, [c_int, c_int, c_int]),
    "dasv_dmha_fwd_nhwc_f32": (c_int, [c_void_p] * 10 + [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_dmha_bwd_nhwc_f32": (c_int, [c_void_p] * 12 + [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dasv_dmha_last_variant": (ctypes.c_char_p, []),
    "dasv_amsoftmax_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dasv_amsoftmax_fwd": (c_int, [c_void_p] * 10 + [c_size_t, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    "dasv_amsoftmax_bwd": (c_int, [c_void_p] * 13 + [c_size_t, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p]),
}

# the developer surface: exported by libdasv_hip_dbg.so only (include/dasv_hip.h, #ifdef DASV_DEBUG)
DEBUG_SIGNATURES = {
    "dasv_debug_set_tuning": (None, [c_int, c_int, c_int, c_int, c_int]),
    "dasv_debug_read_floor": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "dasv_debug_conv_patch_tiles": (None, [c_int]),
}

DTYPE_CODES = {"float32": 0, "bfloat16": 1, "float16": 2}


class DasvError(RuntimeError):
    code = 0   # the C ABI's return value (DASV_E_* < 0, hipError_t > 0)


def lib_path():
    return _build.lib_path()


def _bind(path, signatures):
    if not os.path.exists(path):
        raise DasvError(
            "HIP library %s is not built; run `python -m doubleattentionspeakerverification_amd.build`. "
            "There is no CPU fallback for this path." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    if lib.dasv_abi_version() != 3:
        raise DasvError("%s ABI version %d, expected 3" % (path, lib.dasv_abi_version()))
    return lib


def load():
    """Load (once) and return the bound library.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            _lib = _bind(lib_path(), SIGNATURES)
    return _lib


_probe = None


def debug_probe():
    """libdasv_hip_dbg.so bound for its developer entry points only (bench.py's read-floor probe, kernel sweeps through the C ABI):
    loaded BESIDE the product library, which keeps serving every module call of the process."""
    global _probe
    with _lock:
        if _probe is None:
            _probe = _bind(_build.debug_lib_path(), {**SIGNATURES, **DEBUG_SIGNATURES})
    return _probe


def use_debug_library():
    """Developer runs only (kernel sweeps, the read-floor probe of bench.py, the tests that force a kernel variant): make THIS
    process launch through libdasv_hip_dbg.so — the same sources built with -DDASV_DEBUG — which also exports the developer
    surface (dasv_debug_*).  Call before anything else has loaded the product library."""
    global _lib
    with _lock:
        if _lib is not None and not hasattr(_lib, "_dasv_debug"):
            raise DasvError("the product library is already loaded in this process")
        if _lib is None:
            lib = _bind(_build.debug_lib_path(), {**SIGNATURES, **DEBUG_SIGNATURES})
            lib._dasv_debug = True
            _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        msg = load().dasv_error_string(code)
        err = DasvError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))
        err.code = code
        raise err
