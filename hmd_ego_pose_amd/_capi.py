"""ctypes binding of ``libhep.so`` (the C ABI declared in include/hep.h).

PyTorch is only plumbing here: device memory (``tensor.data_ptr()``) and the current
HIP stream.  ``import torch`` happens before the library is loaded so that libhep binds
to the HIP runtime torch already brought into the process (one runtime, shared streams).
There is no fallback: a missing library raises ``HepError`` on first use.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint, c_void_p

import torch  # noqa: F401  (must precede loading libhep.so, see module docstring)

# HEP_LIB selects another build of the same C-ABI (the opt-in fp8 build libhep_fp8.so; the profiling tools set LIB_PATH themselves)
LIB_PATH = os.environ.get("HEP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhep.so")
HEP_F32, HEP_BF16, HEP_FP8 = 0, 1, 2
FLAG_KEEP_INTERMEDIATES, FLAG_NO_GRAPH = 1, 2
OUT_K = (4, 1, 3, 3, 63)

# every symbol include/hep.h declares: (restype, argtypes)
_P = c_void_p
_FP = c_void_p          # float* passed as raw addresses
SYMBOLS = {
    "hep_abi_version": (c_int, []),
    "hep_build_info": (c_char_p, []),
    "hep_last_error": (c_char_p, []),
    "hep_device_count": (c_int, []),
    "hep_create": (c_int, [c_char_p, c_int, c_int, c_int, c_int, c_int, c_uint, POINTER(_P)]),
    "hep_create_from_memory": (c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_uint, POINTER(_P)]),
    "hep_destroy": (None, [_P]),
    "hep_num_anchors": (c_int, [_P]),
    "hep_num_classes": (c_int, [_P]),
    "hep_output_shape": (c_int, [_P, c_int, c_int, POINTER(c_int64), POINTER(c_int)]),
    "hep_output_device": (c_int, [_P, c_int, POINTER(_FP)]),
    "hep_run": (c_int, [_P, _FP, c_int, POINTER(_FP), _FP, _FP, _FP, _FP, _FP]),
    "hep_run_device": (c_int, [_P, _FP, POINTER(c_int64), c_int, POINTER(_FP), POINTER(_FP), c_void_p]),
    "hep_anchors": (c_int, [c_int, _FP, _FP]),
    "hep_decode": (c_int, [_P, _FP, _FP, _FP, c_int, _FP, _FP]),
    "hep_decode_device": (c_int, [_P, _FP, _FP, _FP, c_int, _FP, _FP, c_void_p]),
    "hep_preprocess_u8_device": (c_int, [_P, c_void_p, c_int, c_int, c_int, _FP, c_void_p]),
    "hep_preprocess_i420_device": (c_int, [_P, c_void_p, c_int, c_int, c_int, c_int, c_int, _FP, c_void_p]),
    "hep_set_class_specific_filter": (c_int, [_P, c_int]),
    "hep_filter": (c_int, [_P, _FP, _FP, _FP, _FP, _FP, c_int, c_float, c_float, c_int] + [_FP] * 8),
    "hep_filter_device": (c_int, [_P, _FP, _FP, _FP, _FP, _FP, c_int, c_float, c_float, c_int] + [_FP] * 8 + [c_void_p]),
    "hep_pose_errors": (c_int, [c_int, _FP, c_int, _FP, _FP, _FP, _FP, c_int, c_int, _FP, _FP]),
    "hep_pose_errors_device": (c_int, [_FP, c_int, _FP, _FP, _FP, _FP, c_int, c_int, _FP, _FP, c_void_p]),
    "hep_anchor_targets_device": (c_int, [_FP, c_int, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_double, c_double, _FP, _FP, _FP, _FP, c_void_p]),
    "hep_losses_device": (c_int, [_FP] * 9 + [c_int] * 7 + [_FP, _FP, c_void_p]),
    "hep_debug_tensor_count": (c_int, [_P]),
    "hep_debug_tensor_info": (c_int, [_P, c_int, POINTER(c_char_p), POINTER(c_int64)]),
    "hep_debug_tensor": (c_int, [_P, c_char_p, c_int, _FP, c_size_t]),
    "hep_kernel_count": (c_int, [_P, c_int]),
    "hep_kernel_info": (c_int, [_P, c_int, c_int, POINTER(c_char_p), POINTER(c_double), POINTER(c_double)]),
    "hep_kernel_symbol": (c_int, [_P, c_int, POINTER(c_char_p)]),
    "hep_fp8_scale": (c_int, [_P, c_int, POINTER(c_float)]),
    "hep_calibrate_fp8": (c_int, [_P, _FP, c_int]),
    "hep_profile": (c_int, [_P, c_int, c_int, POINTER(c_float), _FP]),
    "hep_profile_concurrent": (c_int, [_P, c_int, c_int, c_int, _FP]),
}


class HepError(RuntimeError):
    pass


class HepUnsupported(HepError):
    """HEP_ERR_UNSUPPORTED: a phi / size / dtype this build of libhep.so does not cover (e.g. HEP_FP8 without ``make FP8=1``)."""


_lib = None


def lib() -> ctypes.CDLL:
    """Load libhep.so (once).  Fails loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HepError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C hmd_ego_pose_amd/csrc`). There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)        # AttributeError here = ABI drift between hep.h and the .so
            fn.restype, fn.argtypes = res, args
        if l.hep_abi_version() != 1:
            raise HepError("libhep.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise (HepUnsupported if rc == -4 else HepError)(f"libhep error {rc}: {lib().hep_last_error().decode(errors='replace')}")
    return rc


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    arr = (_FP * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t)
    return arr
