"""ctypes binding of libpdekan.so (the C ABI declared in include/pdekan.h).

The shared object is built in-tree by ``python -m pde_amd.build`` (hipcc,
gfx950).  Loading is lazy; a missing library raises - there is no fallback
implementation of the hot path in this package.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# PDEKAN_LIB: another build of the same library (A/B timing of kernel variants); default: the in-tree build
LIB_PATH = os.environ.get("PDEKAN_LIB") or os.path.join(_HERE, "libpdekan.so")

c_float_p = ctypes.POINTER(ctypes.c_float)


class PdekanLayer(ctypes.Structure):
    _fields_ = [
        ("in_features", ctypes.c_int32),
        ("out_features", ctypes.c_int32),
        ("grid_size", ctypes.c_int32),
        ("spline_order", ctypes.c_int32),
        ("act_kind", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("grid", ctypes.c_void_p),
        ("base_weight", ctypes.c_void_p),
        ("spline_weight", ctypes.c_void_p),
        ("spline_scaler", ctypes.c_void_p),
    ]


class PdekanLayerGrad(ctypes.Structure):
    _fields_ = [
        ("base_weight", ctypes.c_void_p),
        ("spline_weight", ctypes.c_void_p),
        ("spline_scaler", ctypes.c_void_p),
    ]


class PdekanNet(ctypes.Structure):
    _fields_ = [
        ("layers", ctypes.POINTER(PdekanLayer)),
        ("grads", ctypes.POINTER(PdekanLayerGrad)),
        ("n_layers", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/pdekan.h declares
SIGNATURES = {
    "pdekan_version": (ctypes.c_int, []),
    "pdekan_last_error": (ctypes.c_char_p, []),
    "pdekan_seed_jets": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                        ctypes.c_void_p]),
    "pdekan_basis_dense": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                          ctypes.c_void_p]),
    "pdekan_fit_coefficients": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p]),
    "pdekan_layer_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(PdekanLayer), ctypes.c_int64,
                                                       ctypes.c_int32, ctypes.c_int32]),
    "pdekan_layer_forward": (ctypes.c_int, [ctypes.POINTER(PdekanLayer), ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                            ctypes.c_size_t, ctypes.c_void_p]),
    "pdekan_layer_forward_noalloc": (ctypes.c_int, [ctypes.POINTER(PdekanLayer), ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "pdekan_layer_backward": (ctypes.c_int, [ctypes.POINTER(PdekanLayer), ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                             ctypes.POINTER(PdekanLayerGrad), ctypes.c_int32, ctypes.c_void_p,
                                             ctypes.c_size_t, ctypes.c_void_p]),
    "pdekan_model_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(PdekanLayer), ctypes.c_int32,
                                                       ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                       ctypes.c_int64]),
    "pdekan_model_forward_jet": (ctypes.c_int, [ctypes.POINTER(PdekanLayer), ctypes.c_int32, ctypes.c_void_p,
                                                ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                ctypes.c_size_t, ctypes.c_void_p]),
    "pdekan_bs_step": (ctypes.c_int, [ctypes.POINTER(PdekanLayer), ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.POINTER(PdekanLayerGrad), ctypes.c_int32,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "pdekan_deeponet_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(PdekanNet), ctypes.c_int64, ctypes.c_int64,
                                                          ctypes.c_int64]),
    "pdekan_deeponet_step": (ctypes.c_int, [ctypes.POINTER(PdekanNet), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_void_p]),
    "pdekan_profile_begin": (ctypes.c_int, []),
    "pdekan_profile_end": (ctypes.c_int, []),
    "pdekan_profile_count": (ctypes.c_int, []),
    "pdekan_profile_entry": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
}

_lib: Optional[ctypes.CDLL] = None


class PdekanError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libpdekan.so once and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PdekanError(
            f"{LIB_PATH} is missing: build it with `python -m pde_amd.build` (needs hipcc). "
            "pde_amd has no CPU or PyTorch fallback for the KAN hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def profile_results():
    """Per-kernel totals gathered between pdekan_profile_begin/end:
    ``{name: (total_ms, launches)}``."""
    lib = load()
    out = {}
    for i in range(lib.pdekan_profile_count()):
        name = ctypes.create_string_buffer(64)
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        check(lib.pdekan_profile_entry(i, name, 64, ctypes.byref(ms), ctypes.byref(cnt)))
        out[name.value.decode()] = (ms.value, cnt.value)
    return out


def check(rc: int) -> None:
    if rc != 0:
        msg = load().pdekan_last_error().decode("utf-8", "replace")
        raise PdekanError(f"libpdekan error {rc}: {msg}")
