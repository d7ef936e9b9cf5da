"""ctypes binding of libcugs_hip.so (the C ABI declared in include/cugs_hip.h).

The HIP library is the product.  There is no CPU or PyTorch fallback: if the shared
object is missing or does not export a declared symbol, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUGS_HIP_LIBRARY: full path of another build of the same C ABI (the development build libcugs_hip_dev.so, a
# maintainer's own build).  Still a HIP library or nothing.
LIB_PATH = os.environ.get("CUGS_HIP_LIBRARY") or os.path.join(_HERE, "libcugs_hip.so")

PACKED_STRIDE = 12   # CUGS_PACKED_STRIDE
GRAD_STRIDE = 16     # CUGS_GRAD_STRIDE
TILE = 16            # CUGS_TILE


class Camera(C.Structure):
    """struct cugs_camera."""
    _fields_ = [("view", C.c_float * 16), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("cam_center", C.c_float * 3), ("reserved", C.c_float)]


class AdamGroup(C.Structure):
    """struct cugs_adam_group."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("reserved", C.c_float)]


class AdamFused(C.Structure):
    """struct cugs_adam_fused."""
    _fields_ = [("m", C.c_void_p * 5), ("v", C.c_void_p * 5), ("lr", C.c_float * 5), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("bc1", C.c_float), ("bc2", C.c_float)]


class DensifyArray(C.Structure):
    """struct cugs_densify_array."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_floats", C.c_int32), ("mode", C.c_int32)]


DENSIFY_COPY, DENSIFY_POSITIONS, DENSIFY_SCALES, DENSIFY_STATE = 0, 1, 2, 3

_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float

# name -> (restype, argtypes); must list every function of include/cugs_hip.h
SIGNATURES = {
    "cugs_version": (C.c_char_p, []),
    "cugs_error_string": (C.c_char_p, [_I]),
    "cugs_project_forward": (_I, [_L, _I, _I, _P, _P, _P, _P, _P, C.POINTER(Camera), _F,
                                  _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cugs_project_forward_keyed": (_I, [_L, _I, _I, _P, _P, _P, _P, _P, C.POINTER(Camera), _F,
                                        _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cugs_project_forward_geometry": (_I, [_L, _P, _P, _P, _P, C.POINTER(Camera), _F, _P, _P, _P, _P, _P, _P, _P, _P,
                                           C.c_size_t, _P]),
    "cugs_project_forward_colour": (_I, [_L, _I, _I, _P, _P, C.POINTER(Camera), _P, _P, _P, _P]),
    "cugs_evaluate_sh": (_I, [_I, _L, _I, _P, _P, _P, _P]),
    "cugs_evaluate_sh_backward": (_I, [_I, _L, _I, _P, _P, _P, _P, _P]),
    "cugs_pack_projected": (_I, [_L, _P, _P, _P, _P, _P, _P]),
    "cugs_sort_workspace_bytes": (C.c_size_t, [_L]),
    "cugs_sort_pair_workspace_bytes": (C.c_size_t, [_L]),
    "cugs_sort_count_pairs": (_I, [_L, _P, _P, _P, _P, _I, _I, _P, C.c_size_t, C.POINTER(C.c_int64), _P]),
    "cugs_sort_count_pairs_wide": (_I, [_L, _P, _P, _P, _P, _I, _I, _P, C.c_size_t, C.POINTER(C.c_int64), _P]),
    "cugs_sort_pairs_predicted_wide": (_I, [_L, _L, _P, _P, _P, _P, _I, _I, _P, C.c_size_t, _P, C.c_size_t, _P, _P, _P,
                                            C.POINTER(C.c_int64), _P]),
    "cugs_sort_pairs": (_I, [_L, _L, _P, _P, _P, _P, _I, _I, _P, C.c_size_t, _P, C.c_size_t, _P, _P, _P, _P]),
    "cugs_sort_pairs_predicted": (_I, [_L, _L, _P, _P, _P, _P, _I, _I, _P, C.c_size_t, _P, C.c_size_t, _P, _P, _P,
                                       C.POINTER(C.c_int64), _P]),
    "cugs_sort_pairs_predicted_keyed": (_I, [_L, _L, _P, _P, _P, _P, _I, _I, _P, C.c_size_t, _P, C.c_size_t, _P, _P, _P,
                                             C.POINTER(C.c_int64), _P]),
    "cugs_sort_pairs_predicted_keyed_ordered": (_I, [_L, _L, _P, _P, _P, _P, _I, _I, _P, C.c_size_t, _P, C.c_size_t, _P, _P, _P,
                                                     C.POINTER(C.c_int64), _P, _P]),
    "cugs_tile_order": (_I, [_I, _I, _P, _P, _P]),
    "cugs_rasterize_forward_ordered": (_I, [_I, _I, C.POINTER(C.c_float), _P, _P, _P, _P, _P, _P, _P,
                                            _P, _P, _P, _P, C.c_size_t, _P, _P]),
    "cugs_rasterize_backward_ordered": (_I, [_I, _I, C.POINTER(C.c_float), _P, _P, _P, _P, _P, _P, _P,
                                             _P, _P, _P, _L, _P, _P, _P, _P, _P, _I, _P, _P]),
    "cugs_rasterize_forward": (_I, [_I, _I, C.POINTER(C.c_float), _P, _P, _P, _P, _P, _P, _P,
                                    _P, _P, _P, _P]),
    "cugs_rasterize_forward_zero": (_I, [_I, _I, C.POINTER(C.c_float), _P, _P, _P, _P, _P, _P, _P,
                                         _P, _P, _P, _P, C.c_size_t, _P]),
    "cugs_rasterize_backward": (_I, [_I, _I, C.POINTER(C.c_float), _P, _P, _P, _P, _P, _P, _P,
                                     _P, _P, _P, _L, _P, _P, _P, _P, _P, _P]),
    "cugs_rasterize_backward_prezeroed": (_I, [_I, _I, C.POINTER(C.c_float), _P, _P, _P, _P, _P, _P, _P,
                                               _P, _P, _P, _L, _P, _P, _P, _P, _P, _P]),
    "cugs_project_backward": (_I, [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P, C.POINTER(Camera), _F,
                                   _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cugs_project_backward_adam": (_I, [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P, C.POINTER(Camera), _F, _P,
                                        C.POINTER(AdamFused), _P, _P]),
    "cugs_sh_backward_views": (_I, [_I, _L, _I, _P, _I, _P, C.POINTER(C.c_float), _P, _P]),
    "cugs_gated_colour_grad": (_I, [_L, _P, _P, _P, _P]),
    "cugs_adam_bias_correction": (None, [_F, _F, _I, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "cugs_fused_adam": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P]),
    "cugs_fused_adam_groups": (_I, [C.POINTER(AdamGroup), _I, _F, _F, _F, _F, _F, _P]),
    "cugs_loss_workspace_bytes": (C.c_size_t, [_I, _I]),
    "cugs_combined_loss": (_I, [_I, _I, _P, _P, _F, _I, _P, C.c_size_t, _P, _P, _P, _P]),
    "cugs_densify_accumulate": (_I, [_L, _P, _P, _P, _P, _P, _P]),
    "cugs_densify_classify": (_I, [_L, _P, _P, _P, _P, _P, _F, _F, _F, _I, _F, _F, _P, _P, _P]),
    "cugs_densify_workspace_bytes": (C.c_size_t, [_L]),
    "cugs_densify_plan": (_I, [_L, _P, _P, C.c_size_t, C.POINTER(C.c_int64), _P]),
    "cugs_densify_apply": (_I, [_L, _L, _P, C.c_size_t, _P, _P, C.POINTER(DensifyArray), _I, _P]),
    "cugs_ply_vertex_floats": (_I, [_I, _I]),
    "cugs_ply_pack": (_I, [_L, _I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _P, _P]),
    "cugs_ply_unpack": (_I, [_L, _I, _I, _P, _P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _P]),
    "cugs_image_to_float": (_I, [_I, _I, _P, _I, _I, _P, _P]),
    "cugs_device_count": (_I, [C.POINTER(C.c_int)]),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no fallback path: the HIP library is the implementation.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class CugsError(RuntimeError):
    """Non-zero return from the C ABI (the adapter's std::runtime_error, cuda_utils.cuh:12-20)."""


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib.cugs_error_string(int(code)).decode()
        raise CugsError(f"{what} failed with code {code}: {msg}")


def version() -> str:
    return lib.cugs_version().decode()
