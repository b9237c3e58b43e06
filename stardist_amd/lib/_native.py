"""ctypes binding of libstardist_hip.so (C ABI declared in include/stardist_hip.h).

This is the only place the shared library is loaded.  There is NO CPU fallback: if the
library is missing or no HIP device is visible, every compute entry point raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "csrc", "libstardist_hip.so")

_lib = None
# statistics of the most recent native NMS call (see include/stardist_hip.h): {'nms2d': int64[16], 'nms3d': int64[16]}
last_stats = {}

_c_f32p = ctypes.POINTER(ctypes.c_float)
_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_c_u16p = ctypes.POINTER(ctypes.c_uint16)
_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float

# name -> (restype, argtypes); must list every symbol include/stardist_hip.h declares
SIGNATURES = {
    "sd_last_error": (ctypes.c_char_p, []),
    "sd_version": (_i, []),
    "sd_device_count": (_i, []),
    "sd_release_workspace": (_i, []),
    "sd_set_option": (_i, [ctypes.c_char_p, _i]),
    "sd_get_option": (_i, [ctypes.c_char_p]),
    "sd_nms2d_host": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "sd_nms2d_device": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "sd_nms2d_old_host": (_i, [_vp, _i, _i, _vp, _i, _i, _f, _i, _i, _i, _i, _vp]),
    "sd_nms2d_old_device": (_i, [_vp, _i, _i, _vp, _i, _i, _f, _i, _i, _i, _i, _vp, _vp]),
    "sd_clip_pairs_device": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sd_area_bounds_pairs_device": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sd_prepare_polys_device": (_i, [_vp, _vp, _i, _i, _vp, ctypes.c_int64, _vp]),
    "sd_star_dist2d_host": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "sd_star_dist2d_device": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sd_star_dist3d_host": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sd_star_dist3d_device": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sd_edt_prob_device": (_i, [_vp, _i, _i, _i, ctypes.c_double, ctypes.c_double, ctypes.c_double, _i, _vp, _vp]),
    "sd_dist_to_coord_device": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, ctypes.c_double, ctypes.c_double, _vp, _vp]),
    "sd_polygons_to_label_host": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sd_polygons_to_label_device": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sd_polygons_to_label_window_device": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sd_polyhedron_to_label_window_device": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp] + [_i] * 13 + [_vp, _vp]),
    "_LIB_non_maximum_suppression_sparse": (None, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _i, _i, _i, _vp]),
    "sd_nms3d_device": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp]),
    "sd_hiv_pairs_device": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "_LIB_polyhedron_to_label": (None, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sd_polyhedron_to_label_device": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sd_select_candidates_device": (_i, [_vp, _vp, _i, _vp, _vp, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "sd_survivor_positions_device": (_i, [_vp, ctypes.c_longlong, _vp, _vp, _vp]),
    "sd_survivors2d_device": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sd_sorted_rows_device": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sd_sort_scores_desc_device": (_i, [_vp, _i, _vp, _vp, _vp]),
    "sd_bias_act_device": (_i, [_vp, _vp, ctypes.c_longlong, _i, ctypes.c_longlong, _i, _vp]),
    "sd_add_bias_act_device": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, ctypes.c_longlong, _i, _vp]),
    "sd_maxpool_ndhwc_device": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sd_upcat_ndhwc_device": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sd_inside_polyhedron_device": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, ctypes.c_longlong, _i, _vp, _vp]),
    "sd_bias_act_dot_device": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "sd_head_rows_device": (_i, [_vp, _i, _vp, ctypes.c_longlong, _vp, _vp, _i, ctypes.c_float, _vp, _vp]),
    "sd_conv3_packed_floats": (ctypes.c_longlong, [_i, _i, _i]),
    "sd_conv3_pack_weights_host": (_i, [_vp, _i, _i, _i, _vp]),
    "sd_conv3_ndhwc_device": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "sd_conv3_res_ndhwc_device": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "sd_conv3_bf16x6_res_ndhwc_device": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "sd_convg_packed_floats": (ctypes.c_longlong, [_i, _i, _i, _i, _i]),
    "sd_convg_pack_weights_host": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "sd_convg_ndhwc_device": (_i, [_vp] + [_i] * 17 + [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "sd_conv3_bf16x6_packed_floats": (ctypes.c_longlong, [_i, _i, _i]),
    "sd_conv3_bf16x6_pack_weights_host": (_i, [_vp, _i, _i, _i, _vp]),
    "sd_conv3_bf16x6_ndhwc_device": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "sd_conv3_f16x3_packed_floats": (ctypes.c_longlong, [_i, _i, _i]),
    "sd_conv3_f16x3_pack_weights_host": (_i, [_vp, _i, _i, _i, _vp]),
    "sd_conv3_f16x3_ndhwc_device": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sd_conv3_f16x3_res_ndhwc_device": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "sd_conv3_f16x3_dot_ndhwc_device": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sd_dot_combine_device": (_i, [_vp, _i, ctypes.c_longlong, _vp, _i, _vp, _vp]),
    "sd_conv3_f16x3_fmt_ndhwc_device": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sd_conv3_f16x3_rows_device": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, ctypes.c_longlong, _vp, _vp]),
    "sd_conv3_c1x32_split16_device": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "sd_maxpool_split16_ndhwc_device": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sd_split16_pack_device": (_i, [_vp, ctypes.c_longlong, _i, _vp, _vp, _vp]),
    "sd_split16_unpack_device": (_i, [_vp, ctypes.c_longlong, _i, _vp, _vp]),
    "_LIB_non_maximum_suppression_2d": (None, [_vp, _vp, _i, _i, _f, _i, _i, _i, _vp]),
    "_LIB_polygon_to_label": (None, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "_LIB_star_dist": (None, [_vp, _i, _i, _i, _i, _i, _vp]),
    "_LIB_star_dist3d": (None, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
}


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the library is not built."""
    global _lib
    if _lib is None:
        path = os.path.abspath(LIB_PATH)
        if not os.path.exists(path):
            raise NativeError("libstardist_hip.so not built (%s): run `python -m stardist_amd.build` "
                              "or __graft_entry__.build(); there is no CPU fallback" % path)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if it is going to be used
        # it must be the copy that gets loaded first, our library then binds to the already-loaded SONAME.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        l = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class option(object):
    """context manager: `with option("nms3d_volume_bounds", 0): ...` (sd_set_option; restored on exit)"""

    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        self.old = lib().sd_get_option(self.name)
        check(lib().sd_set_option(self.name, self.value))
        return self

    def __exit__(self, *exc):
        lib().sd_set_option(self.name, self.old)
        return False


def require_device():
    if lib().sd_device_count() < 1:
        raise NativeError("no HIP device visible: the stardist_amd natives only run on a GPU "
                          "(there is no CPU fallback)")


def check(rc):
    if rc != 0:
        raise NativeError(lib().sd_last_error().decode(errors="replace"))


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def is_torch(x):
    return type(x).__module__.startswith("torch")


def tptr(t):
    """void* of a contiguous CUDA(=HIP) torch tensor."""
    assert t.is_cuda and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def current_stream(device=None):
    """torch's current stream ON THE GIVEN DEVICE (default: torch's current device)"""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dcall(t, name, *args):
    """Call the device entry point `name` for tensors living on t.device: that device is made current for the call (the
    kernels and the per-device workspace arena follow the current HIP device) and ITS current torch stream is passed as the
    trailing stream argument."""
    import torch
    with torch.cuda.device(t.device):
        check(getattr(lib(), name)(*args, current_stream(t.device)))


def on_device_of(t):
    """context manager: make the tensor's device the current HIP device for the duration of a native call (the workspace
    arena and the kernels of a call live on the current device)"""
    import torch
    return torch.cuda.device(t.device)
