"""ctypes signatures of the C ABI declared in include/mfn_hip.h.

`bind(cdll, prefix)` attaches argtypes/restype to every entry point and returns a namespace
whose attributes are the functions without the prefix.  The product binds libmfn_hip.so with
prefix "mfn_"; tests bind the kernel-logic emulation build with prefix "mfn_emu_".
"""
import ctypes as C
from types import SimpleNamespace

_f = C.c_void_p      # const float* / float*  (device or host address as integer)
_i = C.c_int
_s = C.c_void_p      # hipStream_t
_pi = C.POINTER(C.c_int)

SIGNATURES = {
    # name: (restype, [argtypes])
    "last_error": (C.c_char_p, []),
    "correlation_out_shape": (_i, [_i] * 7 + [_pi, _pi, _pi]),
    "correlation_fwd": (_i, [_f, _f, _f] + [_i] * 10 + [_s]),
    "correlation_workspace_bytes": (C.c_size_t, [_i] * 10),
    "correlation_fwd_ws": (_i, [_f, _f, _f] + [_i] * 10 + [C.c_void_p, C.c_size_t, _s]),
    "correlation_fwd_act": (_i, [_f, _f, _f] + [_i] * 11 + [C.c_void_p, C.c_size_t, _s]),
    "correlation_fwd_into": (_i, [_f, _f, _f, C.c_longlong] + [_i] * 11 + [C.c_void_p, C.c_size_t, _s]),
    "warp_fwd": (_i, [_f, _f, _f] + [_i] * 5 + [_s]),
    "grid_generator_warp": (_i, [_f, _f, _i, _i, _i, _s]),
    "grid_generator_affine": (_i, [_f, _f, _i, _i, _i, _s]),
    "bilinear_sampler_fwd": (_i, [_f, _f, _f] + [_i] * 6 + [_s]),
    "bilinear_sampler_bwd": (_i, [_f] * 5 + [_i] * 8 + [_s]),
    "grid_generator_warp_bwd": (_i, [_f, _f] + [_i] * 4 + [_s]),
    "deform_conv_out_shape": (_i, [_i] * 10 + [_pi, _pi]),
    "deform_conv_workspace_bytes": (C.c_size_t, [_i] * 15),
    "deform_conv_fwd": (_i, [_f] * 5 + [_i] * 15 + [C.c_void_p, C.c_size_t, _s]),
    "deform_conv_shared_fwd": (_i, [_f, _f, C.c_float, C.c_float, _f, _f, _f] + [_i] * 12 + [C.c_void_p, C.c_size_t, _s]),
    "offsets_from_flow": (_i, [_f, _f, _i, _i, _i, _i, C.c_float, C.c_float, _s]),
    "offsets_from_flow_bwd": (_i, [_f, _f, _i, _i, _i, _i, C.c_float, C.c_float, _i, _s]),
    "deform_conv_shared_bwd_workspace_bytes": (C.c_size_t, [_i] * 12),
    "deform_conv_shared_bwd": (_i, [_f, _f, _f, C.c_float, C.c_float, _f, _f, _f, _f, _f] + [_i] * 16 + [C.c_void_p, C.c_size_t, _s]),
    "debug_set_timeline": (_i, [C.c_void_p]),
    "correlation_bwd": (_i, [_f] * 5 + [_i] * 12 + [_s]),
    "warp_bwd": (_i, [_f] * 5 + [_i] * 7 + [_s]),
    "deform_conv_packed_weight_bytes": (C.c_size_t, [_i] * 15),
    "deform_conv_pack_weights": (_i, [_f] + [_i] * 15 + [C.c_void_p, C.c_size_t, C.POINTER(C.c_ulonglong), _s]),
    "deform_conv_fwd_packed": (_i, [_f, _f, C.c_void_p, C.c_size_t, C.c_ulonglong, _f, _f] + [_i] * 15 + [C.c_void_p, C.c_size_t, _s]),
    "deform_conv_shared_fwd_packed": (_i, [_f, _f, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_ulonglong, _f, _f] + [_i] * 12 + [C.c_void_p, C.c_size_t, _s]),
    "upsample_fwd": (_i, [_f, _f] + [_i] * 5 + [_s]),
    "deform_conv_matching_fwd": (_i, [_f, _f, C.c_float, C.c_float, _f, C.c_void_p, C.c_size_t, C.c_ulonglong, _f, _f, _f, _i, _f]
                                 + [_i] * 12 + [C.c_void_p, C.c_size_t, _s]),
    "deform_conv_bwd_workspace_bytes": (C.c_size_t, [_i] * 15),
    "deform_conv_bwd": (_i, [_f] * 8 + [_i] * 19 + [C.c_void_p, C.c_size_t, _s]),
    "conv2d_out_shape": (_i, [_i] * 13 + [_pi, _pi]),
    "conv2d_workspace_bytes": (C.c_size_t, [_i] * 15),
    "conv2d_packed_weight_bytes": (C.c_size_t, [_i] * 15),
    "conv2d_pack_weights": (_i, [_f] + [_i] * 15 + [C.c_void_p, C.c_size_t, C.POINTER(C.c_ulonglong), _s]),
    "conv2d_fwd": (_i, [_f, C.c_longlong, _f, C.c_void_p, C.c_size_t, C.c_ulonglong, _f, _f, C.c_longlong] + [_i] * 18 + [C.c_void_p, C.c_size_t, _s]),
    "upsample_bwd": (_i, [_f, _f] + [_i] * 6 + [_s]),
    "leaky_relu_bwd": (_i, [_f, _f, _f, C.c_size_t, C.c_float, _s]),
    "conv2d_bwd_workspace_bytes": (C.c_size_t, [_i] * 18),
    "conv2d_bwd": (_i, [_f] * 7 + [_i] * 21 + [C.c_void_p, C.c_size_t, _s]),
    "set_arithmetic": (_i, [C.c_char_p, _i]),
    "get_arithmetic": (_i, [C.c_char_p, _pi]),
    "set_tuning": (_i, [C.c_char_p, _i]),
    "get_tuning": (_i, [C.c_char_p, _pi]),
}

# entry points that only the product library exports (HIP runtime plumbing, backward)
PRODUCT_ONLY = {
    "abi_version": (_i, []),
    "version_string": (C.c_char_p, []),
    "graph_begin_capture": (_i, [_s]),
    "graph_end_capture": (_i, [_s, C.POINTER(C.c_void_p)]),
    "graph_launch": (_i, [C.c_void_p, _s]),
    "graph_destroy": (_i, [C.c_void_p]),
    "profile_enable": (_i, [_i]),
    "profile_reset": (_i, []),
    "profile_tag": (_i, [C.c_char_p]),
    "profile_query": (_i, [C.c_char_p, _pi, C.POINTER(C.c_double)]),
    "profile_dump": (_i, [C.c_char_p, _i]),
}


def bind(cdll, prefix="mfn_", product=True):
    ns = SimpleNamespace()
    table = dict(SIGNATURES)
    if product:
        table.update(PRODUCT_ONLY)
    for name, (res, args) in table.items():
        fn = getattr(cdll, prefix + name)  # AttributeError = missing export
        fn.restype = res
        fn.argtypes = args
        setattr(ns, name, fn)
    ns._cdll = cdll
    return ns


def exported_names(product=True):
    names = list(SIGNATURES)
    if product:
        names += list(PRODUCT_ONLY)
    return names
