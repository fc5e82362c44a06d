"""ctypes binding of libroma_hip.so (C ABI declared in include/roma_hip.h).

There is deliberately NO fallback: if the HIP library is missing or does not export a symbol
the import fails loudly (the product path never routes through the CPU oracle).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libroma_hip.so")

ROMA_F32, ROMA_BF16 = 0, 1


class RomaConfig(C.Structure):
    _fields_ = [("coarse_h", C.c_int), ("coarse_w", C.c_int), ("upsample_h", C.c_int), ("upsample_w", C.c_int),
                ("symmetric", C.c_int), ("upsample_preds", C.c_int), ("attenuate_cert", C.c_int),
                ("precision", C.c_int), ("max_batch", C.c_int), ("device", C.c_int)]


_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float

# symbol -> (restype, argtypes); mirrors include/roma_hip.h one to one
SIGNATURES = {
    "roma_last_error": (C.c_char_p, []),
    "roma_version": (C.c_char_p, []),
    "roma_create": (_i, [C.POINTER(RomaConfig), C.POINTER(_vp)]),
    "roma_set_tensor": (_i, [_vp, C.c_char_p, _i, C.POINTER(C.c_int64), _vp, _i]),
    "roma_finalize": (_i, [_vp]),
    "roma_set_option": (_i, [_vp, C.c_char_p, _i]),
    "roma_set_option_f": (_i, [_vp, C.c_char_p, C.c_double]),
    "roma_match": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "roma_debug_fetch": (_l, [_vp, C.c_char_p, _vp, _l]),
    "roma_debug_trace": (_l, [_vp, _i, _vp, _l, C.c_char_p, _l]),
    "roma_debug_inject": (_i, [_vp, C.c_char_p, _vp, _l]),
    "roma_destroy": (_i, [_vp]),
    "roma_tuning": (_i, [C.c_char_p, _i]),
    "roma_profile_enable": (_i, [_i]),
    "roma_profile_report": (_l, [C.c_char_p, _l]),
    "roma_op_local_corr": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_local_corr_window": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _l, _i, _i, _vp]),
    "roma_op_gemm": (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _i, _l, _l, _l, _vp, _vp, _vp, _l, _i, _f, _i, _i, _vp]),
    "roma_op_conv3x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_qkv_scatter_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_layernorm": (_i, [_vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp]),
    "roma_op_layernorm_dt": (_i, [_vp, _i, _vp, _vp, _vp, _l, _i, _f, _i, _vp]),
    "roma_op_gemm_res_bf16": (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _l, _vp]),
    "roma_op_cholesky_solve_t": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "roma_op_gp": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "roma_op_cls_to_flow": (_i, [_vp, _l, _vp, _vp, _l, _vp]),
    "roma_op_resize_bilinear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_refiner_input": (_i, [_vp, _l, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "roma_op_dwconv5x5": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "roma_op_refiner_block": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "roma_op_kde": (_i, [_vp, _l, _i, _f, _i, _vp, _vp]),
    "roma_op_sample_warp_at": (_i, [_vp, _vp, _i, _i, _vp, _l, _vp, _vp, _vp]),
    "roma_op_mutual_nn": (_i, [_vp, _l, _vp, _l, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "roma_op_fb_consistency": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "roma_op_visualize_warp": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "roma_op_multinomial_workspace": (_l, [_l, _l]),
    "roma_op_multinomial": (_i, [_vp, _l, _l, C.c_ulonglong, _vp, _vp, _l, _vp]),
    "roma_op_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "roma_op_tiny_pos_embed": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "roma_op_tiny_matcher_input": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_tiny_update": (_i, [_vp, _i, _vp, _l, _f, _f, _vp, _l, _vp]),
    "roma_op_tiny_final": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "roma_op_maxpool2x2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "roma_op_conv3x3_c3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "roma_op_conv3x3_c3_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "roma_op_refiner_out": (_i, [_vp, _l, _i, _vp, _vp, _vp, _vp, _l, _i, _f, _f, _vp]),
}

_lib = None


class RomaHipError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(roma_amd has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().roma_last_error().decode("utf-8", "replace")


def check(rc: int, exc=RomaHipError):
    if rc != 0:
        raise exc(last_error())
    return rc
