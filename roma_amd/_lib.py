"""ctypes binding of libroma_hip.so / libroma_hip_f16.so (C ABI declared in include/roma_hip.h).

The 16-bit storage format is a build-time property of the library (csrc/common.h): `load("bf16")` (the default) binds
libroma_hip.so, `load("f16")` binds libroma_hip_f16.so - the same sources compiled for IEEE binary16, the reference's
default amp_dtype.  Both export the same symbols; f32-only operators live in either.

There is deliberately NO fallback: if the HIP library is missing or does not export a symbol
the import fails loudly (the product path never routes through the CPU oracle).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ROMA_LIB_DIR: A/B runs of tools/ against another BUILD of the same two libraries (e.g. the previous commit's); the product
# default is the in-tree build next to this file.  There is still no fallback: a missing library raises.
_HERE = os.environ.get("ROMA_LIB_DIR", _HERE)
LIB_PATH = os.path.join(_HERE, "libroma_hip.so")
LIB_PATHS = {"bf16": LIB_PATH, "f16": os.path.join(_HERE, "libroma_hip_f16.so")}

ROMA_F32, ROMA_BF16, ROMA_F16, ROMA_MIXED = 0, 1, 2, 3
H16_CODE = {"bf16": ROMA_BF16, "f16": ROMA_F16}


class RomaConfig(C.Structure):
    _fields_ = [("coarse_h", C.c_int), ("coarse_w", C.c_int), ("upsample_h", C.c_int), ("upsample_w", C.c_int),
                ("symmetric", C.c_int), ("upsample_preds", C.c_int), ("attenuate_cert", C.c_int),
                ("precision", C.c_int), ("max_batch", C.c_int), ("device", C.c_int)]


class RomaForwardArgs(C.Structure):
    """roma_forward_args_t (include/roma_hip.h)"""
    _fields_ = [("upsample", C.c_int), ("symmetric", C.c_int), ("scale_factor", C.c_double),
                ("seed_flow", C.c_void_p), ("seed_cert", C.c_void_p), ("seed_h", C.c_int), ("seed_w", C.c_int),
                ("flow", C.c_void_p * 5), ("cert", C.c_void_p * 5), ("feat", C.c_void_p * 5)]


_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float

# symbol -> (restype, argtypes); mirrors include/roma_hip.h one to one
SIGNATURES = {
    "roma_last_error": (C.c_char_p, []),
    "roma_version": (C.c_char_p, []),
    "roma_h16_format": (_i, []),
    "roma_abi_stamp": (_i, []),
    "roma_self_check": (_i, []),
    "roma_create": (_i, [C.POINTER(RomaConfig), C.POINTER(_vp)]),
    "roma_set_tensor": (_i, [_vp, C.c_char_p, _i, C.POINTER(C.c_int64), _vp, _i]),
    "roma_finalize": (_i, [_vp]),
    "roma_set_option": (_i, [_vp, C.c_char_p, _i]),
    "roma_set_option_f": (_i, [_vp, C.c_char_p, C.c_double]),
    "roma_match": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "roma_forward": (_i, [_vp, _i, _vp, _vp, C.POINTER(RomaForwardArgs), _vp]),
    "roma_debug_fetch": (_l, [_vp, C.c_char_p, _vp, _l]),
    "roma_debug_trace": (_l, [_vp, _i, _vp, _l, C.c_char_p, _l]),
    "roma_debug_inject": (_i, [_vp, C.c_char_p, _vp, _l]),
    "roma_destroy": (_i, [_vp]),
    "roma_tuning": (_i, [C.c_char_p, _i]),
    "roma_debug_gemm_trace": (_l, [_vp, _l]),
    "roma_vit_forward": (_i, [_vp, _vp]),
    "roma_op_convert_from_bf16": (_i, [_vp, _vp, _l, _vp]),
    "roma_profile_enable": (_i, [_i]),
    "roma_profile_report": (_l, [C.c_char_p, _l]),
    "roma_op_local_corr": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_local_corr_window": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _l, _i, _i, _vp]),
    "roma_op_gemm": (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _i, _l, _l, _l, _vp, _vp, _vp, _l, _i, _f, _i, _i, _vp]),
    "roma_op_conv3x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_conv3x3_slab": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
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
    "roma_op_refiner_block_final": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "roma_op_refiner_apply_delta": (_i, [_vp, _vp, _vp, _l, _f, _f, _vp]),
    "roma_op_kde": (_i, [_vp, _l, _i, _f, _i, _vp, _vp]),
    "roma_op_sample_warp_at": (_i, [_vp, _vp, _i, _i, _vp, _l, _vp, _vp, _vp]),
    "roma_op_mutual_nn": (_i, [_vp, _l, _vp, _l, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "roma_op_mutual_nn_count": (_i, [_vp, _l, _vp, _l, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "roma_op_mutual_nn_fill": (_i, [_vp, _l, _vp, _l, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "roma_op_fb_consistency": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "roma_op_visualize_warp": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "roma_op_multinomial_workspace": (_l, [_l, _l]),
    "roma_op_multinomial": (_i, [_vp, _l, _l, C.c_ulonglong, _vp, _vp, _l, _vp]),
    "roma_op_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "roma_op_tiny_pos_embed": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_gray_instnorm": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "roma_op_conv2d_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_avgpool_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "roma_op_add3": (_i, [_vp, _vp, _vp, _vp, _l, _vp]),
    "roma_op_tiny_matcher_input": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_tiny_update": (_i, [_vp, _i, _vp, _l, _f, _f, _vp, _l, _vp]),
    "roma_op_tiny_final": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "roma_op_maxpool2x2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "roma_op_pool_proj": (_i, [_vp, _vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "roma_op_conv3x3_c3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "roma_op_conv3x3_c3_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "roma_op_refiner_out": (_i, [_vp, _l, _i, _vp, _vp, _vp, _vp, _l, _i, _f, _f, _vp]),
}

_libs = {}


class RomaHipError(RuntimeError):
    pass


def load(fmt: str = "bf16"):
    """dlopen the library that stores `fmt` ("bf16" | "f16") and bind every declared symbol (raises if anything is missing)."""
    if fmt in _libs:
        return _libs[fmt]
    path = LIB_PATHS[fmt]
    if not os.path.exists(path):
        raise ImportError(f"{path} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(roma_amd has no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.roma_h16_format() != H16_CODE[fmt]:
        raise ImportError(f"{path} does not store {fmt} (roma_h16_format() = {lib.roma_h16_format()})")
    lib.h16 = fmt
    _libs[fmt] = lib
    return lib


def fmt_of(dtype) -> str:
    """library format for a torch dtype: float16 -> "f16", everything else (bfloat16, float32) -> "bf16"."""
    return "f16" if str(dtype) == "torch.float16" else "bf16"


def last_error(lib=None) -> str:
    return (lib or load()).roma_last_error().decode("utf-8", "replace")


def check(rc: int, exc=RomaHipError, lib=None):
    if rc != 0:
        raise exc(last_error(lib))
    return rc
