"""ctypes binding of ``libimagdressing_hip.so`` (C ABI in ``include/imagdressing_hip.h``).

There is exactly one backend.  Importing this module never needs a GPU, but every compute call
does: a missing library or a non-gfx950 device raises -- nothing falls back to torch or the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMD_LIB_PATH") or os.path.join(_HERE, "libimagdressing_hip.so")     # override: A/B of two builds

ABI_VERSION = 9

u16p = C.POINTER(C.c_uint16)
f32p = C.POINTER(C.c_float)


class _Sized(C.Structure):
    """Parameter blocks of ABI v8 start with ``struct_bytes`` = sizeof(the struct as THIS binding lays it out); the library refuses any
    other value (a binding that mirrors another header version would otherwise hand it a truncated struct)."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_bytes = C.sizeof(self)


class HeadsDest(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("kind", C.c_int), ("DP", C.c_int), ("L", C.c_int), ("scale", C.c_float)]


class ConvGemmParams(_Sized):
    _fields_ = [("struct_bytes", C.c_uint32), 
        ("x", C.c_void_p), ("w", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("Cin", C.c_int), ("taps", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int),
        ("Wout", C.c_int), ("stride", C.c_int), ("ups", C.c_int),
        ("x_pix_stride", C.c_int), ("out_ld", C.c_int), ("res_ld", C.c_int),
        ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("rowvec_stride", C.c_int),
        ("res", C.c_void_p), ("out_scale", C.c_float), ("act", C.c_int), ("out_f32", C.c_int),
        ("mode", C.c_int), ("hC", C.c_int), ("hH", C.c_int), ("hD", C.c_int),
        ("hd", HeadsDest * 3), ("dtype", C.c_int), ("split_k", C.c_int), ("splitk_ws", C.c_void_p),
        ("x_bytes", C.c_uint32), ("w_bytes", C.c_uint32), ("flags", C.c_int),
        ("gn_a", C.c_void_p), ("gn_b", C.c_void_p), ("gn_silu", C.c_int), ("pad_br_only", C.c_int),
        ("splitk_counters", C.c_void_p),
        ("gn_stats_out", C.c_void_p), ("gn_stats_groups", C.c_int),
        ("gn_in_partial", C.c_void_p), ("gn_in_gamma", C.c_void_p), ("gn_in_beta", C.c_void_p),
        ("gn_in_nparts", C.c_int), ("gn_in_groups", C.c_int), ("gn_in_silu", C.c_int), ("gn_in_eps", C.c_float),
        ("gn_out_gamma", C.c_void_p), ("gn_out_beta", C.c_void_p), ("gn_out_eps", C.c_float), ("gn_out_silu", C.c_int), ("gn_out_groups", C.c_int), ("res_rows", C.c_int),
    ]


class FfParams(_Sized):
    _fields_ = [("struct_bytes", C.c_uint32), ("x", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("out", C.c_void_p),
                ("M", C.c_int), ("C", C.c_int), ("inner", C.c_int), ("x_ld", C.c_int), ("out_ld", C.c_int), ("ln", C.c_int),
                ("ln_eps", C.c_float), ("dtype", C.c_int)]


class AttnParams(_Sized):
    _fields_ = [("struct_bytes", C.c_uint32), 
        ("q", C.c_void_p), ("k1", C.c_void_p), ("v1t", C.c_void_p), ("k2", C.c_void_p), ("v2t", C.c_void_p),
        ("scale2", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int), ("H", C.c_int), ("N", C.c_int), ("D", C.c_int),
        ("L1", C.c_int), ("L1P", C.c_int), ("kv1_bdiv", C.c_int),
        ("L2", C.c_int), ("L2P", C.c_int), ("kv2_bdiv", C.c_int),
        ("out_ld", C.c_int), ("dtype", C.c_int), ("flags", C.c_int), ("causal", C.c_int), ("k_pad_one", C.c_int),
        ("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("proj_res", C.c_void_p), ("proj_out", C.c_void_p),
        ("proj_res_ld", C.c_int), ("proj_out_ld", C.c_int), ("proj_counters", C.c_void_p),
        ("out_dup", C.c_void_p), ("phase2_out", C.c_void_p), ("phase2_rows", C.c_int),
    ]


class GroupNormParams(_Sized):
    _fields_ = [("struct_bytes", C.c_uint32), 
        ("x", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("partial", C.c_void_p),
        ("B", C.c_int), ("HW", C.c_int), ("C", C.c_int), ("G", C.c_int), ("x_ld", C.c_int), ("y_ld", C.c_int),
        ("eps", C.c_float), ("silu", C.c_int), ("dtype", C.c_int), ("nparts", C.c_int),
    ]


class LayerNormParams(_Sized):
    _fields_ = [("struct_bytes", C.c_uint32), 
        ("x", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("rows", C.c_int), ("C", C.c_int), ("x_ld", C.c_int), ("y_ld", C.c_int), ("eps", C.c_float), ("dtype", C.c_int),
    ]


class DdimParams(_Sized):
    _fields_ = [("struct_bytes", C.c_uint32), 
        ("z", C.c_void_p), ("eps", C.c_void_p), ("x_next", C.c_void_p),
        ("B", C.c_int), ("HW", C.c_int),
        ("guidance", C.c_float), ("sqrt_a_t", C.c_float), ("sqrt_1m_a_t", C.c_float),
        ("sqrt_a_prev", C.c_float), ("sqrt_1m_a_prev", C.c_float),
        ("mask", C.c_void_p), ("z_img", C.c_void_p), ("noise", C.c_void_p),
        ("sqrt_a_next", C.c_float), ("sqrt_1m_a_next", C.c_float), ("dtype", C.c_int),
        ("coefs", C.c_void_p), ("var_noise", C.c_void_p), ("sigma", C.c_float),
    ]


# every symbol include/imagdressing_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "imd_abi_version": (C.c_int, []),
    "imd_last_error": (C.c_char_p, []),
    "imd_device_check": (C.c_int, [C.c_int]),
    "imd_conv_gemm": (C.c_int, [C.POINTER(ConvGemmParams), C.c_int, C.c_void_p]),
    "imd_conv_gemm_auto_cfg": (C.c_int, [C.c_int, C.c_int]),
    "imd_conv_gemm_auto_split": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "imd_attention": (C.c_int, [C.POINTER(AttnParams), C.c_void_p]),
    "imd_attention_dup_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "imd_attention_phase_split_supported": (C.c_int, [C.c_int]),
    "imd_attention_fp8": (C.c_int, [C.POINTER(AttnParams), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "imd_attn_quantize_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "imd_set_tuning": (C.c_int, [C.c_int, C.c_int]),
    "imd_get_tuning": (C.c_int, [C.c_int]),
    "imd_attn_padded_dims": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "imd_groupnorm": (C.c_int, [C.POINTER(GroupNormParams), C.c_void_p]),
    "imd_groupnorm_coeffs": (C.c_int, [C.POINTER(GroupNormParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    "imd_conv_patch_supported": (C.c_int, [C.POINTER(ConvGemmParams)]),
    "imd_conv_patch2_supported": (C.c_int, [C.POINTER(ConvGemmParams)]),
    "imd_conv_patch3_supported": (C.c_int, [C.POINTER(ConvGemmParams)]),
    "imd_conv_patch4_supported": (C.c_int, [C.POINTER(ConvGemmParams)]),
    "imd_conv_img_supported": (C.c_int, [C.POINTER(ConvGemmParams)]),
    "imd_conv_patch_stats_parts": (C.c_int, [C.POINTER(ConvGemmParams)]),
    "imd_conv_gemm_stats_parts": (C.c_int, [C.POINTER(ConvGemmParams), C.c_int]),
    "imd_conv_gemm_gn_out_supported": (C.c_int, [C.POINTER(ConvGemmParams)]),
    "imd_row_linear_gn_in_supported": (C.c_int, [C.POINTER(ConvGemmParams), C.c_int]),
    "imd_gemm_dma_supported": (C.c_int, [C.POINTER(ConvGemmParams)]),
    "imd_row_linear": (C.c_int, [C.POINTER(ConvGemmParams), C.c_int, C.c_float, C.c_void_p]),
    "imd_row_linear_supported": (C.c_int, [C.POINTER(ConvGemmParams)]),
    "imd_ff_geglu": (C.c_int, [C.POINTER(FfParams), C.c_void_p]),
    "imd_groupnorm_workspace_floats": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "imd_layernorm": (C.c_int, [C.POINTER(LayerNormParams), C.c_void_p]),
    "imd_softmax_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "imd_ddim_cfg_step": (C.c_int, [C.POINTER(DdimParams), C.c_void_p]),
    "imd_timestep_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "imd_add": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "imd_embed_tokens": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "imd_vit_assemble": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "imd_lincomb": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_long, C.c_void_p]),
    "imd_copy2d": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p]),
    "imd_concat2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p]),
    "imd_concat2_gn_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int, C.c_void_p]),
    "imd_groupnorm_parts": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "imd_f32_to_16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]),
}

_lib = None


class ImdError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImdError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or imagdressing_amd/csrc/build.sh).  imagdressing_amd has no CPU / torch fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        if lib.imd_abi_version() != ABI_VERSION:
            raise ImdError(f"ABI version mismatch: library reports {lib.imd_abi_version()}, binding expects {ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc: int):
    if rc != 0:
        raise ImdError(load().imd_last_error().decode("utf-8", "replace"))
