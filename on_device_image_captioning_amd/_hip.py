"""ctypes binding of libodic_hip.so (include/odic_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a kernel rejects its
arguments, a RuntimeError is raised.  Build the library with `python -c "import __graft_entry__ as g;
g.build()"` or `make -C on_device_image_captioning_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libodic_hip.so")

F32, BF16, FP8, F16, H2 = 0, 1, 2, 3, 4
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
ABI_VERSION = 15

_ERR = {-1: "ODIC_EINVAL (bad shape / alignment / enum)", -2: "ODIC_ENULL (required pointer is NULL)",
        -3: "ODIC_EUNSUPPORTED"}


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("out", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int64), ("ldw", C.c_int64), ("ldr", C.c_int64), ("ldc", C.c_int64),
                ("batch", C.c_int32),
                ("strideA", C.c_int64), ("strideW", C.c_int64), ("strideBias", C.c_int64),
                ("strideR", C.c_int64), ("strideC", C.c_int64),
                ("alpha", C.c_float), ("act", C.c_int32), ("bias_axis", C.c_int32),
                ("in_dtype", C.c_int32), ("out_dtype", C.c_int32), ("tile_cfg", C.c_int32),
                ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float), ("workspace", C.c_void_p),
                ("col_scale", C.c_void_p), ("out_scale", C.c_float),
                ("out16", C.c_void_p), ("ld16", C.c_int64), ("stats_out", C.c_void_p), ("ln_stats", C.c_void_p),
                ("a_ln", C.c_void_p), ("ld_aln", C.c_int64)]


class BeamState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("tokens", "logprobs", "anc", "cumul", "n_elem", "has_eos", "row_valid", "next_tok",
                 "pos", "done", "ctr")]


class EmbedArgs(C.Structure):
    """odic_embed_args: the next position's input embedding as the tail of the launch that chooses the words."""
    _fields_ = [("embed", C.c_void_p), ("pos_table", C.c_void_p), ("y", C.c_void_p), ("ldy", C.c_int64),
                ("d", C.c_int32), ("scale", C.c_float), ("pos_rows", C.c_int32)]


_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

_SIGNATURES = {
    "odic_abi_version": (C.c_int, []),
    "odic_build_info": (C.c_char_p, []),
    "odic_gemm": (C.c_int, [C.POINTER(GemmArgs), _P]),
    "odic_layernorm": (C.c_int, [_P, _I64, _P, _P, _P, _I32, _I32, _F, _I32, _P]),
    "odic_cast_f32_to_bf16": (C.c_int, [_P, _I64, _P, _I64, _I32, _I32, _P]),
    "odic_cast_f32_to_h2": (C.c_int, [_P, _I64, _P, _I64, _I32, _I32, _P]),
    "odic_patch_merge_layernorm": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _F, _I32, _P]),
    "odic_patch_embed": (C.c_int, [_P] * 6 + [_I32] * 6 + [_F, _P]),
    "odic_resize_bilinear_normalize": (C.c_int, [_P, _I32, _I32, _I64, _P, _P, _I32, _P, _P, _I32, _P, _P, _I32,
                                                 C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "odic_window_attention": (C.c_int, [_P, _P, _P, _P] + [_I32] * 6 + [_F, _I32, _P]),
    "odic_swin_qkv_attention": (C.c_int, [_P, _I64, _P, _P, _P, _P] + [_I32] * 6 + [_F, _F, _P]),
    "odic_stcexp_normalize": (C.c_int, [_P, _P, _P, _I32, _P, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _I32, _F, _F, _F, _I32,
                                        _P]),
    "odic_selector_mix": (C.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I32, _I32, _P]),
    "odic_copy": (C.c_int, [_P, _P, _I64, _P]),
    "odic_dec_embed": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _F, _P]),
    "odic_dynexp_step": (C.c_int, [_P, _I64, _P, _P] + [_P] * 7 + [_P, _P, _P, _P, _I64, _P, _I64] + [_I32] * 4 + [_F, _P]),
    "odic_cross_attn_step": (C.c_int, [_P, _I64, _P, _I64, _I32, _I32, _P, _P, _P, _I64] + [_I32] * 5 + [_P]),
    "odic_logsoftmax_topk": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _I32, _I32, _I32, _P]),
    "odic_logsoftmax_sample": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _I32, _I32, _I32, C.c_uint64, _P, _P]),
    "odic_ensemble_logprobs": (C.c_int, [C.POINTER(C.c_void_p), _I32, _I64, _P, _I64, _I32, _I32, _P]),
    "odic_topk_rows": (C.c_int, [_P, _I64, _P, _P, _I32, _I32, _I32, _P]),
    "odic_beam_step": (C.c_int, [_P, _P, C.POINTER(BeamState), C.POINTER(EmbedArgs), _I32, _I32, _I32, _I64, _P]),
    "odic_beam_search_step": (C.c_int, [_P, _I64, _I32, C.POINTER(BeamState), C.POINTER(EmbedArgs), _I32, _I32, _I32,
                                        _I64, _P]),
    "odic_beam_finalize": (C.c_int, [C.POINTER(BeamState), _P, _P, _I32, _I32, _P]),
    "odic_beam_finalize_best": (C.c_int, [C.POINTER(BeamState), _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "odic_beam_reset": (C.c_int, [C.POINTER(BeamState), C.POINTER(EmbedArgs), _I32, _I32, _I32, _I64, _P]),
}

#: every symbol include/odic_hip.h declares
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """Load (once) and return the library; raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built and there is no CPU fallback. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` from the repository root.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    got = lib.odic_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libodic_hip.so ABI {got} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status == 0:
        return
    if status < 0:
        raise RuntimeError(f"{what}: rejected with {_ERR.get(status, status)}")
    raise RuntimeError(f"{what}: HIP launch failed with hipError_t {status}")
