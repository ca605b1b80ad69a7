"""ctypes binding of libskimi.so (the C-ABI declared in include/skimi.h).

The library is the product: there is no Python / torch fallback.  If the shared object is
missing or a symbol is absent the import fails loudly, and every wrapper raises
`SkimiError` with the library's own message on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libskimi.so"

F32, BF16 = 0, 1
F16 = 4          # IEEE binary16: operands / activations of PREC_F16
BF16X3_REC = 2   # skimi_gemm_desc.a_dtype: A already split into bf16x3 records
FP8MX = 3        # skimi_gemm_fp8 out_dtype: the result as MXFP8 payload + scales
PREC_BF16, PREC_BF16X3, PREC_FP8 = 0, 1, 2   # PREC_FP8: VGGT aggregator only (MXFP8 qkv / proj / fc1 / fc2)
PREC_F16 = 3     # fp16 operands on the f16 MFMA (Linears of the aggregator blocks + patch embed; skimi_gemm with fp16 / fp32 operands)
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SILU = 0, 1, 2, 3


class SkimiError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    """Mirror of `skimi_gemm_desc` (include/skimi.h) — field order and types must match."""

    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("A", C.c_void_p), ("W", C.c_void_p),
        ("a_dtype", C.c_int32), ("w_dtype", C.c_int32),
        ("lda", C.c_int64), ("ldw", C.c_int64),
        ("prec", C.c_int32),
        ("a_mode", C.c_int32),
        ("cN", C.c_int32), ("cH", C.c_int32), ("cW", C.c_int32), ("cC", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("dil", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("bias", C.c_void_p), ("gamma", C.c_void_p), ("resid", C.c_void_p),
        ("resid_dtype", C.c_int32),
        ("ldr", C.c_int64),
        ("resid_rows_per_batch", C.c_int32),
        ("resid_batch_stride", C.c_int64),
        ("resid_row_off", C.c_int64),
        ("act", C.c_int32),
        ("resid2", C.c_void_p), ("ldr2", C.c_int64), ("post_act", C.c_int32),
        ("out_rows_per_batch", C.c_int32), ("out_batch_stride", C.c_int64), ("out_row_off", C.c_int64),
        ("out", C.c_void_p), ("out2", C.c_void_p),
        ("out_dtype", C.c_int32),
        ("ldo", C.c_int64), ("ldo2", C.c_int64),
        ("store_mode", C.c_int32), ("ps_s", C.c_int32), ("ps_C", C.c_int32),
        ("splitk_scratch", C.c_void_p),
        ("splitk_scratch_bytes", C.c_uint64),
        ("force_splitk", C.c_int32), ("splitk_scratch_zeroed", C.c_int32),
        ("W_split", C.c_void_p), ("x3_scratch", C.c_void_p), ("x3_scratch_bytes", C.c_uint64),
        ("out_records", C.c_void_p),
    ]


# every exported symbol of include/skimi.h: name -> (restype, argtypes)
_vp = C.c_void_p
_SIGNATURES = {
    "skimi_last_error": (C.c_char_p, []),
    "skimi_version": (C.c_int, []),
    "skimi_sizeof_gemm_desc": (C.c_int, []),
    "skimi_device_count": (C.c_int, []),
    "skimi_profile_start": (C.c_int, [C.c_int32, C.c_int64]),
    "skimi_profile_stop": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]),
    "skimi_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "skimi_quant_mx": (C.c_int, [_vp, C.c_int32, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, _vp]),
    "skimi_layernorm_mx": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, C.c_float, _vp, _vp, _vp]),
    "skimi_gemm_fp8": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_int64,
                                 _vp, C.c_int32, C.c_int64, _vp, _vp]),
    "skimi_split_planes": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, _vp]),
    "skimi_split_records": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int32, _vp, _vp]),
    "skimi_resample_u8": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, C.c_int32, C.c_int64, _vp, _vp, C.c_int32, _vp]),
    "skimi_u8_hwc_to_f32_chw": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_float, _vp]),
    "skimi_conv3x3_n32_pack": (C.c_int, [_vp, _vp, C.c_int32, _vp]),
    "skimi_conv3x3_n32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "skimi_layernorm": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, C.c_float, _vp,
                                  C.c_int32, C.c_int64, _vp]),
    "skimi_qknorm_rope": (C.c_int, [_vp, C.c_int32, C.c_int64, C.c_int32, _vp, _vp, _vp, _vp, C.c_float,
                                    _vp, _vp, _vp, C.c_int32, _vp]),
    "skimi_attention": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "skimi_attention_out": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "skimi_vp3d_create": (_vp, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                C.c_int32, C.c_int32]),
    "skimi_vp3d_destroy": (None, [_vp]),
    "skimi_vp3d_set_weight": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int64]),
    "skimi_vp3d_finalize": (C.c_int, [_vp, C.c_int32]),
    "skimi_vp3d_receptive_field": (C.c_int32, [_vp]),
    "skimi_vp3d_workspace_bytes": (C.c_size_t, [_vp, C.c_int32, C.c_int32]),
    "skimi_vp3d_forward": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.c_size_t, _vp]),
    "skimi_pose_to_cameras": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "skimi_unproject_depth": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "skimi_triangulate_dlt": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int32, C.c_int32, _vp]),
    "skimi_vggt_create": (_vp, [_vp]),
    "skimi_vggt_destroy": (None, [_vp]),
    "skimi_vggt_set_weight": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int64, C.c_int32]),
    "skimi_vggt_finalize": (C.c_int, [_vp]),
    "skimi_vggt_set_pos_embed": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, C.c_int32]),
    "skimi_vggt_rope_positions": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "skimi_vggt_workspace_bytes": (C.c_size_t, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "skimi_vggt_forward": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp,
                                     _vp, C.c_size_t, _vp]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def lib() -> C.CDLL:
    """Load libskimi.so once; raise if it (or any declared symbol) is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SkimiError(
            f"{LIB_PATH} not found: build it with `python -m skiing_analysis_pytorch_amd.build` "
            "(or __graft_entry__.build()). There is no fallback path."
        )
    # torch ships its own libamdhip64; load it first so libskimi binds to the SAME HIP runtime
    # instance (streams and device pointers are shared with torch tensors).
    import torch  # noqa: F401

    handle = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:  # pragma: no cover
            raise SkimiError(f"libskimi.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if handle.skimi_sizeof_gemm_desc() != C.sizeof(GemmDesc):
        raise SkimiError(f"skimi_gemm_desc is {handle.skimi_sizeof_gemm_desc()} bytes in libskimi.so but "
                         f"{C.sizeof(GemmDesc)} in _lib.py: rebuild the library (python -m skiing_analysis_pytorch_amd.build)")
    _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().skimi_last_error()
        raise SkimiError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int | None:
    """Device/host address of a torch tensor (None passes NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
