"""Thin torch-tensor wrappers over the generic C-ABI ops (skimi_gemm, skimi_layernorm,
skimi_qknorm_rope, skimi_attention).  Tensors provide device memory and the stream only; all
arithmetic happens in libskimi.so.  Used by the parity tests and by the Python host side."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import ACT_NONE, BF16, F16, F32, PREC_BF16, PREC_BF16X3, GemmDesc, check, lib, ptr


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SkimiError("libskimi ops need device (HBM) tensors; got a CPU tensor")


def gemm(a, w, *, prec=PREC_BF16X3, bias=None, gamma=None, resid=None, act=ACT_NONE, out=None,
         out_dtype=torch.float32, conv=None, resid_map=None, pixel_shuffle=None, splitk_scratch=None,
         force_splitk=0, M=None, lda=None, w_split=None, x3_scratch=None, a_records=None, out_records=None,
         records_only=False, post_act=ACT_NONE):
    """out = epilogue(gather(a) @ w.T).  a: [rows, lda] (f32|bf16), w: [N, K].

    conv = dict(N,H,W,C,KH,KW,stride,pad,dil,OH,OW) selects the implicit-im2col gather;
    resid_map = (rows_per_batch, batch_stride, row_off); pixel_shuffle = (s, Cout, N, H, W).
    out_records = a `records_buffer(M, N)`: the result also as bf16x3 records (+ zero page) for a
    following `gemm(None, ..., a_records=that buffer)` (its [rows, C] shape comes through `conv`, or
    `M` and `lda` for plain rows); records_only: no fp32 result at all (returns None).
    post_act: activation applied after the residual add."""
    _require_cuda(w, bias, gamma, resid, out)
    d = GemmDesc()
    N, K = w.shape
    d.N, d.K = N, K
    d.W, d.w_dtype = ptr(w), _dt(w)
    if a_records is not None:
        _require_cuda(a_records)
        d.A, d.a_dtype = ptr(a_records), _lib.BF16X3_REC
        d.lda = lda if lda is not None else (conv["C"] if conv is not None else K)
    else:
        _require_cuda(a)
        d.A, d.a_dtype = ptr(a), _dt(a)
        d.lda = lda if lda is not None else a.stride(-2) if a.dim() >= 2 else a.shape[-1]
    dev = w.device
    d.ldw = w.stride(0)
    d.prec = prec
    if conv is not None:
        d.a_mode = 2 if conv.get("slice_major") else 1   # 2: weights packed [Cout][Cin/32][ky][kx][32]
        d.cN, d.cH, d.cW, d.cC = conv["N"], conv["H"], conv["W"], conv["C"]
        d.KH, d.KW, d.stride, d.pad, d.dil = conv["KH"], conv["KW"], conv["stride"], conv["pad"], conv["dil"]
        d.OH, d.OW = conv["OH"], conv["OW"]
        d.M = d.cN * d.OH * d.OW
    else:
        d.M = M if M is not None else a.shape[0]
    if records_only:
        assert out is None and out_records is not None and pixel_shuffle is None
        d.out_dtype = _lib.F32
    elif pixel_shuffle is not None:
        s, cout, n_img, h, w_ = pixel_shuffle
        d.store_mode, d.ps_s, d.ps_C = 1, s, cout
        d.cN, d.cH, d.cW = n_img, h, w_
        if out is None:
            out = torch.empty((n_img, h * s, w_ * s, cout), dtype=out_dtype, device=dev)
        d.ldo = out.stride(-2)
    else:
        if out is None:
            out = torch.empty((d.M, N), dtype=out_dtype, device=dev)
        d.ldo = out.stride(-2)
    if not records_only:
        d.out, d.out_dtype = ptr(out), _dt(out)
    d.bias, d.gamma, d.resid = ptr(bias), ptr(gamma), ptr(resid)
    if resid is not None:
        d.ldr = resid.stride(-2)
        d.resid_dtype = _dt(resid)
    if resid_map is not None:
        d.resid_rows_per_batch, d.resid_batch_stride, d.resid_row_off = resid_map
    d.act = act
    d.post_act = post_act
    if splitk_scratch is not None:
        d.splitk_scratch = ptr(splitk_scratch)
        d.splitk_scratch_bytes = splitk_scratch.numel() * splitk_scratch.element_size()
    d.force_splitk = force_splitk
    if w_split is not None:
        d.W_split = ptr(w_split)
        if a_records is not None:   # the zero page behind the records
            d.x3_scratch = ptr(a_records) + a_records.numel() * 2 - 256
            d.x3_scratch_bytes = 256
        else:
            d.x3_scratch = ptr(x3_scratch)
            d.x3_scratch_bytes = x3_scratch.numel() * x3_scratch.element_size()
    if out_records is not None:
        _require_cuda(out_records)
        d.out_records = ptr(out_records)
    check(lib().skimi_gemm(C.byref(d), _lib.current_stream()), "skimi_gemm")
    return out


def layernorm(x, gamma=None, beta=None, eps=1e-5, *, x2=None, out_dtype=torch.float32):
    _require_cuda(x, x2, gamma, beta)
    rows = x.numel() // x.shape[-1]
    Cc = x.shape[-1] * (2 if x2 is not None else 1)
    out = torch.empty((*x.shape[:-1], Cc), dtype=out_dtype, device=x.device)
    check(lib().skimi_layernorm(ptr(x), ptr(x2), x.stride(-2), rows, Cc, ptr(gamma), ptr(beta), eps, ptr(out),
                                _dt(out), Cc, _lib.current_stream()), "skimi_layernorm")
    return out


def qknorm_rope_(qkv, heads, qn_w=None, qn_b=None, kn_w=None, kn_b=None, eps=1e-5, pos=None, rope_cos=None,
                 rope_sin=None):
    """in place on qkv [tokens, 3*heads*64]"""
    _require_cuda(qkv, pos, rope_cos, rope_sin)
    tokens = qkv.numel() // (3 * heads * 64)
    npos = rope_cos.shape[0] if rope_cos is not None else 0
    check(lib().skimi_qknorm_rope(ptr(qkv), _dt(qkv), tokens, heads, ptr(qn_w), ptr(qn_b), ptr(kn_w), ptr(kn_b),
                                  eps, ptr(pos), ptr(rope_cos), ptr(rope_sin), npos, _lib.current_stream()),
          "skimi_qknorm_rope")
    return qkv


def attention(qkv, batch, seq, heads, head_dim, out_dtype=None):
    """qkv: [batch*seq, 3*heads*head_dim] -> [batch*seq, heads*head_dim] (same dtype; out_dtype=torch.float16 with
    bf16 qkv: the result rows as fp16, PREC_F16's proj operand; out_dtype="fp8mx" with bf16 qkv, head_dim 64: the result
    rows as MXFP8 -> (payload uint8 [rows, Kp], scales uint8 [rows, Kp/32]), PREC_FP8's proj operand)"""
    _require_cuda(qkv)
    if isinstance(out_dtype, str):
        assert out_dtype == "fp8mx", out_dtype
        rows, Kp = batch * seq, (heads * head_dim + 127) // 128 * 128
        buf = torch.zeros(rows * (Kp + Kp // 32), dtype=torch.uint8, device=qkv.device)
        check(lib().skimi_attention_out(ptr(qkv), ptr(buf), _dt(qkv), _lib.FP8MX, batch, seq, heads, head_dim,
                                        _lib.current_stream()), "skimi_attention_out")
        return buf[:rows * Kp].view(rows, Kp), buf[rows * Kp:].view(rows, Kp // 32)
    out = torch.empty((batch * seq, heads * head_dim), dtype=out_dtype or qkv.dtype, device=qkv.device)
    if out_dtype is None or out_dtype == qkv.dtype:
        check(lib().skimi_attention(ptr(qkv), ptr(out), _dt(qkv), batch, seq, heads, head_dim, _lib.current_stream()),
              "skimi_attention")
    else:
        check(lib().skimi_attention_out(ptr(qkv), ptr(out), _dt(qkv), _dt(out), batch, seq, heads, head_dim,
                                        _lib.current_stream()), "skimi_attention_out")
    return out


def conv3x3_n32(x, w, bias=None, relu=False):
    """Direct fp32-accurate 3x3 conv (stride 1, pad 1) to 32 channels.  x: fp32 [F, H, W, C] channels-last,
    w: the reference's [32, C, 3, 3] fp32 weight -> fp32 [F, H, W, 32]."""
    _require_cuda(x, w, bias)
    F_, H, W_, Cc = x.shape
    assert w.shape == (32, Cc, 3, 3) and Cc % 32 == 0
    planes = split_planes(x.reshape(-1, Cc).contiguous())
    packed = torch.empty(2 * 32 * Cc * 9, dtype=torch.bfloat16, device=x.device)
    st = _lib.current_stream()
    check(lib().skimi_conv3x3_n32_pack(ptr(w.contiguous()), ptr(packed), Cc, st), "skimi_conv3x3_n32_pack")
    out = torch.empty((F_, H, W_, 32), dtype=torch.float32, device=x.device)
    check(lib().skimi_conv3x3_n32(ptr(planes[0]), ptr(planes[1]), ptr(packed), ptr(bias), ptr(out), F_, H, W_, Cc,
                                  1 if relu else 0, st), "skimi_conv3x3_n32")
    return out


def split_records(x):
    """fp32 [rows, C] -> bf16 [rows, ceil(C/32), 2, 32]: (hi, lo) halves of every 32-element slice in
    one 128-byte record (the `w_split` operand of `gemm`; a ragged last slice is zero-filled)."""
    _require_cuda(x)
    rows, Cc = x.shape
    out = torch.empty((rows, (Cc + 31) // 32, 2, 32), dtype=torch.bfloat16, device=x.device)
    check(lib().skimi_split_records(ptr(x), x.stride(0), rows, Cc, ptr(out), _lib.current_stream()), "skimi_split_records")
    return out


def records_buffer(rows, Cc, device="cuda"):
    """bf16 buffer for [rows, Cc] as bf16x3 records + the 256-byte zero page behind them."""
    return torch.empty(rows * ((Cc + 31) // 32) * 64 + 128, dtype=torch.bfloat16, device=device)


def x3_scratch_numel(rows, Cc):
    """fp32 elements of the `x3_scratch` that `gemm(..., w_split=...)` needs for an A buffer of [rows, Cc]."""
    return rows * ((Cc + 31) // 32 * 32) + 64


def split_planes(x):
    """fp32 [rows, C] -> bf16 [2, rows, C] (hi, lo) with hi + lo ~= x to ~2^-17 relative."""
    _require_cuda(x)
    rows, Cc = x.shape
    out = torch.empty((2, rows, Cc), dtype=torch.bfloat16, device=x.device)
    check(lib().skimi_split_planes(ptr(x), x.stride(0), rows, Cc, ptr(out[0]), ptr(out[1]), _lib.current_stream()),
          "skimi_split_planes")
    return out


def quant_mx(x: torch.Tensor):
    """x [rows, K] (f32 | bf16, device) -> (payload uint8 [rows, Kp], scales uint8 [rows, Kp / 32]): the MXFP8
    operand form of `gemm_fp8` (e4m3 elements, one E8M0 scale per 32 elements, Kp = K rounded up to 128)."""
    _require_cuda(x)
    x = x.contiguous()
    rows, K = x.shape
    Kp = (K + 127) // 128 * 128
    q = torch.empty((rows, Kp), dtype=torch.uint8, device=x.device)
    s = torch.empty((rows, Kp // 32), dtype=torch.uint8, device=x.device)
    check(lib().skimi_quant_mx(ptr(x), _dt(x), x.stride(0), rows, K, ptr(q), ptr(s), _lib.current_stream()), "skimi_quant_mx")
    return q, s


def layernorm_mx(x, gamma, beta, eps=1e-5):
    """LayerNorm of fp32 rows [rows, C] written directly as an MXFP8 operand (payload, scales) -- `quant_mx(layernorm(x))`
    in one pass (C % 256 == 0)."""
    _require_cuda(x, gamma, beta)
    rows, Cc = x.shape
    q = torch.empty((rows, Cc), dtype=torch.uint8, device=x.device)
    s = torch.empty((rows, Cc // 32), dtype=torch.uint8, device=x.device)
    check(lib().skimi_layernorm_mx(ptr(x), x.stride(0), rows, Cc, ptr(gamma), ptr(beta), eps, ptr(q), ptr(s),
                                   _lib.current_stream()), "skimi_layernorm_mx")
    return q, s


def gemm_fp8(a_q, a_s, w_q, w_s, K, *, bias=None, act=ACT_NONE, gamma=None, resid=None, out=None, out_dtype=torch.float32,
             out_mx=False):
    """out[m][n] = epilogue(sum_k A[m][k] W[n][k]) on the MXFP8 MFMA; operands from `quant_mx`.
    out_mx: return (payload uint8 [M, N], scales uint8 [M, N / 32]) -- the result directly as the next gemm_fp8's operand."""
    _require_cuda(a_q, a_s, w_q, w_s, bias, gamma, resid, out)
    M, N = a_q.shape[0], w_q.shape[0]
    Kp = (K + 127) // 128 * 128
    for t, cols, what in ((a_q, Kp, "a_q"), (w_q, Kp, "w_q"), (a_s, Kp // 32, "a_s"), (w_s, Kp // 32, "w_s")):
        if t.dtype != torch.uint8 or t.dim() != 2 or t.shape[1] != cols or not t.is_contiguous():
            raise _lib.SkimiError(f"gemm_fp8: {what} must be a contiguous uint8 [rows, {cols}] array for K = {K} "
                                  f"(got {tuple(t.shape)} {t.dtype}); operands come from quant_mx")
    if a_s.shape[0] != M or w_s.shape[0] != N:
        raise _lib.SkimiError("gemm_fp8: scale rows do not match the payload rows")
    if out_mx:
        q = torch.empty((M, N), dtype=torch.uint8, device=a_q.device)
        sc = torch.empty((M, N // 32), dtype=torch.uint8, device=a_q.device)
        check(lib().skimi_gemm_fp8(ptr(a_q), ptr(a_s), ptr(w_q), ptr(w_s), M, N, K, ptr(bias), act, None, None, 0, ptr(q),
                                   _lib.FP8MX, N, ptr(sc), _lib.current_stream()), "skimi_gemm_fp8")
        return q, sc
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a_q.device)
    check(lib().skimi_gemm_fp8(ptr(a_q), ptr(a_s), ptr(w_q), ptr(w_s), M, N, K, ptr(bias), act, ptr(gamma), ptr(resid),
                               resid.stride(0) if resid is not None else 0, ptr(out), _dt(out), out.stride(0), None,
                               _lib.current_stream()), "skimi_gemm_fp8")
    return out
