"""TEST INFRASTRUCTURE (checker only).  OCP microscaling FP8 (MXFP8) in NumPy: the number format of
`skimi_quant_mx` / `skimi_gemm_fp8` (csrc/gemm_fp8.hip; BASELINE config 5).

The reference has no fp8 path at all (it runs bf16 autocast, vggt/vggt/infer.py:78-84): this file restates the
published formats, not reference code --
  * element format e4m3 "fn" of the OCP 8-bit floating point specification (OFP8, rev 1.0): 1 sign, 4 exponent
    (bias 7), 3 mantissa bits, no infinities, S.1111.111 = NaN, largest finite 448, subnormals m/8 * 2^-6;
    conversion from fp32 rounds to nearest, ties to even;
  * block scaling of the OCP Microscaling Formats (MX) specification v1.0: one E8M0 scale (value 2^(byte - 127))
    per 32 consecutive elements.  The scale rule is this build's: the smallest power of two with
    amax / scale <= 448 (no element clips), byte 0 for an all-zero block.
Known-answer vectors for the element format are checked in tests/test_host_cpu.py.
"""
from __future__ import annotations

import numpy as np


def e4m3_decode_table() -> np.ndarray:
    """value of every byte code (NaN for 0x7F / 0xFF)"""
    codes = np.arange(256)
    s = np.where(codes & 0x80, -1.0, 1.0)
    e = (codes >> 3) & 0xF
    m = codes & 0x7
    val = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))
    val = s * val
    val[(codes & 0x7F) == 0x7F] = np.nan
    return val


_TABLE = e4m3_decode_table()
_POS = _TABLE[:127]          # codes 0 .. 126: 0 .. 448, increasing


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """fp32/fp64 array -> uint8 codes: round to nearest, ties to even, saturating at +-448"""
    x = np.asarray(x, dtype=np.float64)
    a = np.minimum(np.abs(x), 448.0)
    hi = np.searchsorted(_POS, a, side="left").clip(0, 126)      # first code with value >= a
    lo = np.maximum(hi - 1, 0)
    d_hi, d_lo = _POS[hi] - a, a - _POS[lo]
    pick_hi = (d_hi < d_lo) | ((d_hi == d_lo) & ((hi & 1) == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    return np.where(np.signbit(x), code | 0x80, code).astype(np.uint8)


def e4m3_decode(codes: np.ndarray) -> np.ndarray:
    return _TABLE[np.asarray(codes, dtype=np.uint8)]


def mx_quantize(x: np.ndarray):
    """x [rows, K] -> (payload uint8 [rows, Kp], scales uint8 [rows, Kp / 32]); Kp = K rounded up to 128"""
    x = np.asarray(x, dtype=np.float32)
    rows, K = x.shape
    Kp = (K + 127) // 128 * 128
    xp = np.zeros((rows, Kp), dtype=np.float32)
    xp[:, :K] = x
    blk = xp.reshape(rows, Kp // 32, 32)
    amax = np.abs(blk).max(axis=2)
    t = (amax.astype(np.float32) * np.float32(1.0 / 448.0)).astype(np.float32)       # as the kernel: fp32 multiply
    bits = t.view(np.uint32)
    e = ((bits >> 23) & 0xFF).astype(np.int32) - 127 + ((bits & 0x7FFFFF) != 0)
    sb = np.where(amax > 0, np.clip(e + 127, 1, 254), 0).astype(np.uint8)
    inv = np.where(sb > 0, np.ldexp(1.0, 127 - sb.astype(np.int32)), 0.0).astype(np.float32)
    q = e4m3_encode((blk * inv[..., None]).astype(np.float32))
    return q.reshape(rows, Kp), sb


def mx_dequantize(payload: np.ndarray, scales: np.ndarray) -> np.ndarray:
    rows, Kp = payload.shape
    v = e4m3_decode(payload).reshape(rows, Kp // 32, 32)
    return (v * np.ldexp(1.0, scales.astype(np.int32) - 127)[..., None]).reshape(rows, Kp)


def gemm_mx(a: np.ndarray, w: np.ndarray) -> np.ndarray:
    """out[m][n] = sum_k dq(A)[m][k] dq(W)[n][k] in float64 on the quantised operands"""
    aq, asx = mx_quantize(a)
    wq, wsx = mx_quantize(w)
    return mx_dequantize(aq, asx) @ mx_dequantize(wq, wsx).T
