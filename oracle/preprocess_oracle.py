"""TEST INFRASTRUCTURE ONLY — CPU restatement of the image resize of vggt/load.py:104
(`img.resize((new_width, new_height), Image.Resampling.BICUBIC)`).

The algorithm lives in Pillow (a dependency of the reference, not vendored in /root/reference;
the container has Pillow 12.2.0), file src/libImaging/Resample.c: `bicubic_filter`,
`precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`,
`ImagingResampleVertical_8bpc`, `ImagingResample` (horizontal pass first, uint8 intermediate).
Pinned: tests/test_oracle_golden.py compares this restatement bit for bit with PIL itself.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds, ksize


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    kk, bounds, _ = precompute_coeffs(img.shape[0], out_size)
    out = np.zeros((out_size,) + img.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        ss = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            ss += img[xmin + x].astype(np.int64) * int(kk[xx, x])
        out[xx] = np.clip(ss >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bicubic(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """uint8 [H, W, C] -> uint8 [new_h, new_w, C], as Image.resize((new_w, new_h), BICUBIC)."""
    out = img
    if new_w != img.shape[1]:
        out = _resample_axis0(out.transpose(1, 0, 2), new_w).transpose(1, 0, 2)
    if new_h != img.shape[0]:
        out = _resample_axis0(out, new_h)
    return np.ascontiguousarray(out)
