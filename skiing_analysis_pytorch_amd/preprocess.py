"""Device-side `load_and_preprocess_images` (vggt/load.py:38-183).

The reference resizes each frame on the host with `PIL.Image.resize((w, h), Image.Resampling.BICUBIC)`
and converts with `TF.ToTensor()`.  Here the uint8 frame is uploaded once and resized on the GPU by
the two integer passes of Pillow's own 8-bit resampler (src/libImaging/Resample.c; Pillow is a
dependency of the reference, not vendored in it: the algorithm below restates
`precompute_coeffs` / `normalize_coeffs_8bpc`, pinned against the installed Pillow by the tests), so
the result is bit-identical to the host path of `infer.load_and_preprocess_images`.
Only the coefficient tables (a few KB per image size, cached) are computed on the host.
"""
from __future__ import annotations

import functools
import math
from typing import Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr

PRECISION_BITS = 32 - 8 - 2   # Resample.c: 8-bit pixels, 2 bits of headroom for the overshoot of the cubic


def _bicubic(x: np.ndarray) -> np.ndarray:
    """Resample.c bicubic_filter (a = -0.5), support 2."""
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1,
                    np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


@functools.lru_cache(maxsize=64)
def bicubic_coeffs(in_size: int, out_size: int):
    """-> (kk int32 [out, ksize], bounds int32 [out, 2], ksize): Resample.c precompute_coeffs +
    normalize_coeffs_8bpc for the whole axis (box = full image)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _bicubic((np.arange(xmax) + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:            # C accumulates left to right in double
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
        kk[xx, :xmax] = np.trunc(q).astype(np.int32)   # (int) cast truncates toward zero
        bounds[xx] = (xmin, xmax)
    return kk, bounds, ksize


def _tables(in_size: int, out_size: int, device):
    kk, bounds, ksize = bicubic_coeffs(in_size, out_size)
    return torch.from_numpy(kk).to(device), torch.from_numpy(bounds).to(device), ksize


def resize_bicubic_u8(img: torch.Tensor, new_w: int, new_h: int) -> torch.Tensor:
    """uint8 [H, W, 3] on the GPU -> uint8 [new_h, new_w, 3]: PIL's Image.resize(BICUBIC), bit for bit."""
    if not img.is_cuda or img.dtype != torch.uint8 or img.dim() != 3:
        raise _lib.SkimiError("resize_bicubic_u8 needs a uint8 [H, W, C] device tensor")
    img = img.contiguous()
    H, W, Cc = img.shape
    st = _lib.current_stream()
    cur = img
    if new_w != W:   # Pillow: horizontal pass first, uint8 intermediate
        kk, bounds, ksize = _tables(W, new_w, img.device)
        out = torch.empty((H, new_w, Cc), dtype=torch.uint8, device=img.device)
        check(lib().skimi_resample_u8(ptr(cur), ptr(out), H, W, new_w, Cc, ptr(kk), ptr(bounds), ksize, st), "skimi_resample_u8")
        cur = out
    if new_h != H:
        kk, bounds, ksize = _tables(H, new_h, img.device)
        out = torch.empty((new_h, cur.shape[1], Cc), dtype=torch.uint8, device=img.device)
        check(lib().skimi_resample_u8(ptr(cur), ptr(out), 1, H, new_h, cur.shape[1] * Cc, ptr(kk), ptr(bounds), ksize, st),
              "skimi_resample_u8")
        cur = out
    return cur


def target_size(width: int, height: int, mode: str, target: int = 518):
    """load.py:86-101: the size the frame is resized to."""
    if mode == "pad":
        if width >= height:
            return target, round(height * (target / width) / 14) * 14
        return round(width * (target / height) / 14) * 14, target
    return target, round(height * (target / width) / 14) * 14


def load_and_preprocess_images_device(image_list: Sequence, mode: str = "crop", device="cuda") -> torch.Tensor:
    """Same contract as `infer.load_and_preprocess_images` (HWC uint8 RGB arrays / tensors in,
    [N, 3, H, W] float32 in [0, 1] out) with the resize, ToTensor, centre crop and white padding on
    the GPU; the result stays in HBM."""
    if len(image_list) == 0:
        raise ValueError("At least 1 image is required")
    if mode not in ["crop", "pad"]:
        raise ValueError("Mode must be either 'crop' or 'pad'")
    target = 518
    dev = torch.device(device)
    resized, shapes = [], []
    for im in image_list:
        arr = im if isinstance(im, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(im)))
        if arr.dtype != torch.uint8 or arr.dim() != 3 or arr.shape[2] != 3:
            raise ValueError("device preprocessing takes HWC uint8 RGB frames (composite RGBA on the host first)")
        arr = arr.to(dev, non_blocking=True)
        height, width = int(arr.shape[0]), int(arr.shape[1])
        new_w, new_h = target_size(width, height, mode, target)
        r = resize_bicubic_u8(arr, new_w, new_h)
        oh, ow = new_h, new_w
        if mode == "crop" and new_h > target:
            oh = target
        if mode == "pad":
            oh = ow = target
        resized.append((r, new_h, new_w, oh, ow))
        shapes.append((oh, ow))
    mh, mw = max(s[0] for s in shapes), max(s[1] for s in shapes)   # load.py:149-170: pad to the largest, white
    out = torch.empty((len(resized), 3, mh, mw), dtype=torch.float32, device=dev)
    st = _lib.current_stream()
    for i, (r, new_h, new_w, oh, ow) in enumerate(resized):
        # offsets of the frame inside its own (oh, ow) box, then of that box inside (mh, mw)
        y_off = (new_h - target) // 2 if (mode == "crop" and new_h > target) else -((oh - new_h) // 2)
        x_off = -((ow - new_w) // 2)
        y_off -= (mh - oh) // 2
        x_off -= (mw - ow) // 2
        check(lib().skimi_u8_hwc_to_f32_chw(ptr(r), new_h, new_w, ptr(out[i]), mh, mw, y_off, x_off, 1.0, st),
              "skimi_u8_hwc_to_f32_chw")
    return out
