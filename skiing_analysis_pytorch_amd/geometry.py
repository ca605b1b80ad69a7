"""Device-side geometry post-processing (C-ABI: skimi_pose_to_cameras, skimi_unproject_depth,
skimi_triangulate_dlt) plus the small host helpers of the reference's VGGT wrapper.

Reference: vggt/vggt/utils/pose_enc.py:62-124, rotation.py:14-44, geometry.py:15-117,
vggt/triangulate.py:13-71, vggt/vggt/infer.py:107-155.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr


def pose_encoding_to_extri_intri(pose_encoding: torch.Tensor, image_size_hw, build_intrinsics=True):
    """[B, S, 9] -> (extrinsics [B, S, 3, 4], intrinsics [B, S, 3, 3] | None); device tensors."""
    if not pose_encoding.is_cuda:
        raise _lib.SkimiError("pose_encoding_to_extri_intri needs a device tensor")
    pe = pose_encoding.contiguous().to(torch.float32)
    lead = pe.shape[:-1]
    rows = pe.numel() // 9
    H, W = image_size_hw
    E = torch.empty((*lead, 3, 4), dtype=torch.float32, device=pe.device)
    K = torch.empty((*lead, 3, 3), dtype=torch.float32, device=pe.device) if build_intrinsics else None
    check(lib().skimi_pose_to_cameras(ptr(pe), rows, int(H), int(W), ptr(E), ptr(K), _lib.current_stream()),
          "skimi_pose_to_cameras")
    return E, K


def unproject_depth_map_to_point_map(depth: torch.Tensor, extrinsic: torch.Tensor, intrinsic: torch.Tensor):
    """depth [S, H, W, 1] | [S, H, W], E [S, 3, 4], K [S, 3, 3] -> world points [S, H, W, 3] (device)."""
    if depth.dim() == 4:
        depth = depth[..., 0]
    d = depth.contiguous().to(torch.float32)
    S, H, W = d.shape
    out = torch.empty((S, H, W, 3), dtype=torch.float32, device=d.device)
    E = extrinsic.contiguous().to(torch.float32)
    K = intrinsic.contiguous().to(torch.float32)
    check(lib().skimi_unproject_depth(ptr(d), ptr(E), ptr(K), ptr(out), S, H, W, _lib.current_stream()),
          "skimi_unproject_depth")
    return out


def triangulate_joints(K: torch.Tensor, R: torch.Tensor, t: torch.Tensor, keypoints: torch.Tensor):
    """K, R [T, V, 3, 3], t [T, V, 3], keypoints [T, V, J, 2] (pixels) -> [T, J, 3].
    V = 2 is the reference's triangulate_one_frame; V > 2 stacks two DLT rows per view."""
    for a in (K, R, t, keypoints):
        if not a.is_cuda:
            raise _lib.SkimiError("triangulate_joints needs device tensors")
    T, V, J, _ = keypoints.shape
    K, R, t = (a.contiguous().to(torch.float32) for a in (K, R, t))
    kp = keypoints.contiguous().to(torch.float32)
    out = torch.empty((T, J, 3), dtype=torch.float32, device=kp.device)
    check(lib().skimi_triangulate_dlt(ptr(K), ptr(R), ptr(t), ptr(kp), ptr(out), T, V, J, _lib.current_stream()),
          "skimi_triangulate_dlt")
    return out


# ---- host helpers of the wrapper (small arrays, NumPy as in the reference) -----------------
def extrinsic_to_RT(extrinsic):
    """vggt/vggt/infer.py:107-126: E (T,3,4)|(T,4,4)|(3,4)|(4,4) -> R (T,3,3), t (T,3), C = -R^T t"""
    E = np.asarray(extrinsic)
    if E.ndim == 2:
        E = E[None, ...]
    if E.shape[-2:] == (4, 4):
        E = E[:, :3, :]
    R = E[:, :3, :3]
    t = E[:, :3, 3]
    C = -np.einsum("tij,tj->ti", R.transpose(0, 2, 1), t)
    return R, t, C


def scale_intrinsics(K: np.ndarray, orig_size, new_size) -> np.ndarray:
    """vggt/vggt/infer.py:128-155: rescale K from `orig_size` (H, W) pixels to `new_size`."""
    H0, W0 = orig_size
    H1, W1 = new_size
    sx, sy = W1 / W0, H1 / H0
    K_new = K.copy().astype(float)
    K_new[0, 0] *= sx
    K_new[1, 1] *= sy
    K_new[0, 2] *= sx
    K_new[1, 2] *= sy
    return K_new
