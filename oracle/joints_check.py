"""TEST INFRASTRUCTURE (checker only: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

The error the metric is defined on (BASELINE.json: "MPJPE vs ref"): mean per-joint
position error between the build's 3D joints and the reference CPU path's 3D joints on identical
2D inputs.

  mpjpe                  VideoPose3D/common/loss.py:11-17
  reference joints       pose_enc (fp32 CPU oracle) -> pose_encoding_to_extri_intri
                         (vggt/vggt/utils/pose_enc.py:62-124) -> DLT over the S views
                         (vggt/triangulate.py:13-71)

The 2D keypoints are projections of known 3D points through the ORACLE's cameras (a noise-free,
consistent observation set: the reference DLT recovers the points, so the system is well conditioned
by construction and `conditioning_error` says how well), not uniform random pixels -- random pixels
give DLT systems with no consistent solution, whose smallest singular vector is arbitrarily
sensitive to the cameras.
"""
from __future__ import annotations

import numpy as np
import torch

from . import vggt_oracle


def mpjpe(predicted, target) -> float:
    """loss.py:11-17: mean over all joints of the Euclidean distance."""
    predicted = torch.as_tensor(np.asarray(predicted), dtype=torch.float64)
    target = torch.as_tensor(np.asarray(target), dtype=torch.float64)
    assert predicted.shape == target.shape
    return float(torch.mean(torch.norm(predicted - target, dim=len(target.shape) - 1)))


def keypoints_from_oracle_cameras(pose_enc_ref: torch.Tensor, image_hw, joints: int = 17, seed: int = 0):
    """pose_enc_ref [T, S, 9] (fp32 CPU oracle) -> (kps [T, S, J, 2] pixels, X [T, J, 3] world points,
    joints_ref [T, J, 3] = reference DLT on those keypoints with the oracle's cameras).

    The J points of a time step sit around a centre three baselines in front of view 1 (views 1..S-1 of
    the synthetic model are a tight cluster, view 0 -- own camera token, aggregator.py:308-331 -- is
    the far one), spread by a third of the view-0 / view-1 baseline."""
    pe = pose_enc_ref.detach().to(torch.float32).cpu()
    T, S, _ = pe.shape
    E, K = vggt_oracle.pose_encoding_to_extri_intri(pe, image_hw)
    E64, K64 = E.numpy().astype(np.float64), K.numpy().astype(np.float64)
    rng = np.random.default_rng(seed)
    kps = np.zeros((T, S, joints, 2), dtype=np.float32)
    Xw = np.zeros((T, joints, 3), dtype=np.float64)
    ref = np.zeros((T, joints, 3), dtype=np.float32)
    for t in range(T):
        R, tr = E64[t, :, :3, :3], E64[t, :, :3, 3]
        C = -np.einsum("vji,vj->vi", R, tr)                 # camera centres, C = -R^T t
        base = max(np.linalg.norm(C[0] - C[min(1, S - 1)]), 1e-2)
        view = R[min(1, S - 1)].T @ np.array([0.0, 0.0, 1.0])   # viewing direction of view 1 in the world
        centre = 0.5 * (C[0] + C[min(1, S - 1)]) + 3.0 * base * view
        X = centre + rng.normal(size=(joints, 3)) * (base / 3.0)
        cam = np.einsum("vab,jb->vja", R, X) + tr[:, None, :]
        pix = np.einsum("vab,vjb->vja", K64[t], cam)
        kps[t] = (pix[..., :2] / pix[..., 2:]).astype(np.float32)
        Xw[t] = X
        ref[t] = vggt_oracle.triangulate_one_frame(K64[t], R, tr, kps[t].astype(np.float64))
    return torch.from_numpy(kps), Xw, ref


def conditioning_error(Xw, joints_ref) -> float:
    """how far the reference DLT lands from the points the keypoints were projected from"""
    return mpjpe(joints_ref, Xw)
