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


# ---------------------------------------------------------------------------------------------------
# The ring rig: a well-conditioned standard scene for "what does a pose_enc error do to the joints".
#
# The synthetic model's own cameras are a degenerate scene (random weights: FoV above pi, seven of eight
# views in a 0.05-unit cluster), on which the same pose_enc error moves the joints by anything between
# 0.2 and 13 units.  The ring rig applies the error the mode under test makes on pose_enc,
# d = pose_enc_test - pose_enc_oracle, to S cameras on a ring (radius 3, looking at the origin, FoV 55
# degrees, unit quaternions, adjacent baseline 2.3): the reference cameras are the ring itself, the test
# cameras are ring + d, and the metric is the same MPJPE of the same S-view DLT on identical keypoints.
# ---------------------------------------------------------------------------------------------------
def _mat_to_quat_xyzw(R: np.ndarray) -> np.ndarray:
    """rotation matrix -> unit quaternion, scalar last (the layout rotation.py:14-44 consumes)"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def ring_rig_pose_enc(S: int, radius: float = 3.0, height: float = 0.6, fov_deg: float = 55.0) -> np.ndarray:
    """[S, 9] pose encodings (T, quat XYZW, fov_h, fov_w: pose_enc.py:62-124) of S cameras on a ring around the
    origin, OpenCV convention (x right, y down, z forward), cam-from-world extrinsics."""
    pe = np.zeros((S, 9), dtype=np.float64)
    for v in range(S):
        a = 2 * np.pi * v / S
        C = np.array([radius * np.cos(a), -height, radius * np.sin(a)])
        z = -C / np.linalg.norm(C)                              # looks at the origin
        x = np.cross(np.array([0.0, 1.0, 0.0]), z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z], axis=0)                         # rows = camera axes in the world
        pe[v, :3] = -R @ C
        pe[v, 3:7] = _mat_to_quat_xyzw(R)
        pe[v, 7:] = np.deg2rad(fov_deg)
    return pe


def ring_rig_scene(pose_enc_ref: torch.Tensor, image_hw, joints: int = 17, seed: int = 0):
    """pose_enc_ref [T, S, 9] (only its shape is used) -> (ring [T, S, 9] float32 pose encodings, kps [T, S, J, 2]
    pixels of J points of a 1.7-unit 'skier' around the origin seen by the ring cameras, joints_ref [T, J, 3] =
    the reference DLT with the ring cameras)."""
    T, S, _ = pose_enc_ref.shape
    ring = torch.from_numpy(np.broadcast_to(ring_rig_pose_enc(S), (T, S, 9)).copy()).to(torch.float32)
    E, K = vggt_oracle.pose_encoding_to_extri_intri(ring, image_hw)
    E64, K64 = E.numpy().astype(np.float64), K.numpy().astype(np.float64)
    rng = np.random.default_rng(seed)
    kps = np.zeros((T, S, joints, 2), dtype=np.float32)
    ref = np.zeros((T, joints, 3), dtype=np.float32)
    for t in range(T):
        X = rng.normal(size=(joints, 3)) * np.array([0.25, 0.45, 0.25])
        R, tr = E64[t, :, :3, :3], E64[t, :, :3, 3]
        cam = np.einsum("vab,jb->vja", R, X) + tr[:, None, :]
        pix = np.einsum("vab,vjb->vja", K64[t], cam)
        kps[t] = (pix[..., :2] / pix[..., 2:]).astype(np.float32)
        ref[t] = vggt_oracle.triangulate_one_frame(K64[t], R, tr, kps[t].astype(np.float64))
    return ring, torch.from_numpy(kps), ref


def ring_rig_test_pose_enc(ring: torch.Tensor, pose_enc_test: torch.Tensor, pose_enc_ref: torch.Tensor) -> torch.Tensor:
    """the cameras of the mode under test on the rig: ring + (pose_enc_test - pose_enc_oracle)"""
    d = pose_enc_test.detach().to(torch.float32).cpu() - pose_enc_ref.detach().to(torch.float32).cpu()
    return ring + d
