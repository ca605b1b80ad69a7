"""Counterpart of vggt/multi_view_process.py with the reference's signature:

    process_multi_view_video(left_video_path, left_pt_path, right_video_path, right_pt_path,
                             out_root, inference_output_path, cfg) -> Optional[Path]
                                                              (multi_view_process.py:68-76)

What it reproduces per time step (multi_view_process.py:133-309): left/right frame -> VGGT
(`CameraHead.reconstruct_from_frames`) -> person-centred world origin from the two point maps inside the
detector boxes (`extract_person_points`, :356-395) -> the right camera's 180-degree alignment (:204-217)
-> DLT triangulation of the 17 joints (`triangulate_one_frame`, vggt/triangulate.py:38-71), and at the
end the camera / joints NPZ of `save_camera_info` (vggt/save.py:84-110, called at :312-319).

What it leaves out (out of scope): video decode (frames come from the `.pt` files, which
`prepare_dataset` can embed; the video paths only name the subject), PNG / GLB / matplotlib output, the
Open3D ICP refinement (:285-296; Open3D is not part of this build, "parity unpinned") and the commented-out
bundle adjustment.  The time steps are independent: they go through the HIP model `steps_per_call` at a
time, sharded over ranks under torch.distributed, and the per-step joints are re-assembled with one
all-gather (parallel.py).
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch

from . import formats, fuse, geometry, parallel
from .infer import CameraHead, cfg_get, save_camera_info

logger = logging.getLogger(__name__)

_R_ALIGN = np.array([[-1, 0, 0], [0, 1, 0], [0, 0, -1]], dtype=np.float64)   # multi_view_process.py:204-210


def extract_person_points(pointmap: np.ndarray, bbox, img_size) -> np.ndarray:
    """multi_view_process.py:356-395: the point-map pixels inside the detector box (box given in source
    pixels), finite, within 3 sigma of the median depth -> [N, 3]."""
    H_img, W_img = img_size
    H_pm, W_pm = pointmap.shape[:2]
    sx, sy = W_pm / W_img, H_pm / H_img
    x1, y1, x2, y2 = bbox
    x1, x2, y1, y2 = int(x1 * sx), int(x2 * sx), int(y1 * sy), int(y2 * sy)
    x1, x2 = np.clip(x1, 0, W_pm - 1), np.clip(x2, 0, W_pm)
    y1, y2 = np.clip(y1, 0, H_pm - 1), np.clip(y2, 0, H_pm)
    P = pointmap[y1:y2, x1:x2, :].reshape(-1, 3)
    P = P[np.isfinite(P).all(axis=1)]
    if len(P) > 0:
        z = P[:, 2]
        P = P[np.abs(z - np.median(z)) < 3.0 * np.std(z)]
    return P


def scale_bbox(bbox, source_size, target_size) -> List[float]:
    """multi_view_process.py:398-424"""
    (src_h, src_w), (tgt_h, tgt_w) = source_size, target_size
    sx, sy = tgt_w / src_w, tgt_h / src_h
    x1, y1, x2, y2 = bbox
    return [x1 * sx, y1 * sy, x2 * sx, y2 * sy]


def recenter_and_align(R: np.ndarray, t: np.ndarray, origin: np.ndarray):
    """multi_view_process.py:196-217 on copies: move the world origin onto the person
    (t_c += R_c @ origin), then turn the right camera by 180 degrees about y and mirror its x / z
    translation."""
    R, t = np.array(R, dtype=np.float64), np.array(t, dtype=np.float64)
    for cam in range(len(R)):
        t[cam] = t[cam] + R[cam] @ origin
    R[1] = _R_ALIGN @ R[1]
    t[1] = _R_ALIGN @ t[1]
    t[1][0] = -t[1][0]
    t[1][2] = -t[1][2]
    return R, t


def _bbox_of(bboxes: np.ndarray, idx: int):
    b = bboxes[idx]
    return b if b.ndim == 1 else b[0]


def process_multi_view_video(left_video_path: Path, left_pt_path: Path, right_video_path: Path, right_pt_path: Path,
                             out_root: Path, inference_output_path: Path, cfg, camera_head: Optional[CameraHead] = None,
                             steps_per_call: int = 4) -> Optional[Path]:
    """Same arguments as the reference; `camera_head` (optional) supplies an already loaded model instead
    of `CameraHead(cfg, out_dir / "vggt_infer")` (offline there is no checkpoint URL: cfg.infer.ckpt_path
    names a local model.pt otherwise).  Returns the output directory."""
    left_video_path, right_video_path = Path(left_video_path), Path(right_video_path)
    out_root, inference_output_path = Path(out_root), Path(inference_output_path)
    subject = left_video_path.parent.name or "default"
    out_dir = out_root / "multi_view" / subject
    out_dir.mkdir(parents=True, exist_ok=True)
    inference_output_path.mkdir(parents=True, exist_ok=True)
    logger.info(f"[Run-MV] {left_video_path} & {right_video_path} -> {out_dir} | ")

    lk, _ls, lb, _lbs, lf = formats.load_info(left_pt_path, video_file_path=left_video_path, assume_normalized=False)
    rk, _rs, rb, _rbs, rf = formats.load_info(right_pt_path, video_file_path=right_video_path, assume_normalized=False)
    for frames_, pt_, vid_ in ((lf, left_pt_path, left_video_path), (rf, right_pt_path, right_video_path)):
        if frames_ is None:
            raise RuntimeError(f"{pt_} embeds no frames and {vid_} cannot be decoded here (formats.read_video_frames)")
    if cfg_get(cfg, "infer.hflip", False):       # multi_view_process.py:116-127
        W0 = lf[0].shape[1]
        rf = torch.flip(rf, [2])
        rk = rk.copy()
        rk[..., 0] = W0 - rk[..., 0]
        rb = rb.copy()
        x1, x2 = rb[..., 0].copy(), rb[..., 2].copy()
        rb[..., 0], rb[..., 2] = W0 - x2, W0 - x1

    head = camera_head if camera_head is not None else CameraHead(cfg, out_dir / "vggt_infer")
    if head.outdir is None:
        head.outdir = out_dir / "vggt_infer"
    T = min(len(lf), len(rf))
    lo, hi, _T_pad = parallel.shard_range(T)
    source_size = tuple(lf.shape[1:3])
    x3d_l, K_l, R_l, t_l, C_l = [], [], [], [], []
    for a in range(lo, hi, steps_per_call):
        idx = [min(i, T - 1) for i in range(a, min(a + steps_per_call, hi))]     # padded steps repeat the last one
        # a padded step (index >= T on the last ranks) is computed to keep the collective uniform, but does not
        # write frame_{T-1}/predictions.npz a second time
        write = [i < T for i in range(a, min(a + steps_per_call, hi))]
        recs = head.reconstruct_batch(idx, [[lf[i], rf[i]] for i in idx], write=write)
        Ks, Rs, ts = [], [], []
        for i, (_E, K_res, R, t, C, wp) in zip(idx, recs):
            pl = extract_person_points(wp[0], _bbox_of(lb, i), source_size)
            pr = extract_person_points(wp[1], _bbox_of(rb, i), source_size)
            origin = 0.5 * (pl.mean(axis=0) + pr.mean(axis=0)) if len(pl) and len(pr) else np.zeros(3)
            R2, t2 = recenter_and_align(R, t, origin)
            Ks.append(np.stack(K_res[:2]))
            Rs.append(R2)
            ts.append(t2)
            C_l.append(np.asarray(C))
        Kd = torch.from_numpy(np.stack(Ks)).to(head.device, torch.float32)
        Rd = torch.from_numpy(np.stack(Rs)).to(head.device, torch.float32)
        td = torch.from_numpy(np.stack(ts)).to(head.device, torch.float32)
        kp = torch.from_numpy(np.stack([np.stack([lk[i], rk[i]]) for i in idx])).to(head.device, torch.float32)
        x3d_l.append(geometry.triangulate_joints(Kd, Rd, td, kp))          # [n, 17, 3] on the device
        K_l += Ks
        R_l += Rs
        t_l += ts
    # the path's ONE collective: joints + K + R + t + C of this rank's steps as one packed record per step
    # (no-op on one rank)
    dev = head.device
    todev = lambda lst: torch.from_numpy(np.stack(lst)).to(dev)   # noqa: E731
    x3d, Ka, Ra, ta, Ca = (a.cpu().numpy() for a in parallel.all_gather_packed(
        [torch.cat(x3d_l), todev(K_l), todev(R_l), todev(t_l), todev(C_l)], T))
    # fuse/'s temporal smoothing of the gathered joints (BASELINE config 4; fuse/fuse.py:329-412)
    x3d_smoothed = fuse.temporal_smooth_ema(x3d.astype(np.float64)) if cfg_get(cfg, "infer.smooth", True) else None
    if parallel.world()[0] == 0:
        # icp_refined = False: the reference stores R, t and the joints AFTER its Open3D ICP update
        # (multi_view_process.py:285-319); this build has no ICP (out of scope, "parity unpinned"), the arrays are
        # the pre-ICP quantities under the same keys
        logger.warning("[Run-MV] cameras / joints are written without the reference's Open3D ICP refinement (icp_refined=False)")
        save_camera_info(out_pt_path=inference_output_path / f"{subject}_multi_view_3d_info.npz",
                         all_frame_x3d=list(x3d), all_frame_camera_intrinsics=list(Ka), all_frame_R=list(Ra),
                         all_frame_t=list(ta), all_frame_C=list(Ca),
                         extra={"icp_refined": np.array(False)} | ({"x3d_smoothed": x3d_smoothed} if x3d_smoothed is not None else {}))
    return out_dir
