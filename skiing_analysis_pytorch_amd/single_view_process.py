"""Counterpart of vggt/single_view_process.py with the reference's signature:

    process_single_view_video(video_path, pt_path, out_root, inference_output_path, cfg) -> Optional[Path]
                                                              (single_view_process.py:90-96)

Every 30th frame of ONE camera forms a single S = ceil(T / 30) view stack (single_view_process.py:130),
which goes through `CameraHead.reconstruct_from_frames` once (:152-163); the cameras of those frames are
written with `save_camera_info` (:171-178, vggt/save.py:84-110).  Left out: the video decode
(frames come from the `.pt` file when embedded), the per-frame skeleton PNGs, GLB export.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Optional

from . import formats
from .infer import CameraHead, save_camera_info

logger = logging.getLogger(__name__)

FRAME_STRIDE = 30    # single_view_process.py:130


def process_single_view_video(video_path: Path, pt_path: Path, out_root: Path, inference_output_path: Path, cfg,
                              camera_head: Optional[CameraHead] = None) -> Optional[Path]:
    """Same arguments as the reference; `camera_head` (optional) supplies an already loaded model instead of
    `CameraHead(cfg, out_dir / "vggt_infer")`.  Returns the output directory."""
    video_path, out_root, inference_output_path = Path(video_path), Path(out_root), Path(inference_output_path)
    subject = video_path.parent.name or "default"
    out_dir = out_root / "single_view" / subject
    out_dir.mkdir(parents=True, exist_ok=True)
    inference_output_path.mkdir(parents=True, exist_ok=True)
    logger.info(f"[Run-SV] {video_path} -> {out_dir} | ")
    _kpts, _scores, _bboxes, _bscores, frames = formats.load_info(pt_path, video_file_path=video_path, assume_normalized=False)
    if frames is None:
        raise RuntimeError(f"{pt_path} embeds no frames and {video_path} cannot be decoded here (formats.read_video_frames)")
    head = camera_head if camera_head is not None else CameraHead(cfg, out_dir / "vggt_infer")
    if head.outdir is None:
        head.outdir = out_dir / "vggt_infer"
    inference_imgs = [frames[idx] for idx in range(0, len(frames), FRAME_STRIDE)]
    _E, K_resized, R, t, C, _wp = head.reconstruct_from_frames(imgs=inference_imgs, frame_id=0)
    # the reference keeps the "multi_view" file name here too (single_view_process.py:172)
    save_camera_info(out_pt_path=inference_output_path / f"{subject}_multi_view_3d_info.npz",
                     all_frame_camera_intrinsics=[K_resized], all_frame_R=[R], all_frame_t=[t], all_frame_C=[C])
    return out_dir
