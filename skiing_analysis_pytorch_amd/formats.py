"""On-disk formats either side of the hot path.

Input: the per-clip `.pt` written by the reference's prepare_dataset stage
(prepare_dataset/main.py:53-98, process/preprocess.py:160-171):
    {video_name, video_path, frame_count, img_shape, fps,
     detectron2: {bbox [T,4] | [T,N,4], keypoints [T,17,2|3], keypoints_score [T,17]}, yolo: {...}, depth?, frames?}
read as the reference's loaders do (vggt/load.py:268-370 `load_info`,
VideoPose3D/common/custom_dataset.py:106-164), minus the video decode: torchvision / PyAV are
not part of this build, so frames come from the `.pt` itself when present or from the caller.

Output: `.npy [T,17,3]` of the lifter (VideoPose3D/run.py:1089-1092) and the camera NPZ of the
VGGT stage (vggt/save.py:84-110, `infer.save_camera_info`).
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple

import numpy as np
import torch


def _load_pt(path) -> dict:
    # only loaders that execute nothing from the file
    return torch.load(str(path), map_location="cpu", weights_only=True)


def _to_numpy_xy(kpts) -> np.ndarray:
    a = kpts.detach().cpu().numpy() if isinstance(kpts, torch.Tensor) else np.asarray(kpts)
    return a[..., :2].astype(np.float32)


def read_video_frames(video_file_path) -> torch.Tensor:
    """[T, H, W, 3] uint8 RGB frames of a video file, as the reference's
    `read_video(path, pts_unit="sec", output_format="THWC")[0]` (vggt/load.py:292).  Video decode is
    outside this build: an installed torchvision / OpenCV is used when there is one, and the
    error says what to do otherwise."""
    try:
        from torchvision.io import read_video   # the reference's own decoder
        return read_video(str(video_file_path), pts_unit="sec", output_format="THWC")[0]
    except ImportError:
        pass
    try:
        import cv2
    except ImportError:
        raise RuntimeError(f"no video decoder in this environment (torchvision / cv2) for {video_file_path}: embed the "
                           "frames in the .pt file (key 'frames', [T,H,W,3] uint8, as prepare_dataset can) or pass frames=")
    cap, out = cv2.VideoCapture(str(video_file_path)), []
    while True:
        ok, fr = cap.read()
        if not ok:
            break
        out.append(torch.from_numpy(cv2.cvtColor(fr, cv2.COLOR_BGR2RGB)))
    cap.release()
    return torch.stack(out)


def load_info(pt_file_path, frames: Optional[torch.Tensor] = None, assume_normalized: Optional[bool] = None,
              clip_bbox_to_image: bool = True, dtype=np.float32, video_file_path=None
              ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, Optional[torch.Tensor]]:
    """vggt/load.py:268-370.  Returns (keypoints_xy [T,K,2] pixels, keypoints_score [T,K],
    bboxes_xyxy [T,4|N,4] pixels, bbox_scores, frames [T,H,W,3] uint8 | None).  Frames: the `frames`
    argument, else the `.pt` file's own `frames`, else decoded from `video_file_path` when a decoder is
    installed (the reference always decodes the video)."""
    data = _load_pt(pt_file_path)
    if "detectron2" not in data:
        raise KeyError(f"pt file missing 'detectron2' root: {pt_file_path}")
    d2 = data["detectron2"]
    if frames is None and "frames" in data and data["frames"] is not None:
        frames = data["frames"]
    if frames is None and video_file_path is not None and Path(video_file_path).exists():
        frames = read_video_frames(video_file_path)
    if frames is not None:
        H, W = int(frames.shape[1]), int(frames.shape[2])
    elif "img_shape" in data:
        H, W = int(data["img_shape"][0]), int(data["img_shape"][1])
    else:
        raise KeyError("neither frames nor img_shape available to de-normalise the keypoints")
    if "keypoints" not in d2:
        raise KeyError(f"pt file missing detectron2.keypoints: {pt_file_path}")
    kpts_t = d2["keypoints"]
    kpts_xy = _to_numpy_xy(kpts_t)
    mx = np.nanmax(kpts_xy) if kpts_xy.size else 0.0
    if assume_normalized is True or (assume_normalized is None and mx <= 1.5):
        kpts_xy = kpts_xy * np.array([W, H], dtype=np.float32)
    if "keypoints_score" in d2:
        kpt_scores = np.asarray(d2["keypoints_score"].detach().cpu().numpy() if isinstance(d2["keypoints_score"], torch.Tensor) else d2["keypoints_score"])
    elif kpts_t.shape[-1] >= 3:
        kpt_scores = np.asarray(kpts_t[..., 2])
    else:
        kpt_scores = np.ones(kpts_xy.shape[:2], dtype=dtype)
    if kpts_xy.ndim != 3 or kpts_xy.shape[2] != 2:
        raise ValueError(f"Invalid D2 keypoints shape after processing: {kpts_xy.shape}")
    if kpt_scores.shape != kpts_xy.shape[:2]:
        raise ValueError(f"D2 keypoints_score shape {kpt_scores.shape} mismatches keypoints {kpts_xy.shape}")
    if "bbox" not in d2:
        raise KeyError(f"pt file missing detectron2.bbox: {pt_file_path}")
    bb = d2["bbox"]
    bboxes = (bb.detach().cpu().numpy() if isinstance(bb, torch.Tensor) else np.asarray(bb)).astype(dtype, copy=True)
    mxb = np.nanmax(bboxes) if bboxes.size else 0.0
    if assume_normalized is True or (assume_normalized is None and mxb <= 1.5):
        bboxes[..., 0::2] *= float(W)
        bboxes[..., 1::2] *= float(H)
    if "scores" in d2:
        bbox_scores = np.asarray(d2["scores"], dtype=dtype)
    elif "bbox_score" in d2:
        bbox_scores = np.asarray(d2["bbox_score"], dtype=dtype)
    else:
        bbox_scores = np.ones(bboxes.shape[:-1], dtype=dtype)
    if clip_bbox_to_image:
        x1 = np.minimum(bboxes[..., 0], bboxes[..., 2]); x2 = np.maximum(bboxes[..., 0], bboxes[..., 2])
        y1 = np.minimum(bboxes[..., 1], bboxes[..., 3]); y2 = np.maximum(bboxes[..., 1], bboxes[..., 3])
        bboxes = np.stack([np.clip(x1, 0, W - 1), np.clip(y1, 0, H - 1), np.clip(x2, 0, W - 1), np.clip(y2, 0, H - 1)], axis=-1)
    return (kpts_xy.astype(dtype, copy=False), kpt_scores.astype(dtype, copy=False), bboxes.astype(dtype, copy=False),
            bbox_scores.astype(dtype, copy=False), frames)


def save_pose_npy(path, prediction: np.ndarray) -> Path:
    """VideoPose3D/run.py:1089-1092: `<video>.npy` holding [T, 17, 3] float32 camera-space joints."""
    p = Path(path)
    p.parent.mkdir(parents=True, exist_ok=True)
    np.save(p, np.asarray(prediction, dtype=np.float32))
    return p if p.suffix == ".npy" else p.with_suffix(p.suffix + ".npy")


def save_3d_joints(fused_joints_3d: np.ndarray, left_joints_3d: np.ndarray, right_joints_3d: np.ndarray, save_path,
                   fmt: str = "npy") -> Path:
    """VideoPose3D/save.py:31-61: one `.npy` holding a dict of nested lists
    {"fused_joints_3d", "left_joints_3d", "right_joints_3d"} (an object array, i.e. a pickle: readers use
    `np.load(path, allow_pickle=True).item()`).  Any other `fmt` raises ValueError like the reference."""
    if fmt != "npy":
        raise ValueError(f"Unsupported format: {fmt}")
    p = Path(save_path)
    p.parent.mkdir(parents=True, exist_ok=True)
    payload = {
        "fused_joints_3d": np.asarray(fused_joints_3d).tolist(),
        "left_joints_3d": np.asarray(left_joints_3d).tolist(),
        "right_joints_3d": np.asarray(right_joints_3d).tolist(),
    }
    np.save(p, payload)
    return p if p.suffix == ".npy" else p.with_suffix(p.suffix + ".npy")


def load_3d_joints(path) -> dict:
    """Reader of `save_3d_joints` files written by THIS package (unpickles: never point it at a file of
    unknown origin).  -> dict of float64 arrays."""
    d = np.load(Path(path), allow_pickle=True).item()
    return {k: np.asarray(d[k], dtype=np.float64) for k in ("fused_joints_3d", "left_joints_3d", "right_joints_3d")}


def save_predictions_npz(outdir, preds: dict) -> Path:
    """vggt/save.py:52-56: `<outdir>/predictions.npz` = np.savez of every array of the prediction dict
    (device tensors are brought to the host; None entries, e.g. pose_enc_list, are skipped as np.savez
    cannot hold them without pickling)."""
    out = Path(outdir)
    out.mkdir(parents=True, exist_ok=True)
    arrays = {}
    for k, v in preds.items():
        if v is None:
            continue
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        arrays[k] = np.asarray(v)
    np.savez(out / "predictions.npz", **arrays)
    return out / "predictions.npz"
