"""Counterpart of VideoPose3D/run.py's entry point with the reference's signature:

    run_video_pose_3d(config, pt_path, out_dir, args) -> (prediction [T, 17, 3], depth)     (run.py:107)

The inference slice of that function, in its order: the clip's 2D keypoints from the `.pt` file
(CustomDataset, common/custom_dataset.py:106-151), screen normalisation (run.py:191-199), the lifter built
from `args` (:224-262) and loaded from `config.model.ckpt_path` (:284-289), UnchunkedGenerator padding +
flip TTA (:1070-1081, common/generators.py:216-239), `evaluate(return_predictions=True)` (:961-989),
`<out_dir>/<video_name>.npy` with the camera-space joints (:1086-1092), and the returned joints turned by
the dummy H36M camera and rebased in height (:1094-1108).  Left out: training, the
evaluation protocols, the rendered GIF.  The model call is the HIP `TemporalModel` (vp3d.py).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import formats
from .infer import cfg_get
from .vp3d import (JOINTS_LEFT, JOINTS_RIGHT, KPS_LEFT, KPS_RIGHT, TemporalModel, merge_augmented,
                   normalize_screen_coordinates, pad_and_augment)

# common/custom_dataset.py:60-74: "Dummy camera parameters (taken from Human3.6M), only for visualization"
CUSTOM_CAMERA_ORIENTATION = np.array([0.1407056450843811, -0.1500701755285263, -0.755240797996521, 0.6223280429840088],
                                     dtype="float32")


def qrot(q: np.ndarray, v: np.ndarray) -> np.ndarray:
    """common/quaternion.py:10-24 (w first): v + 2 (w (q x v) + q x (q x v))"""
    qvec = q[..., 1:]
    uv = np.cross(qvec, v)
    uuv = np.cross(qvec, uv)
    return v + 2 * (q[..., :1] * uv + uuv)


def camera_to_world(X: np.ndarray, R: np.ndarray, t) -> np.ndarray:
    """common/camera.py:33-34"""
    return qrot(np.tile(R, (*X.shape[:-1], 1)), X) + t


def run_video_pose_3d(config, pt_path: Path, out_dir: Path, args, model_pos: TemporalModel = None):
    """Same arguments as the reference (`args` = the namespace of common/arguments.py; the fields read are
    architecture, causal, dropout, channels, dense, test_time_augmentation).  `model_pos` (optional) supplies
    an already loaded lifter instead of building one from `args` and `config.model.ckpt_path`."""
    pt_path, out_dir = Path(pt_path), Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    pt = formats._load_pt(pt_path)
    video_name = pt["video_name"]
    H, W = (int(v) for v in pt["img_shape"])
    kps = pt["detectron2"]["keypoints"]
    kps = (kps.numpy() if isinstance(kps, torch.Tensor) else np.asarray(kps)).copy()
    # run.py:191-199: normalised in place, in the array's own dtype
    kps[..., :2] = normalize_screen_coordinates(kps[..., :2], w=W, h=H)

    if model_pos is None:
        filter_widths = [int(x) for x in str(getattr(args, "architecture", "3,3,3,3,3")).split(",")]
        model_pos = TemporalModel(kps.shape[-2], kps.shape[-1], 17, filter_widths=filter_widths,
                                  causal=bool(getattr(args, "causal", False)), dropout=getattr(args, "dropout", 0.25),
                                  channels=int(getattr(args, "channels", 1024)), dense=bool(getattr(args, "dense", False)))
        chk = cfg_get(config, "model.ckpt_path")
        if chk is None:
            raise RuntimeError("config.model.ckpt_path is not set (run.py:284-289)")
        checkpoint = torch.load(str(chk), map_location="cpu", weights_only=True)
        model_pos.load_state_dict(checkpoint["model_pos"])
    receptive_field = model_pos.receptive_field()
    pad = (receptive_field - 1) // 2
    causal_shift = pad if model_pos.causal else 0
    augment = bool(getattr(args, "test_time_augmentation", True))
    batch_2d = pad_and_augment(kps, pad, causal_shift, augment, KPS_LEFT, KPS_RIGHT)
    with torch.no_grad():
        pred = model_pos(torch.from_numpy(batch_2d.astype("float32")).cuda())
        if augment:
            pred = merge_augmented(pred, JOINTS_LEFT, JOINTS_RIGHT)
    prediction = pred.squeeze(0).cpu().numpy()
    # Predictions are in camera space (run.py:1086-1092)
    np.save(out_dir / (str(video_name) + ".npy"), prediction)
    # run.py:1094-1108: invert the (dummy) camera rotation, rebase the height
    prediction = camera_to_world(prediction, R=CUSTOM_CAMERA_ORIENTATION, t=0)
    prediction[:, :, 2] -= np.min(prediction[:, :, 2])
    depth = pt.get("depth", None)
    if isinstance(depth, torch.Tensor):
        depth = depth.squeeze()
    return prediction, depth
