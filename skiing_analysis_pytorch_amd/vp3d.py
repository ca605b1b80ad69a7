"""Host-side mirror of the VideoPose3D lifter boundary.

`TemporalModel` has the constructor, `receptive_field()`, `load_state_dict()`, `eval()`,
`cuda()` and call signature of the reference class (VideoPose3D/common/model.py:79-138) so
that the call site `predicted_3d_pos = model_pos(inputs_2d)` (VideoPose3D/run.py:974) and
the weight load `model_pos.load_state_dict(checkpoint["model_pos"])` (run.py:288-289) work
unchanged; the forward itself is `skimi_vp3d_forward` in libskimi.so.

`lift_clip` restates the inference slice of `run_video_pose_3d` around it
(run.py:191-199 normalisation, :1070-1081 UnchunkedGenerator padding + flip TTA,
:979-986 un-flip + mean) with the host part in numpy, as the reference does.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from . import _lib
from ._lib import PREC_BF16, PREC_BF16X3, check, lib, ptr

# upstream-VideoPose3D conventions: COCO-17 2D keypoints, H36M-17 3D joints
KPS_LEFT = [1, 3, 5, 7, 9, 11, 13, 15]
KPS_RIGHT = [2, 4, 6, 8, 10, 12, 14, 16]
JOINTS_LEFT = [4, 5, 6, 11, 12, 13]
JOINTS_RIGHT = [1, 2, 3, 14, 15, 16]


class TemporalModel:
    """Drop-in for VideoPose3D.common.model.TemporalModel (inference only)."""

    def __init__(self, num_joints_in, in_features, num_joints_out, filter_widths, causal=False, dropout=0.25,
                 channels=1024, dense=False, prec=PREC_BF16X3):
        for fw in filter_widths:
            assert fw % 2 != 0, "Only odd filter widths are supported"  # model.py:20-21
        if dense:
            raise NotImplementedError("dense=True (ablation, model.py:114-116) is not on the hot path")
        self.num_joints_in = num_joints_in
        self.in_features = in_features
        self.num_joints_out = num_joints_out
        self.filter_widths = list(filter_widths)
        self.causal = bool(causal)
        self.channels = channels
        self.prec = prec
        fw = (C.c_int32 * len(filter_widths))(*filter_widths)
        self._h = lib().skimi_vp3d_create(num_joints_in, in_features, num_joints_out, fw, len(filter_widths),
                                          channels, int(causal))
        if not self._h:
            raise _lib.SkimiError(lib().skimi_last_error().decode())
        # model.py:31,105-110 — kept on the host too: callers read .pad / .causal_shift
        self.pad = [filter_widths[0] // 2]
        self.causal_shift = [(filter_widths[0] // 2) if causal else 0]
        nd = filter_widths[0]
        for i in range(1, len(filter_widths)):
            self.pad.append((filter_widths[i] - 1) * nd // 2)
            self.causal_shift.append((filter_widths[i] // 2 * nd) if causal else 0)
            nd *= filter_widths[i]
        self._ws = None
        self._finalized = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:   # module globals may already be torn down at interpreter exit
            try:
                lib().skimi_vp3d_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- reference API --------------------------------------------------------------
    def receptive_field(self) -> int:
        return int(lib().skimi_vp3d_receptive_field(self._h))

    def total_causal_shift(self) -> int:  # model.py:50-61
        frames = self.causal_shift[0]
        nd = self.filter_widths[0]
        for i in range(1, len(self.filter_widths)):
            frames += self.causal_shift[i] * nd
            nd *= self.filter_widths[i]
        return frames

    def eval(self):
        return self

    def cuda(self):
        return self

    def to(self, *_a, **_k):
        return self

    def load_state_dict(self, state_dict, strict=True):
        from .weights import vp3d_spec

        spec = vp3d_spec(self.num_joints_in, self.in_features, self.num_joints_out, self.filter_widths,
                         self.channels)
        missing = [k for k in spec if k not in state_dict]
        unexpected = [k for k in state_dict if k not in spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        for k, (shape, kind) in spec.items():
            if kind == "count" or k not in state_dict:
                continue
            t = state_dict[k].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(shape)}")
            check(lib().skimi_vp3d_set_weight(self._h, k.encode(), t.data_ptr(), t.numel()), f"set_weight({k})")
        check(lib().skimi_vp3d_finalize(self._h, self.prec), "skimi_vp3d_finalize")
        self._finalized = True
        return self

    def __call__(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        return self.forward(x, out)

    def forward(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """`out` (optional, beyond the reference signature): write the result into a caller buffer."""
        assert len(x.shape) == 4  # model.py:64-66
        assert x.shape[-2] == self.num_joints_in
        assert x.shape[-1] == self.in_features
        if not x.is_cuda:
            raise _lib.SkimiError("TemporalModel.forward needs a device tensor (the HIP path is the only path)")
        x = x.contiguous().to(torch.float32)
        B, L = x.shape[0], x.shape[1]
        rf = self.receptive_field()
        oshape = (B, L - rf + 1, self.num_joints_out, 3)
        if out is None:
            out = torch.empty(oshape, dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != oshape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
            raise _lib.SkimiError(f"out must be a contiguous float32 tensor of shape {oshape} on {x.device}")
        need = lib().skimi_vp3d_workspace_bytes(self._h, B, L)
        if need == 0:
            raise _lib.SkimiError(f"input of {L} frames is shorter than the receptive field {rf}")
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        check(lib().skimi_vp3d_forward(self._h, ptr(x), ptr(out), B, L, ptr(self._ws), self._ws.numel(),
                                       _lib.current_stream()), "skimi_vp3d_forward")
        return out


# ---- host pre/post (numpy, as in the reference) -----------------------------------------
def normalize_screen_coordinates(X, w, h):
    """VideoPose3D/common/camera.py:14-18"""
    assert X.shape[-1] == 2
    return X / w * 2 - [1, h / w]


def pad_and_augment(seq_2d: np.ndarray, pad: int, causal_shift: int, augment: bool,
                    kps_left: Sequence[int] = KPS_LEFT, kps_right: Sequence[int] = KPS_RIGHT) -> np.ndarray:
    """UnchunkedGenerator.next_epoch, VideoPose3D/common/generators.py:216-239:
    edge-pad the clip by (pad+shift, pad-shift); with TTA append the x-flipped copy with the
    left/right keypoints swapped.  Index path: bit-exact."""
    batch_2d = np.expand_dims(
        np.pad(seq_2d, ((pad + causal_shift, pad - causal_shift), (0, 0), (0, 0)), "edge"), axis=0)
    if augment:
        batch_2d = np.concatenate((batch_2d, batch_2d), axis=0)
        batch_2d[1, :, :, 0] *= -1
        batch_2d[1, :, list(kps_left) + list(kps_right)] = batch_2d[1, :, list(kps_right) + list(kps_left)]
    return batch_2d


def merge_augmented(pred: torch.Tensor, joints_left: Sequence[int] = JOINTS_LEFT,
                    joints_right: Sequence[int] = JOINTS_RIGHT) -> torch.Tensor:
    """VideoPose3D/run.py:979-986: undo the flip of copy 1 and average with copy 0."""
    pred = pred.clone()
    pred[1, :, :, 0] *= -1
    jl, jr = list(joints_left), list(joints_right)
    pred[1, :, jl + jr] = pred[1, :, jr + jl]
    return torch.mean(pred, dim=0, keepdim=True)


def lift_clip(model: TemporalModel, keypoints_px: np.ndarray, w: int, h: int, augment: bool = True,
              device="cuda") -> np.ndarray:
    """[T, 17, 2] pixel keypoints -> [T, 17, 3] (the `evaluate(return_predictions=True)` slice
    of run_video_pose_3d, VideoPose3D/run.py:191-199,1070-1083,961-989)."""
    kps = normalize_screen_coordinates(keypoints_px[..., :2].astype(np.float64), w=w, h=h)
    pad = (model.receptive_field() - 1) // 2
    shift = pad if model.causal else 0  # run.py:266-271
    batch_2d = pad_and_augment(kps, pad, shift, augment)
    x = torch.from_numpy(batch_2d.astype("float32")).to(device)
    pred = model(x)
    if augment:
        pred = merge_augmented(pred)
    return pred.squeeze(0).cpu().numpy()
