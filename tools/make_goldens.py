#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own model classes
(imported from /root/reference, never copied) on this build's deterministic synthetic
weights and inputs.  Runs only in the build container (the GPU box has no reference).

Fixtures hold data only: seeds/config, inputs, and the reference's outputs.  Weights are
rebuilt from (spec, seed) by skiing_analysis_pytorch_amd.weights on whichever machine runs
the tests, so they are not stored.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py [vp3d|vggt_tiny|...]
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
REF = os.environ.get("SKIMI_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

GOLD = ROOT / "tests" / "golden"
GOLD.mkdir(parents=True, exist_ok=True)

from skiing_analysis_pytorch_amd import weights as W  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def gen_vp3d():
    from VideoPose3D.common.camera import normalize_screen_coordinates
    from VideoPose3D.common.generators import UnchunkedGenerator
    from VideoPose3D.common.model import TemporalModel

    kps_left, kps_right = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]
    joints_left, joints_right = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    W_, H_ = 1920, 1080
    for name, fw, causal, frames in (
        ("rf27", [3, 3, 3], False, 243),
        ("rf27_causal", [3, 3, 3], True, 60),
        ("rf243", [3, 3, 3, 3, 3], False, 243),
        ("rf81_w5", [3, 3, 3, 3], False, 40),
    ):
        sd = W.make_vp3d_state_dict(seed=0, filter_widths=fw)
        m = TemporalModel(17, 2, 17, filter_widths=fw, causal=causal, channels=1024).eval()
        m.load_state_dict(sd, strict=True)
        kp = W.make_keypoints_2d(frames=frames, seed=1).numpy()
        kps_n = normalize_screen_coordinates(kp.astype(np.float64), w=W_, h=H_)
        rf = m.receptive_field()
        pad = (rf - 1) // 2
        shift = pad if causal else 0
        out = {}
        for aug in (False, True):
            gen = UnchunkedGenerator(None, None, [kps_n], pad=pad, causal_shift=shift, augment=aug,
                                     kps_left=kps_left, kps_right=kps_right, joints_left=joints_left,
                                     joints_right=joints_right)
            for _, _, batch_2d in gen.next_epoch():
                x = torch.from_numpy(batch_2d.astype("float32"))
                pred = m(x)
                out[f"batch2d_aug{int(aug)}"] = batch_2d.astype(np.float32)
                out[f"raw_aug{int(aug)}"] = pred.numpy().copy()
                if aug:  # VideoPose3D/run.py:979-986
                    pred[1, :, :, 0] *= -1
                    pred[1, :, joints_left + joints_right] = pred[1, :, joints_right + joints_left]
                    pred = torch.mean(pred, dim=0, keepdim=True)
                out[f"pred_aug{int(aug)}"] = pred.squeeze(0).numpy()
        np.savez_compressed(GOLD / f"vp3d_{name}.npz", filter_widths=np.array(fw), causal=np.array(causal),
                            seed=np.array(0), kp_seed=np.array(1), frames=np.array(frames), w=np.array(W_),
                            h=np.array(H_), keypoints_px=kp, receptive_field=np.array(rf), **out)
        print("wrote", f"vp3d_{name}.npz", {k: v.shape for k, v in out.items()})


GENERATORS = {"vp3d": gen_vp3d}

if __name__ == "__main__":
    which = sys.argv[1:] or list(GENERATORS)
    for w in which:
        GENERATORS[w]()
