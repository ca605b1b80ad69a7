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


TINY_CONFIGS = {
    # conv patch embed, everything on: exercises aggregator, camera, both DPT heads, track head
    "tiny_conv": dict(S=3, H=140, W=140, queries=5, cfg=dict(
        img_size=140, embed_dim=128, depth=4, num_heads=2, patch_embed="conv", cam_trunk_depth=2, cam_heads=2,
        dpt_features=128, dpt_out_channels=(64, 128, 256, 256), dpt_layers=(0, 1, 2, 3), track_features=64,
        track_hidden=128, track_corr_levels=3, track_corr_radius=3, track_iters=3, track_depth=2,
        track_heads=8, track_virtual=16)),
    # DINOv2 ViT-S/14-reg patch embed (12 blocks, LayerNorm eps 1e-6, LayerScale), no track head
    "tiny_dino_rect": dict(S=2, H=42, W=70, queries=0, cfg=dict(   # non-square: resized pos_embed
        img_size=70, embed_dim=384, depth=2, num_heads=6, patch_embed="dinov2_vits14_reg", dino_depth=12,
        dino_heads=6, cam_trunk_depth=1, cam_heads=6, dpt_features=64, dpt_out_channels=(64, 64, 128, 128),
        dpt_layers=(0, 0, 1, 1), enable_track=False)),
    "tiny_dino": dict(S=2, H=70, W=70, queries=0, cfg=dict(
        img_size=70, embed_dim=384, depth=2, num_heads=6, patch_embed="dinov2_vits14_reg", dino_depth=12,
        dino_heads=6, cam_trunk_depth=1, cam_heads=6, dpt_features=64, dpt_out_channels=(64, 64, 128, 128),
        dpt_layers=(0, 0, 1, 1), enable_track=False)),
}


def build_reference_vggt(cfg: "W.VGGTConfig"):
    """The reference VGGT module with non-default sizes: VGGT.__init__ hard-codes VGGT-1B, so the
    sub-modules are built with the reference's own classes and attached under the same attribute
    names; forward() is the reference's VGGT.forward unchanged."""
    import torch.nn as nn
    from vggt.vggt.heads.camera_head import CameraHead
    from vggt.vggt.heads.dpt_head import DPTHead
    from vggt.vggt.heads.track_head import TrackHead
    from vggt.vggt.models.aggregator import Aggregator
    from vggt.vggt.models.vggt import VGGT

    m = VGGT.__new__(VGGT)
    nn.Module.__init__(m)
    m.aggregator = Aggregator(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim,
                              depth=cfg.depth, num_heads=cfg.num_heads, patch_embed=cfg.patch_embed)
    D = 2 * cfg.embed_dim
    m.camera_head = CameraHead(dim_in=D, trunk_depth=cfg.cam_trunk_depth, num_heads=cfg.cam_heads) if cfg.enable_camera else None
    dk = dict(dim_in=D, features=cfg.dpt_features, out_channels=list(cfg.dpt_out_channels),
              intermediate_layer_idx=list(cfg.dpt_layers))
    m.point_head = DPTHead(output_dim=4, activation="inv_log", conf_activation="expp1", **dk) if cfg.enable_point else None
    m.depth_head = DPTHead(output_dim=2, activation="exp", conf_activation="expp1", **dk) if cfg.enable_depth else None
    if cfg.enable_track:
        th = TrackHead(dim_in=D, patch_size=cfg.patch_size, features=cfg.track_features, iters=cfg.track_iters,
                       corr_levels=cfg.track_corr_levels, corr_radius=cfg.track_corr_radius,
                       hidden_size=cfg.track_hidden)
        # TrackHead builds its DPT with default out_channels/layers and its tracker with depth 6 /
        # 64 virtual tracks; rebuild those two with the reference classes for the tiny sizes
        from vggt.vggt.heads.track_modules.base_track_predictor import BaseTrackerPredictor
        from vggt.vggt.heads.track_modules.blocks import EfficientUpdateFormer
        th.feature_extractor = DPTHead(dim_in=D, patch_size=cfg.patch_size, features=cfg.track_features,
                                       out_channels=list(cfg.dpt_out_channels),
                                       intermediate_layer_idx=list(cfg.dpt_layers), feature_only=True,
                                       down_ratio=2, pos_embed=False)
        th.tracker = BaseTrackerPredictor(latent_dim=cfg.track_features, predict_conf=True, stride=2,
                                          corr_levels=cfg.track_corr_levels, corr_radius=cfg.track_corr_radius,
                                          hidden_size=cfg.track_hidden, depth=cfg.track_depth)
        th.tracker.updateformer = EfficientUpdateFormer(
            space_depth=cfg.track_depth, time_depth=cfg.track_depth, input_dim=3 * cfg.track_features + 4,
            hidden_size=cfg.track_hidden, num_heads=cfg.track_heads, output_dim=cfg.track_features + 2,
            mlp_ratio=4.0, add_space_attn=True, num_virtual_tracks=cfg.track_virtual)
        m.track_head = th
    else:
        m.track_head = None
    return m.eval()


def run_reference_vggt(name, cfgd, S, H, W_, nq, seed=0):
    cfg = W.VGGTConfig(**cfgd)
    sd = W.make_vggt_state_dict(cfg, seed=seed)
    m = build_reference_vggt(cfg)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    images = W.make_images(S, H, W_, seed=seed + 1)
    q = None
    if nq:
        g = torch.Generator().manual_seed(seed + 2)
        q = torch.rand((nq, 2), generator=g) * torch.tensor([W_ - 20.0, H - 20.0]) + 10.0
    captured = {}
    hook = m.aggregator.register_forward_hook(lambda mod, i, o: captured.__setitem__("tokens", o[0]))
    preds = m(images, query_points=q)
    hook.remove()
    out = {"images_seed": np.array(seed + 1), "S": np.array(S), "H": np.array(H), "W": np.array(W_)}
    if q is not None:
        out["query_points"] = q.numpy()
    for k, v in preds.items():
        if k == "images":
            continue
        if k == "pose_enc_list":
            out[k] = torch.stack(v).numpy()
        else:
            out[k] = v.numpy()
    toks = captured["tokens"]
    out["tokens_last"] = toks[-1].numpy()
    out["tokens_first"] = toks[0].numpy()
    return cfg, out


def gen_vggt_tiny():
    import json

    for name, spec in TINY_CONFIGS.items():
        cfg, out = run_reference_vggt(name, spec["cfg"], spec["S"], spec["H"], spec["W"], spec["queries"])
        np.savez_compressed(GOLD / f"vggt_{name}.npz", cfg_json=np.array(json.dumps(spec["cfg"])), seed=np.array(0), **out)
        print("wrote", f"vggt_{name}.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


GENERATORS = {"vp3d": gen_vp3d, "vggt_tiny": gen_vggt_tiny}



def gen_fuse():
    """fuse/fuse.py (pure NumPy): fuse_frame_3d + temporal_smooth_ema on a synthetic MHR-70-id
    sequence with missing joints (NaN) on either side."""
    from fuse.fuse import fuse_frame_3d, temporal_smooth_ema

    rng = np.random.default_rng(0)
    ids = [1, 2, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 41, 62, 69, 20, 33]
    T, J = 40, len(ids)
    base = rng.normal(size=(1, J, 3))
    L = base + np.cumsum(rng.normal(scale=0.03, size=(T, J, 3)), axis=0)
    R = L + rng.normal(scale=0.02, size=(T, J, 3))
    miss_l = rng.random((T, J)) < 0.08
    miss_r = rng.random((T, J)) < 0.08
    ql, qr = rng.normal(size=(T, J)), rng.normal(size=(T, J))
    fused = np.full((T, J, 3), np.nan)
    seq = []
    for t in range(T):
        dl = {jid: L[t, j] for j, jid in enumerate(ids) if not miss_l[t, j]}
        dr = {jid: R[t, j] for j, jid in enumerate(ids) if not miss_r[t, j]}
        if not dl:
            dl = {ids[0]: L[t, 0]}
        if not dr:
            dr = {ids[0]: R[t, 0]}
        f = fuse_frame_3d(dl, dr, ql[t], qr[t], ids)
        seq.append(f)
        for j, jid in enumerate(ids):
            if jid in f:
                fused[t, j] = f[jid]
    out = {}
    for name, kw in (("adaptive", dict()), ("plain", dict(adaptive=False, alpha=0.6))):
        sm = temporal_smooth_ema(seq, ids, **kw)
        Y = np.full((T, J, 3), np.nan)
        for t in range(T):
            for j, jid in enumerate(ids):
                if jid in sm[t]:
                    Y[t, j] = sm[t][jid]
        out["smooth_" + name] = Y
    np.savez_compressed(GOLD / "fuse_ema.npz", ids=np.array(ids), L=np.where(miss_l[..., None], np.nan, L),
                        R=np.where(miss_r[..., None], np.nan, R), ql=ql, qr=qr, fused=fused, **out)
    print("wrote fuse_ema.npz")


GENERATORS["fuse"] = gen_fuse


def gen_fuse_align():
    """The rest of SURVEY f2: fuse/main_raw.py (Kabsch right->left alignment), fuse/confidence.py
    (weak-perspective reprojection and cross-view consistency confidences) and
    VideoPose3D/fuse/fuse.py (H36M-17 left/right fusion without extrinsics) on synthetic poses with
    missing joints.  The reference functions print diagnostics; stdout is silenced."""
    import contextlib, io
    from fuse.main_raw import _kabsch_rigid_align, _align_right_to_left
    from fuse.confidence import weakpersp_reproj_confidence, crossview_consistency_confidence
    from VideoPose3D.fuse.fuse import fuse_pose_no_extrinsics_h36m

    rng = np.random.default_rng(1)
    out = {}
    # --- Kabsch: 17 COCO joints, right view = rotated + translated left view + noise, NaNs on both sides
    ids = list(range(17))
    J = len(ids)
    A = rng.normal(size=(6, J, 3))
    ang = rng.normal(size=(6, 3))
    Rm = []
    for a in ang:
        cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
        Rm.append(np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                  @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    Rm = np.stack(Rm)
    Bv = np.einsum("tij,tkj->tki", Rm, A) + rng.normal(size=(6, 1, 3)) + rng.normal(scale=0.01, size=(6, J, 3))
    Bv[5] = -Bv[5]                                   # a reflected frame: exercises the det < 0 branch
    missA = rng.random((6, J)) < 0.15
    missB = rng.random((6, J)) < 0.15
    missA[4] = True; missA[4, :2] = False            # fewer than 3 common joints -> right view returned as is
    A_n = np.where(missA[..., None], np.nan, A)
    B_n = np.where(missB[..., None], np.nan, Bv)
    aligned = np.full((6, J, 3), np.nan)
    rots, trs = [], []
    for t in range(6):
        dl = {j: A_n[t, j] for j in ids if not missA[t, j]}
        dr = {j: B_n[t, j] for j in ids if not missB[t, j]}
        res = _align_right_to_left(dl, dr, ids)
        for j in ids:
            if j in res:
                aligned[t, j] = res[j]
        v = ~missA[t] & ~missB[t]
        if v.sum() >= 3:
            r, tr = _kabsch_rigid_align(B_n[t][v], A_n[t][v])
        else:
            r, tr = np.full((3, 3), np.nan), np.full(3, np.nan)
        rots.append(r); trs.append(tr)
    out.update(kab_left=A_n, kab_right=B_n, kab_aligned=aligned, kab_R=np.stack(rots), kab_t=np.stack(trs))
    # --- confidences (dict inputs in the reference: every joint present, NaN allowed in the values)
    X3 = rng.normal(size=(4, J, 3))
    M = np.linalg.qr(rng.normal(size=(3, 3)))[0][:, :2]
    U2 = 180.0 * (X3 @ M) + np.array([320.0, 240.0]) + rng.normal(scale=4.0, size=(4, J, 2))
    X3[1, 3] = np.nan; U2[2, 5] = np.nan
    conf, err, uhat, ps, pM, pt = [], [], [], [], [], []
    for t in range(4):
        c, e, uh, prm = weakpersp_reproj_confidence({j: X3[t, j] for j in ids}, {j: U2[t, j] for j in ids}, sigma_px=12.0)
        conf.append(c); err.append(e); uhat.append(uh); ps.append(prm["s"]); pM.append(prm["M"]); pt.append(prm["t"])
    out.update(wp_X=X3, wp_U=U2, wp_conf=np.stack(conf), wp_err=np.stack(err), wp_uhat=np.stack(uhat), wp_s=np.array(ps),
               wp_M=np.stack(pM), wp_t=np.stack(pt))
    Xa = rng.normal(size=(4, J, 3))
    Xb = np.einsum("ij,tkj->tki", Rm[0], Xa) * 1.7 + rng.normal(scale=0.05, size=(4, J, 3))
    Xa[2, 9] = np.nan
    Xb[3, 11] = np.nan                                # a key joint missing -> everything NaN / conf 0
    kw = dict(root_idx=0, left_hip_idx=11, right_hip_idx=12, left_shoulder_idx=5, right_shoulder_idx=6)
    cc, dd, xa, xb = [], [], [], []
    for t in range(4):
        for mode in ("hip", "torso"):
            c, d, a_c, b_c, _ = crossview_consistency_confidence({j: Xa[t, j] for j in ids}, {j: Xb[t, j] for j in ids},
                                                                 sigma_3d=0.3, scale_mode=mode, **kw)
            cc.append(c); dd.append(d); xa.append(a_c); xb.append(b_c)
    out.update(cv_A=Xa, cv_B=Xb, cv_conf=np.stack(cc), cv_dist=np.stack(dd), cv_Ac=np.stack(xa), cv_Bc=np.stack(xb))
    # --- VideoPose3D left/right fusion (H36M-17)
    T = 12
    Lh = rng.normal(size=(T, 17, 3))
    Rh = np.einsum("ij,tkj->tki", Rm[1], Lh) + rng.normal(scale=0.04, size=(T, 17, 3)) + rng.normal(size=(T, 1, 3))
    Lh[3, 13] = np.nan; Rh[4, 16] = np.nan; Lh[5, 2] = np.nan; Rh[5, 2] = np.nan
    wl, wr = rng.random((T, 17)), rng.random((T, 17))
    with contextlib.redirect_stdout(io.StringIO()):
        f0, d0 = fuse_pose_no_extrinsics_h36m(Lh, Rh, tau=0.08)
        f1, d1 = fuse_pose_no_extrinsics_h36m(Lh, Rh, tau=0.3, allow_scale=True, mirror_right_x=True, wL=wl, wR=wr)
        f2, _ = fuse_pose_no_extrinsics_h36m(Lh[0], Rh[0], tau=0.5, wL=wl[0], wR=wr[0], return_diagnostics=False)
    out.update(h36_L=Lh, h36_R=Rh, h36_wl=wl, h36_wr=wr, h36_f0=f0, h36_f1=f1, h36_f2=f2,
               h36_gain0=np.array([d["gain"] for d in d0["per_frame"]]), h36_gain1=np.array([d["gain"] for d in d1["per_frame"]]),
               h36_R0=np.stack([d["R"] for d in d0["per_frame"]]), h36_s1=np.array([d["s"] for d in d1["per_frame"]]),
               h36_bad0=np.array(d0["bad_frames"], dtype=np.int64), h36_mean_gain0=np.array(d0["mean_gain"]))
    np.savez_compressed(GOLD / "fuse_align.npz", **out)
    print("wrote fuse_align.npz")


GENERATORS["fuse_align"] = gen_fuse_align


def gen_geometry():
    """The geometry post-processing of the path, from the reference's own functions:
    pose_encoding_to_extri_intri (vggt/vggt/utils/pose_enc.py:62-124), quat_to_mat (utils/rotation.py:14-44),
    unproject_depth_map_to_point_map (utils/geometry.py:15-117), camera_to_world / qrot
    (VideoPose3D/common/camera.py:33-34, quaternion.py:10-24), mpjpe (VideoPose3D/common/loss.py:11-17).
    (vggt/triangulate.py and vggt/multi_view_process.py import cv2 / open3d and cannot be imported here:
    the DLT and the person-origin helpers stay pinned by the oracle's restatement only.)"""
    from VideoPose3D.common.camera import camera_to_world
    from VideoPose3D.common.loss import mpjpe
    from vggt.vggt.utils.geometry import unproject_depth_map_to_point_map
    from vggt.vggt.utils.pose_enc import pose_encoding_to_extri_intri
    from vggt.vggt.utils.rotation import quat_to_mat

    g = torch.Generator().manual_seed(21)
    pe = torch.randn((2, 5, 9), generator=g) * 0.3
    pe[..., 3:7] += torch.tensor([0.0, 0.0, 0.0, 1.0])
    pe[..., 2] += 3.0
    pe[..., 7:] = 0.6 + 0.5 * torch.rand((2, 5, 2), generator=g)
    E, K = pose_encoding_to_extri_intri(pe, (294, 518))
    q = torch.randn((7, 4), generator=g)
    depth = torch.rand((5, 24, 36, 1), generator=g) * 4 + 0.5
    wp = unproject_depth_map_to_point_map(depth.numpy(), E[0].numpy(), K[0].numpy())
    pred = torch.randn((11, 17, 3), generator=g).numpy().astype("float32")
    rot = np.array([0.1407056450843811, -0.1500701755285263, -0.755240797996521, 0.6223280429840088], dtype="float32")
    world = camera_to_world(pred.copy(), R=rot, t=0)
    a, b = torch.randn((4, 9, 17, 3), generator=g), torch.randn((4, 9, 17, 3), generator=g)
    np.savez_compressed(GOLD / "geometry.npz", pose_enc=pe.numpy(), image_hw=np.array([294, 518]), extrinsic=E.numpy(),
                        intrinsic=K.numpy(), quat=q.numpy(), rotmat=quat_to_mat(q).numpy(), depth=depth.numpy(),
                        world_points=wp, cam_pred=pred, cam_rot=rot, cam_world=world, mpjpe_a=a.numpy(), mpjpe_b=b.numpy(),
                        mpjpe=np.array(float(mpjpe(a, b))))
    print("wrote geometry.npz")


GENERATORS["geometry"] = gen_geometry


def gen_formats():
    """a file written by the reference's OWN writer, VideoPose3D/save.py:31-61 `save_3d_joints` (imports
    cleanly here: logging, pathlib, numpy) -- the fused / left / right joints of a clip as one `.npy` holding a dict of
    nested lists -- next to the arrays it was given.  The build's writer (formats.save_3d_joints) must produce the same
    bytes, its reader the same values.  (vggt/save.py and VideoPose3D/run.py import cv2 / trimesh / omegaconf and cannot
    be imported: predictions.npz, the camera NPZ and the pose .npy stay pinned by the cited lines only.)"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_vp3d_save", os.path.join(REF, "VideoPose3D", "save.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(33)
    T = 4
    fused = rng.normal(size=(T, 17, 3))
    left = rng.normal(size=(T, 17, 3)).astype(np.float32)      # float32 inputs: tolist() widens them to Python floats
    right = rng.normal(size=(T, 17, 3))
    fused[1, 5] = np.nan                                       # a missing joint
    right[3, 0, 2] = np.inf
    out = GOLD / "formats_3d_joints.npy"
    mod.save_3d_joints(fused, left, right, out)
    np.savez(GOLD / "formats_3d_joints_inputs.npz", fused=fused, left=left, right=right)
    print("wrote", out.name, out.stat().st_size, "bytes (by the reference's save_3d_joints) + its inputs")


GENERATORS["formats"] = gen_formats


if __name__ == "__main__":
    which = sys.argv[1:] or list(GENERATORS)
    for w in which:
        GENERATORS[w]()
