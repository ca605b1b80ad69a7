"""GPU: the reference-signature entry points (SURVEY §8 b) on `.pt` clips that embed their frames:

    process_multi_view_video(left_video_path, left_pt_path, right_video_path, right_pt_path, out_root,
                             inference_output_path, cfg)                 vggt/multi_view_process.py:68-76
    process_single_view_video(video_path, pt_path, out_root, inference_output_path, cfg)
                                                                         vggt/single_view_process.py:90-96
    run_video_pose_3d(config, pt_path, out_dir, args)                    VideoPose3D/run.py:107
    CameraHead.reconstruct_from_frames(frame_id, imgs)                   vggt/vggt/infer.py:157-215

checked against the same chain built from the CPU oracle (host PIL preprocessing -> oracle forward ->
oracle geometry -> NumPy SVD DLT), and the device geometry kernels against the reference's own outputs
(tests/golden/geometry.npz)."""
import json
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import vggt_oracle, vp3d_oracle
from skiing_analysis_pytorch_amd import formats, geometry, infer, vggt, weights as W
from skiing_analysis_pytorch_amd import multi_view_process as mv
from skiing_analysis_pytorch_amd import run as vp_run
from skiing_analysis_pytorch_amd import single_view_process as sv
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3

pytestmark = pytest.mark.gpu


def test_device_geometry_matches_reference_outputs(golden_dir):
    g = np.load(golden_dir / "geometry.npz")
    hw = tuple(int(v) for v in g["image_hw"])
    E, K = geometry.pose_encoding_to_extri_intri(torch.from_numpy(g["pose_enc"]).cuda(), hw)
    assert (E.cpu().numpy() - g["extrinsic"]).__abs__().max() < 1e-5
    assert (np.abs(K.cpu().numpy() - g["intrinsic"]) / (np.abs(g["intrinsic"]) + 1)).max() < 1e-5
    wp = geometry.unproject_depth_map_to_point_map(torch.from_numpy(g["depth"]).cuda(), E[0], K[0])
    assert np.abs(wp.cpu().numpy() - g["world_points"]).max() < 1e-4


@pytest.fixture(scope="module")
def tiny(golden_dir):
    g = np.load(golden_dir / "vggt_tiny_conv.npz")
    cfg = W.VGGTConfig(**json.loads(str(g["cfg_json"])))
    sd = W.make_vggt_state_dict(cfg, seed=0)
    m = vggt.VGGT(config=cfg, prec=PREC_BF16X3, head_prec=PREC_BF16X3)
    m.load_state_dict(sd)
    return cfg, sd, m


def _clip_pt(path, name, frames, kps, boxes, depth=None):
    T, H, Wd = frames.shape[:3]
    torch.save({"video_name": name, "video_path": f"/videos/{name}.mp4", "frame_count": T, "img_shape": (H, Wd), "fps": 30,
                "detectron2": {"bbox": torch.from_numpy(boxes), "keypoints": torch.from_numpy(kps),
                               "keypoints_score": torch.ones(T, 17)},
                "depth": depth if depth is not None else torch.zeros(T, 1, 4, 4), "frames": frames}, path)


def _oracle_step(cfg, sd, imgs_u8):
    """the reference chain on the host for one view stack: PIL preprocessing -> oracle -> E, K, R, t, C, points"""
    x = infer.load_and_preprocess_images(imgs_u8)                     # host path (PIL), as the reference
    d = cfg.to_dict()
    d["enable_point"] = False
    d["enable_track"] = False
    with torch.no_grad():
        ref = vggt_oracle.vggt_forward(sd, x, d)
    oh, ow = x.shape[-2:]
    E, K = vggt_oracle.pose_encoding_to_extri_intri(ref["pose_enc"], (oh, ow))
    E, K = E[0].numpy(), K[0].numpy()
    wp = vggt_oracle.unproject_depth_map_to_point_map(ref["depth"][0].numpy(), E, K)
    R, t, C = geometry.extrinsic_to_RT(E)
    H, Wd = imgs_u8[0].shape[:2]
    Kr = [geometry.scale_intrinsics(K[i], (oh, ow), (H, Wd)) for i in range(len(K))]
    return E, Kr, R, t, C, wp


def test_reconstruct_from_frames_matches_oracle_chain(tiny, tmp_path):
    cfg, sd, m = tiny
    rng = np.random.default_rng(0)
    imgs = [torch.from_numpy(rng.integers(0, 256, (135, 240, 3), dtype=np.uint8)) for _ in range(3)]
    head = infer.CameraHead({"infer": {"gpu": 0}}, tmp_path / "vggt_infer", model=m)
    E, Kr, R, t, C, wp = head.reconstruct_from_frames(frame_id=7, imgs=imgs)
    Eo, Kro, Ro, to, Co, wpo = _oracle_step(cfg, sd, imgs)
    assert E.shape == (3, 3, 4) and len(Kr) == 3 and Kr[0].shape == (3, 3) and wp.shape == (3, 294, 518, 3)
    assert np.abs(E - Eo).max() < 1e-3 and np.abs(R - Ro).max() < 1e-3 and np.abs(t - to).max() < 1e-3
    assert np.abs(C - Co).max() < 1e-3
    assert max((np.abs(a - b) / (np.abs(b) + 1)).max() for a, b in zip(Kr, Kro)) < 1e-3
    assert (np.abs(wp - wpo) / (np.abs(wpo) + 1)).max() < 1e-3
    # the per-frame predictions.npz (camera arrays of vggt/save.py:52-56)
    z = np.load(tmp_path / "vggt_infer" / "frame_0007" / "predictions.npz")
    assert np.array_equal(z["extrinsic"], E) and z["intrinsic"].shape == (3, 3, 3) and z["pose_enc"].shape == (3, 9)
    # run_vggt keeps the reference's return convention: numpy, batch axis squeezed, pose_enc_list None
    out, oh, ow = head.run_vggt(imgs)
    assert (oh, ow) == (294, 518) and out["pose_enc_list"] is None and out["depth"].shape == (3, 294, 518, 1)
    assert np.abs(out["extrinsic"] - Eo).max() < 1e-3


def test_process_multi_view_video_writes_reference_outputs(tiny, tmp_path):
    cfg, sd, m = tiny
    rng = np.random.default_rng(1)
    T, H, Wd = 5, 135, 240
    lf = torch.from_numpy(rng.integers(0, 256, (T, H, Wd, 3), dtype=np.uint8))
    rf = torch.from_numpy(rng.integers(0, 256, (T, H, Wd, 3), dtype=np.uint8))
    lk = (rng.random((T, 17, 2)) * [Wd - 40, H - 40] + 20).astype(np.float32)
    rk = (rng.random((T, 17, 2)) * [Wd - 40, H - 40] + 20).astype(np.float32)
    lb = np.tile(np.array([[60, 30, 180, 110]], np.float32), (T, 1))
    rb = np.tile(np.array([[50, 20, 170, 120]], np.float32), (T, 1))
    (tmp_path / "subj01").mkdir()
    _clip_pt(tmp_path / "subj01" / "left.pt", "left", lf, lk, lb)
    _clip_pt(tmp_path / "subj01" / "right.pt", "right", rf, rk, rb)
    head = infer.CameraHead({"infer": {"gpu": 0}}, None, model=m)
    out_dir = mv.process_multi_view_video(tmp_path / "subj01" / "left.mp4", tmp_path / "subj01" / "left.pt",
                                          tmp_path / "subj01" / "right.mp4", tmp_path / "subj01" / "right.pt",
                                          tmp_path / "out", tmp_path / "inference", {"infer": {"gpu": 0, "hflip": False}},
                                          camera_head=head, steps_per_call=2)
    assert out_dir == tmp_path / "out" / "multi_view" / "subj01"
    z = np.load(tmp_path / "inference" / "subj01_multi_view_3d_info.npz")       # vggt/save.py:84-110 (+ x3d)
    assert z["camera_intrinsics"].shape == (T, 2, 3, 3) and z["R"].shape == (T, 2, 3, 3)
    assert z["t"].shape == (T, 2, 3) and z["C"].shape == (T, 2, 3) and z["x3d"].shape == (T, 17, 3)
    assert (out_dir / "vggt_infer" / "frame_0004" / "predictions.npz").exists()
    # BASELINE config 4's tail: the gathered joints smoothed by fuse/'s EMA (fuse/fuse.py:329-412), and the marker that
    # these arrays are the pre-ICP quantities (the reference stores them after its Open3D refinement)
    from skiing_analysis_pytorch_amd import fuse
    assert z["icp_refined"].item() is False or not bool(z["icp_refined"])
    assert z["x3d_smoothed"].shape == (T, 17, 3)
    assert np.array_equal(z["x3d_smoothed"], fuse.temporal_smooth_ema(z["x3d"].astype(np.float64)))
    # the same chain from the oracle (per step: S = 2)
    for i in (0, 3):
        E, Kr, R, t, C, wp = _oracle_step(cfg, sd, [lf[i], rf[i]])
        pl = mv.extract_person_points(wp[0], lb[i], (H, Wd))
        pr = mv.extract_person_points(wp[1], rb[i], (H, Wd))
        origin = 0.5 * (pl.mean(axis=0) + pr.mean(axis=0))
        R2, t2 = mv.recenter_and_align(R, t, origin)
        assert np.abs(z["R"][i] - R2).max() < 1e-3 and np.abs(z["t"][i] - t2).max() / (np.abs(t2).max() + 1) < 1e-3
        assert np.abs(z["C"][i] - C).max() < 1e-3
        x3d = vggt_oracle.triangulate_one_frame(np.stack(Kr).astype(np.float64), R2, t2, np.stack([lk[i], rk[i]]).astype(np.float64))
        # random pixel pairs are not consistent observations: compare through the reprojection the DLT minimises
        got = z["x3d"][i].astype(np.float64)
        for v in range(2):
            P = np.stack(Kr)[v] @ np.concatenate([R2[v], t2[v][:, None]], axis=1)
            pa = P @ np.concatenate([got, np.ones((17, 1))], axis=1).T
            pb = P @ np.concatenate([x3d.astype(np.float64), np.ones((17, 1))], axis=1).T
            d = np.abs(pa[:2] / pa[2] - pb[:2] / pb[2])
            assert np.median(d) < 0.5, (i, v, np.median(d))


def test_process_multi_view_video_hflip(tiny, tmp_path):
    """cfg.infer.hflip (multi_view_process.py:116-127): right frames mirrored, right keypoints / boxes mapped"""
    cfg, sd, m = tiny
    rng = np.random.default_rng(2)
    T, H, Wd = 2, 135, 240
    lf = torch.from_numpy(rng.integers(0, 256, (T, H, Wd, 3), dtype=np.uint8))
    rf = torch.from_numpy(rng.integers(0, 256, (T, H, Wd, 3), dtype=np.uint8))
    k = (rng.random((T, 17, 2)) * [Wd - 40, H - 40] + 20).astype(np.float32)
    b = np.tile(np.array([[60, 30, 180, 110]], np.float32), (T, 1))
    (tmp_path / "s").mkdir()
    _clip_pt(tmp_path / "s" / "l.pt", "l", lf, k, b)
    _clip_pt(tmp_path / "s" / "r.pt", "r", rf, k, b)
    _clip_pt(tmp_path / "s" / "rflip.pt", "rflip", torch.flip(rf, [2]), np.stack([Wd - k[..., 0], k[..., 1]], -1).astype(np.float32),
             np.stack([Wd - b[:, 2], b[:, 1], Wd - b[:, 0], b[:, 3]], -1).astype(np.float32))
    head = infer.CameraHead(None, None, model=m)
    a = tmp_path / "a"
    mv.process_multi_view_video(tmp_path / "s" / "l.mp4", tmp_path / "s" / "l.pt", tmp_path / "s" / "r.mp4", tmp_path / "s" / "r.pt",
                                a, a / "inf", SimpleNamespace(infer=SimpleNamespace(gpu=0, hflip=True)), camera_head=head)
    bdir = tmp_path / "b"
    mv.process_multi_view_video(tmp_path / "s" / "l.mp4", tmp_path / "s" / "l.pt", tmp_path / "s" / "r.mp4", tmp_path / "s" / "rflip.pt",
                                bdir, bdir / "inf", {"infer": {"hflip": False}}, camera_head=head)
    za, zb = np.load(a / "inf" / "s_multi_view_3d_info.npz"), np.load(bdir / "inf" / "s_multi_view_3d_info.npz")
    for key in ("R", "t", "C", "camera_intrinsics"):     # split-K atomics: not bit-reproducible from run to run
        assert (np.abs(za[key] - zb[key]) / (np.abs(zb[key]) + 1)).max() < 1e-4, key


def test_process_single_view_video(tiny, tmp_path):
    cfg, sd, m = tiny
    rng = np.random.default_rng(4)
    T, H, Wd = 95, 135, 240                         # frames 0, 30, 60, 90 -> S = 4
    fr = torch.from_numpy(rng.integers(0, 256, (T, H, Wd, 3), dtype=np.uint8))
    k = (rng.random((T, 17, 2)) * [Wd, H]).astype(np.float32)
    (tmp_path / "skier").mkdir()
    _clip_pt(tmp_path / "skier" / "cam.pt", "cam", fr, k, np.tile(np.array([[0, 0, 10, 10]], np.float32), (T, 1)))
    head = infer.CameraHead(None, None, model=m)
    out_dir = sv.process_single_view_video(tmp_path / "skier" / "cam.mp4", tmp_path / "skier" / "cam.pt", tmp_path / "out",
                                           tmp_path / "inf", {"infer": {"gpu": 0}}, camera_head=head)
    assert out_dir == tmp_path / "out" / "single_view" / "skier"
    z = np.load(tmp_path / "inf" / "skier_multi_view_3d_info.npz")
    assert z["camera_intrinsics"].shape == (1, 4, 3, 3) and z["R"].shape == (1, 4, 3, 3) and z["C"].shape == (1, 4, 3)
    E, Kr, R, t, C, _ = _oracle_step(cfg, sd, [fr[i] for i in (0, 30, 60, 90)])
    assert np.abs(z["R"][0] - R).max() < 1e-3 and np.abs(z["t"][0] - t).max() < 1e-3 and np.abs(z["C"][0] - C).max() < 1e-3
    assert max((np.abs(z["camera_intrinsics"][0, i] - Kr[i]) / (np.abs(Kr[i]) + 1)).max() for i in range(4)) < 1e-3
    # a clip without embedded frames and without a decodable video fails with a message, not a fallback
    d = torch.load(tmp_path / "skier" / "cam.pt", weights_only=True)
    d["frames"] = None
    torch.save(d, tmp_path / "skier" / "noframes.pt")
    with pytest.raises(RuntimeError, match="embeds no frames"):
        sv.process_single_view_video(tmp_path / "skier" / "cam.mp4", tmp_path / "skier" / "noframes.pt", tmp_path / "out",
                                     tmp_path / "inf", None, camera_head=head)


@pytest.mark.parametrize("arch,causal,tta", [("3,3,3", False, True), ("3,3,3,3,3", False, True), ("3,3,3", True, False)])
def test_run_video_pose_3d(tmp_path, arch, causal, tta):
    fw = [int(v) for v in arch.split(",")]
    sd = W.make_vp3d_state_dict(seed=0, filter_widths=fw)
    torch.save({"epoch": 80, "model_pos": sd}, tmp_path / "ckpt.bin")
    T, H, Wd = 50, 1080, 1920
    kp = W.make_keypoints_2d(frames=T, seed=2)
    depth = torch.rand(T, 1, 6, 8)
    torch.save({"video_name": "run01", "video_path": "/videos/run01.mp4", "img_shape": (H, Wd),
                "detectron2": {"keypoints": kp, "bbox": torch.zeros(T, 4)}, "depth": depth}, tmp_path / "run01.pt")
    args = SimpleNamespace(architecture=arch, causal=causal, dropout=0.25, channels=1024, dense=False, test_time_augmentation=tta)
    config = {"model": {"ckpt_path": str(tmp_path / "ckpt.bin")}}
    pred, dep = vp_run.run_video_pose_3d(config, tmp_path / "run01.pt", tmp_path / "vp3d_out", args)
    saved = np.load(tmp_path / "vp3d_out" / "run01.npy")              # run.py:1089-1092: camera-space joints
    assert saved.shape == (T, 17, 3) and saved.dtype == np.float32
    ref = vp3d_oracle.lift_clip(sd, kp.numpy(), Wd, H, fw, causal, augment=tta)
    assert np.abs(saved - ref).max() < 1e-3
    world = vp_run.camera_to_world(ref.astype(np.float32), R=vp_run.CUSTOM_CAMERA_ORIENTATION, t=0)
    world[:, :, 2] -= np.min(world[:, :, 2])
    assert pred.shape == (T, 17, 3) and np.abs(pred - world).max() < 1e-3 and abs(float(pred[:, :, 2].min())) < 1e-6
    assert torch.equal(dep, depth.squeeze())
    # the .pt's keypoints are not modified on disk and the lifter refuses a clip shorter than nothing: T = 1 works (edge pad)
    torch.save({"video_name": "one", "video_path": "", "img_shape": (H, Wd), "detectron2": {"keypoints": kp[:1]}, "depth": depth[:1]},
               tmp_path / "one.pt")
    p1, _ = vp_run.run_video_pose_3d(config, tmp_path / "one.pt", tmp_path / "vp3d_out", args)
    assert p1.shape == (1, 17, 3) and np.isfinite(p1).all()


def test_config4_chain_clip_to_smoothed_joints_matches_oracle_chain(tiny):
    """BASELINE config 4 on one GPU, end to end: process_multi_view_clip = per time step one S-view VGGT call ->
    pose_enc -> cameras -> DLT over the views -> (one packed all-gather: a no-op on one rank) -> fuse.temporal_smooth_ema
    (fuse/fuse.py:329-412) over the clip, against the same chain built from the CPU oracle: oracle forward, oracle
    cameras, NumPy-SVD DLT (vggt/triangulate.py:13-71), the same EMA.  2D keypoints = projections of known points
    through the oracle's cameras (a consistent observation set)."""
    from oracle import joints_check
    from skiing_analysis_pytorch_amd import fuse

    cfg, sd, m = tiny
    T, S, H, Wd = 5, 3, 140, 140
    frames = torch.stack([W.make_images(S, H, Wd, seed=40 + t) for t in range(T)])           # [T, S, 3, H, W]
    d = cfg.to_dict()
    d["enable_point"] = d["enable_track"] = d["enable_depth"] = False
    with torch.no_grad():
        pe_ref = torch.cat([vggt_oracle.vggt_forward(sd, frames[t], d)["pose_enc"] for t in range(T)])    # [T, S, 9]
    kps, Xw, joints_ref = joints_check.keypoints_from_oracle_cameras(pe_ref, (H, Wd), joints=17, seed=3)
    out = infer.process_multi_view_clip(m, frames.cuda(), kps.cuda(), steps_per_call=2, smooth=True)
    assert out["joints3d"].shape == (T, 17, 3) and out["joints3d_smoothed"].shape == (T, 17, 3)
    scale = float(np.abs(joints_ref).max()) + 1.0
    assert np.abs(out["joints3d"].cpu().numpy() - joints_ref).max() / scale < 1e-3
    want = fuse.temporal_smooth_ema(joints_ref.astype(np.float64))
    assert np.abs(out["joints3d_smoothed"].numpy() - want).max() / scale < 1e-3
    E, K = vggt_oracle.pose_encoding_to_extri_intri(pe_ref, (H, Wd))
    assert (out["extrinsic"].cpu() - E).abs().max().item() < 1e-3
    assert ((out["intrinsic"].cpu() - K).abs() / (K.abs() + 1)).max().item() < 1e-3
    # two batches in flight (two host threads / HIP streams) give the same clip
    out2 = infer.process_multi_view_clip(m, frames.cuda(), kps.cuda(), steps_per_call=1, smooth=True, streams=2)
    assert np.abs(out2["joints3d_smoothed"].numpy() - out["joints3d_smoothed"].numpy()).max() / scale < 1e-4


def _clip_inputs(T=5, S=3, H=140, Wd=140):
    frames = torch.stack([W.make_images(S, H, Wd, seed=60 + t) for t in range(T)])
    g = torch.Generator().manual_seed(9)
    kps = torch.rand((T, S, 17, 2), generator=g) * (Wd - 40) + 20
    return frames, kps


def _rank_worker(rank, world, port, golden_path, q):
    """one rank of the two-rank clip: both ranks share cuda:0 (a one-GPU box), the collective runs over gloo"""
    import os

    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(golden_path)
        cfg = W.VGGTConfig(**json.loads(str(g["cfg_json"])))
        m = vggt.VGGT(config=cfg, prec=PREC_BF16X3, head_prec=PREC_BF16X3)
        m.load_state_dict(W.make_vggt_state_dict(cfg, seed=0))
        frames, kps = _clip_inputs()
        out = infer.process_multi_view_clip(m, frames.cuda(), kps.cuda(), steps_per_call=2, smooth=True)
        q.put((rank, out["joints3d"].cpu().numpy(), out["joints3d_smoothed"].numpy(), out["extrinsic"].cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_clip_sharded_over_two_ranks_equals_one_rank(tiny, golden_dir):
    """BASELINE config 4 across ranks, with the real model: two processes (gloo; both on this box's one GPU) shard the 5 time
    steps of a clip 3 + 2 (the second rank's block is padded by repeating the last step and cut after the gather), exchange
    joints + cameras in the one packed all-gather, smooth -- and every rank holds exactly what one process computes alone (the
    fp32-accurate mode is run-to-run deterministic, so bit for bit)."""
    import socket

    import torch.multiprocessing as mp
    cfg, sd, m = tiny
    frames, kps = _clip_inputs()
    one = infer.process_multi_view_clip(m, frames.cuda(), kps.cuda(), steps_per_call=2, smooth=True)
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, str(golden_dir / "vggt_tiny_conv.npz"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, j, js, E in res:
        assert j.shape == (5, 17, 3) and js.shape == (5, 17, 3) and E.shape == (5, 3, 3, 4)
        assert np.array_equal(j, one["joints3d"].cpu().numpy()), rank
        assert np.array_equal(js, one["joints3d_smoothed"].numpy()), rank
        assert np.array_equal(E, one["extrinsic"].cpu().numpy()), rank
