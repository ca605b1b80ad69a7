"""Counterparts of the reference's VGGT entry points, with the same signatures and return values
but none of the per-frame PNG / GLB / matplotlib I/O (out of scope: SURVEY §8):

    load_and_preprocess_images        vggt/load.py:38-183
    CameraHead                        vggt/vggt/infer.py:46-215   (run_vggt, reconstruct_from_frames, ...)
    process_multi_view_clip           the hot loop of vggt/multi_view_process.py:133-309
    process_single_view_clip          vggt/single_view_process.py:130-170

The model call is the HIP VGGT (skiing_analysis_pytorch_amd.vggt.VGGT); pose decoding,
depth unprojection and DLT triangulation also run on device (geometry.py); only the final small
arrays cross to the host.  Time steps of a clip are independent, so under torch.distributed they
are sharded across ranks and the per-step 3D joints are re-assembled with ONE all-gather
(parallel.py).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import geometry, parallel
from ._lib import PREC_BF16, PREC_BF16X3
from .vggt import VGGT
from .weights import VGGTConfig


def load_and_preprocess_images(image_list: Sequence, mode: str = "crop", device=None) -> torch.Tensor:
    """vggt/load.py:38-183.  image_list: HWC uint8 tensors / arrays.  Width -> 518 (bicubic, PIL),
    height -> round(h*518/w/14)*14, centre-crop heights > 518 ("crop") or pad to 518x518 with
    white ("pad").  Returns [N, 3, H, W] float32 in [0, 1] on the host (as the reference), or, with
    `device="cuda"`, computed on and left in GPU memory (preprocess.py: Pillow's resampler as HIP
    kernels, bit-identical to the host path)."""
    if device is not None and torch.device(device).type == "cuda":
        from .preprocess import load_and_preprocess_images_device
        return load_and_preprocess_images_device(image_list, mode, device)
    from PIL import Image

    if len(image_list) == 0:
        raise ValueError("At least 1 image is required")
    if mode not in ["crop", "pad"]:
        raise ValueError("Mode must be either 'crop' or 'pad'")
    images, shapes = [], set()
    target = 518
    for im in image_list:
        arr = im.numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
        img = Image.fromarray(arr)
        if img.mode == "RGBA":
            bg = Image.new("RGBA", img.size, (255, 255, 255, 255))
            img = Image.alpha_composite(bg, img)
        img = img.convert("RGB")
        width, height = img.size
        if mode == "pad":
            if width >= height:
                new_w = target
                new_h = round(height * (new_w / width) / 14) * 14
            else:
                new_h = target
                new_w = round(width * (new_h / height) / 14) * 14
        else:
            new_w = target
            new_h = round(height * (new_w / width) / 14) * 14
        img = img.resize((new_w, new_h), Image.Resampling.BICUBIC)
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32) / 255.0
        if mode == "crop" and new_h > target:
            y0 = (new_h - target) // 2
            t = t[:, y0:y0 + target, :]
        if mode == "pad":
            hp, wp = target - t.shape[1], target - t.shape[2]
            if hp > 0 or wp > 0:
                t = torch.nn.functional.pad(t, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2), mode="constant", value=1.0)
        shapes.add((t.shape[1], t.shape[2]))
        images.append(t)
    if len(shapes) > 1:
        mh, mw = max(s[0] for s in shapes), max(s[1] for s in shapes)
        padded = []
        for t in images:
            hp, wp = mh - t.shape[1], mw - t.shape[2]
            if hp > 0 or wp > 0:
                t = torch.nn.functional.pad(t, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2), mode="constant", value=1.0)
            padded.append(t)
        images = padded
    return torch.stack(images)


_MISSING = object()


def cfg_get(cfg, dotted: str, default=None):
    """`cfg.a.b` / `cfg["a"]["b"]` / `cfg.a.get("b", default)` of the reference's OmegaConf configs
    (configs/vggt.yaml), for any mapping- or attribute-style object (omegaconf itself is not needed)."""
    cur = cfg
    for key in dotted.split("."):
        if cur is None:
            return default
        nxt = _MISSING
        if hasattr(cur, "get"):
            try:
                nxt = cur.get(key, _MISSING)
            except TypeError:
                nxt = _MISSING
        if nxt is _MISSING:
            nxt = getattr(cur, key, _MISSING)
        if nxt is _MISSING:
            return default
        cur = nxt
    return cur


class CameraHead:
    """Drop-in for vggt.vggt.infer.CameraHead (the reference-side VGGT wrapper)."""

    def __init__(self, cfg=None, out_dir: Optional[Path] = None, model: Optional[VGGT] = None, state_dict=None,
                 prec=PREC_BF16, head_prec=PREC_BF16X3):
        gpu = int(cfg_get(cfg, "infer.gpu", 0) or 0)   # configs/vggt.yaml infer.gpu
        if not torch.cuda.is_available():
            raise RuntimeError("VGGT needs a GPU.")   # infer.py:49-51
        self.device = f"cuda:{gpu}"
        torch.cuda.set_device(gpu)
        self.outdir = Path(out_dir) if out_dir is not None else None
        # the reference reads these two at the ROOT of the config (infer.py:56-57)
        self.conf_thres = cfg_get(cfg, "conf_thres", 50.0)
        self.prediction_mode = cfg_get(cfg, "prediction_mode", "All")
        if model is None and state_dict is None:
            ckpt = cfg_get(cfg, "infer.ckpt_path", None)     # offline stand-in for the URL of infer.py:62-66
            self.vggt = self.load_vggt_model(self.device, ckpt_path=ckpt, prec=prec, head_prec=head_prec)
        else:
            self.vggt = model if model is not None else self.load_vggt_model(self.device, state_dict=state_dict, prec=prec,
                                                                            head_prec=head_prec)

    @staticmethod
    def load_vggt_model(device="cuda", verbose=True, state_dict=None, ckpt_path=None, prec=PREC_BF16,
                        head_prec=PREC_BF16X3):
        """infer.py:59-69 fetches model.pt from a URL; offline, pass the same flat state_dict
        (or a local path to it)."""
        model = VGGT(prec=prec, head_prec=head_prec)
        if state_dict is None:
            if ckpt_path is None:
                raise RuntimeError("no network here: pass state_dict= or ckpt_path= (the reference's model.pt format)")
            state_dict = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        model.load_state_dict(state_dict)
        return model.eval()

    @torch.no_grad()
    def run_vggt(self, images: List[torch.Tensor], to_numpy: bool = True):
        """infer.py:71-105.  Returns (out dict, H, W).  With to_numpy=False everything stays in HBM."""
        # RGB uint8 frames are resized on the GPU (bit-identical to the PIL path, preprocess.py);
        # anything else (RGBA, other dtypes) takes the reference's host path
        def _rgb8(im):
            return getattr(im, "dtype", None) in (torch.uint8, np.uint8) and getattr(im, "ndim", 0) == 3 and im.shape[2] == 3
        on_dev = all(_rgb8(im) for im in images)
        imgs = load_and_preprocess_images(images, device=self.device if on_dev else None).to(self.device)
        preds = self.vggt(imgs)
        H, W = imgs.shape[-2:]
        E, K = geometry.pose_encoding_to_extri_intri(preds["pose_enc"], (H, W))
        preds["extrinsic"], preds["intrinsic"] = E, K
        preds["world_points_from_depth"] = geometry.unproject_depth_map_to_point_map(
            preds["depth"][0], E[0], K[0])[None]
        if not to_numpy:
            return preds, H, W
        out = {k: (v.detach().cpu().numpy().squeeze(0) if isinstance(v, torch.Tensor) else v) for k, v in preds.items()}
        out["pose_enc_list"] = None
        return out, H, W

    extrinsic_to_RT = staticmethod(geometry.extrinsic_to_RT)
    scale_intrinsics = staticmethod(geometry.scale_intrinsics)

    def reconstruct_from_frames(self, frame_id: int, imgs: List[torch.Tensor]):
        """infer.py:157-215 -> (extrinsics [S,3,4], intrinsics rescaled to the source resolution
        (list of [3,3]), R [S,3,3], t [S,3], C [S,3], world_points_from_depth)."""
        return self.reconstruct_batch([frame_id], [imgs])[0]

    @torch.no_grad()
    def reconstruct_batch(self, frame_ids: Sequence[int], steps: Sequence[List[torch.Tensor]], write: Optional[Sequence[bool]] = None):
        """`reconstruct_from_frames` for several independent time steps in ONE model call (B = len(steps),
        every step S frames of one source size): the steps of a clip are independent
        (vggt/multi_view_process.py:133), and batching them is what fills the chip.  Returns one
        reconstruct_from_frames tuple per step; per step `<outdir>/frame_XXXX/predictions.npz` holds the
        camera arrays of the reference's predictions.npz (vggt/save.py:52-56; the dense maps are returned,
        not written: the per-frame PNG / GLB / dense dumps are out of scope)."""
        B, S = len(steps), len(steps[0])
        H, W = steps[0][0].shape[:2]
        flat = [im for st in steps for im in st]
        if len(flat) != B * S:
            raise ValueError("every time step needs the same number of views")

        def _rgb8(im):
            return getattr(im, "dtype", None) in (torch.uint8, np.uint8) and getattr(im, "ndim", 0) == 3 and im.shape[2] == 3
        on_dev = all(_rgb8(im) for im in flat)
        imgs = load_and_preprocess_images(flat, device=self.device if on_dev else None).to(self.device)
        oh, ow = imgs.shape[-2:]
        preds = self.vggt(imgs.view(B, S, 3, oh, ow), want={"camera", "depth"})
        E, K = geometry.pose_encoding_to_extri_intri(preds["pose_enc"], (oh, ow))
        wp = torch.stack([geometry.unproject_depth_map_to_point_map(preds["depth"][b], E[b], K[b]) for b in range(B)])
        En, Kn, wpn, pen = E.cpu().numpy(), K.cpu().numpy(), wp.cpu().numpy(), preds["pose_enc"].cpu().numpy()
        out = []
        for b in range(B):
            R, t, C = self.extrinsic_to_RT(En[b])
            K_resized = [self.scale_intrinsics(Kn[b, i], orig_size=(oh, ow), new_size=(H, W)) for i in range(S)]
            if self.outdir is not None and (write is None or write[b]):
                d = self.outdir / f"frame_{int(frame_ids[b]):04d}"
                d.mkdir(parents=True, exist_ok=True)
                np.savez(d / "predictions.npz", extrinsic=En[b], intrinsic=Kn[b], pose_enc=pen[b])
            out.append((En[b], K_resized, R, t, C, wpn[b]))
        return out


def save_camera_info(out_pt_path: Path, all_frame_camera_intrinsics, all_frame_R, all_frame_t, all_frame_C,
                     all_frame_x3d=None, extra: Optional[dict] = None):
    """vggt/save.py:84-110 (NPZ with camera_intrinsics [N,C,3,3], R [N,C,3,3], t [N,C,3], C [N,C,3]);
    accepts the all_frame_x3d the reference's caller passes (multi_view_process.py:312-319)."""
    data = {"camera_intrinsics": np.stack(all_frame_camera_intrinsics, axis=0), "R": np.stack(all_frame_R, axis=0),
            "t": np.stack(all_frame_t, axis=0), "C": np.stack(all_frame_C, axis=0)}
    if all_frame_x3d is not None:
        data["x3d"] = np.stack(all_frame_x3d, axis=0)
    if extra:
        data.update(extra)      # build-side additions (e.g. icp_refined, x3d_smoothed); the reference's keys are untouched
    np.savez_compressed(Path(out_pt_path).with_suffix(".npz"), **data)


_SIDE_STREAMS: Dict = {}


def _side_streams(dev, n):
    """n long-lived side streams of a device (the model keeps one workspace per stream it has run on)"""
    pool = _SIDE_STREAMS.setdefault(torch.device(dev), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(dev))
    return pool[:n]


@torch.no_grad()
def process_multi_view_clip(model: VGGT, frames: torch.Tensor, keypoints: torch.Tensor, steps_per_call: int = 4,
                            want_dense: bool = False, streams: int = 1, smooth: bool = False) -> Dict[str, torch.Tensor]:
    """The hot loop of process_multi_view_video (vggt/multi_view_process.py:133-309) for a clip
    already in memory: frames [T, S, 3, H, W] in [0,1] (device), keypoints [T, S, J, 2] in the
    pixels of the H x W frames.  Per time step: one S-view VGGT call -> cameras -> DLT
    triangulation of the J joints over the S views.

    Under torch.distributed the T time steps are split in contiguous blocks across ranks and the
    [T, J, 3] joints (+ cameras) are re-assembled on every rank with ONE all-gather of packed per-step
    records (parallel.all_gather_packed).  smooth=True chains BASELINE config 4's last stage on the gathered
    joints: `fuse.temporal_smooth_ema` (fuse/fuse.py:329-412) -> "joints3d_smoothed" [T, J, 3] float64 (host).

    streams > 1: the calls of this rank (chunks of steps_per_call time steps, independent of each
    other) are issued from that many host threads on as many HIP streams, so the HBM-bound phases of
    one call (GEMM store bursts, LayerNorm, upsamples) overlap the MFMA-bound phases of another:
    measured +5.5 % frames/s at 2 x 4 time steps in flight on one MI355X (tools/two_streams.py)."""
    T, S = frames.shape[:2]
    H, W = frames.shape[-2:]
    lo, hi, T_pad = parallel.shard_range(T)
    want = {"camera", "depth", "point"} if want_dense else {"camera"}
    starts = list(range(lo, hi, steps_per_call))

    n_par = max(1, min(int(streams), len(starts) - 1))

    def one_call(a):
        b = min(a + steps_per_call, hi)
        idx = [min(i, T - 1) for i in range(a, b)]          # padded steps repeat the last one
        n = len(idx)
        out = model(frames[idx], want=want)
        E, K = geometry.pose_encoding_to_extri_intri(out["pose_enc"], (H, W))
        R, t = E[..., :3, :3].contiguous(), E[..., :3, 3].contiguous()
        return geometry.triangulate_joints(K, R, t, keypoints[idx])[:n], E[:n], K[:n]

    results = [None] * len(starts)
    if n_par <= 1:
        for i, a in enumerate(starts):
            results[i] = one_call(a)
    else:
        import threading

        dev = frames.device
        main = torch.cuda.current_stream(dev)
        results[0] = one_call(starts[0])    # sizes this stream's workspace before the side streams start
        side = _side_streams(dev, n_par)
        errors = []

        def worker(k):
            try:
                with torch.cuda.device(dev), torch.cuda.stream(side[k]):
                    for i in range(1 + k, len(starts), n_par):
                        results[i] = one_call(starts[i])
            except BaseException as e:   # re-raised on the calling thread
                errors.append(e)

        for s in side:
            s.wait_stream(main)
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_par)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        for s in side:
            main.wait_stream(s)
        if errors:
            raise errors[0]
        for r in results:   # the side streams' results are read on the caller's stream from here on
            for x in r:
                x.record_stream(main)
    joints = torch.cat([r[0] for r in results])
    Es = torch.cat([r[1] for r in results])
    Ks = torch.cat([r[2] for r in results])
    # the path's ONE collective: joints + cameras of this rank's steps as one packed record per step
    joints, Es, Ks = parallel.all_gather_packed([joints, Es, Ks], T)
    out = {"joints3d": joints, "extrinsic": Es, "intrinsic": Ks}
    if smooth:
        # BASELINE config 4: after the gather, fuse/'s temporal smoothing over the whole clip (sequential in t,
        # O(T J) on the host as in the reference: fuse/fuse.py:329-412); every rank holds the same result
        from . import fuse
        out["joints3d_smoothed"] = torch.from_numpy(fuse.temporal_smooth_ema(joints.cpu().numpy().astype(np.float64)))
    return out


@torch.no_grad()
def process_single_view_clip(model: VGGT, frames: torch.Tensor, every: int = 30):
    """vggt/single_view_process.py:130-170: every `every`-th frame of ONE camera forms a single
    S = ceil(T/every) call; returns the cameras of those frames.  frames [T, 3, H, W] (device)."""
    sel = frames[::every]
    H, W = sel.shape[-2:]
    out = model(sel, want={"camera"})
    E, K = geometry.pose_encoding_to_extri_intri(out["pose_enc"], (H, W))
    return {"extrinsic": E[0], "intrinsic": K[0], "pose_enc": out["pose_enc"][0]}
