"""Host-side mirror of the VGGT boundary.

`VGGT` has the call signature and output dict of the reference module
(vggt/vggt/models/vggt.py:29-96) and loads the reference's flat state_dict
(vggt/vggt/infer.py:62-67), so the call site `preds = self.vggt(imgs)` (infer.py:84) works
unchanged.  The forward is `skimi_vggt_forward` in libskimi.so; PyTorch only supplies device
memory for the inputs, outputs and the workspace, and the stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import PREC_BF16, PREC_BF16X3, check, lib, ptr
from .weights import VGGTConfig, vggt_spec


class VGGTConfigC(C.Structure):
    """Mirror of `skimi_vggt_config` (include/skimi.h)."""

    _fields_ = [
        ("patch_size", C.c_int32), ("embed_dim", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32),
        ("num_register_tokens", C.c_int32),
        ("use_dino", C.c_int32), ("dino_depth", C.c_int32), ("dino_heads", C.c_int32), ("dino_img_size", C.c_int32),
        ("cam_trunk_depth", C.c_int32), ("cam_heads", C.c_int32), ("cam_iters", C.c_int32),
        ("dpt_features", C.c_int32), ("dpt_out_channels", C.c_int32 * 4), ("dpt_layers", C.c_int32 * 4),
        ("track_features", C.c_int32), ("track_hidden", C.c_int32), ("track_corr_levels", C.c_int32),
        ("track_corr_radius", C.c_int32), ("track_iters", C.c_int32), ("track_depth", C.c_int32),
        ("track_heads", C.c_int32), ("track_virtual", C.c_int32),
        ("enable_camera", C.c_int32), ("enable_depth", C.c_int32), ("enable_point", C.c_int32),
        ("enable_track", C.c_int32),
        ("prec", C.c_int32), ("head_prec", C.c_int32),
    ]


class VGGTOutputsC(C.Structure):
    """Mirror of `skimi_vggt_outputs`."""

    _fields_ = [(n, C.c_void_p) for n in ("pose_enc", "pose_enc_list", "depth", "depth_conf", "world_points",
                                          "world_points_conf", "track", "vis", "conf", "tokens_last")]


class VGGT:
    """Drop-in for vggt.vggt.models.vggt.VGGT (inference only).

    prec: MFMA mode of the DINOv2 + aggregator blocks (PREC_BF16 = the reference's autocast mode; PREC_F16 = fp16
    operands in the Linears at the same matrix rate, the cheapest mode whose 3D joints stay within 1e-3 of the fp32
    CPU path; PREC_BF16X3 = fp32-accurate; PREC_FP8 = MXFP8 qkv / proj / fc1 / fc2).  head_prec: mode of the camera/DPT heads, which the reference
    runs in fp32 (vggt.py:65)."""

    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, enable_camera=True, enable_point=True,
                 enable_depth=True, enable_track=True, prec=PREC_BF16, head_prec=PREC_BF16X3, config: VGGTConfig = None,
                 cam_iters=4):
        self.cfg = config or VGGTConfig(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim,
                                        enable_camera=enable_camera, enable_point=enable_point,
                                        enable_depth=enable_depth, enable_track=enable_track)
        self.prec, self.head_prec = prec, head_prec
        self.cam_iters = cam_iters
        cfg = self.cfg
        cc = VGGTConfigC()
        cc.patch_size, cc.embed_dim, cc.depth, cc.num_heads = cfg.patch_size, cfg.embed_dim, cfg.depth, cfg.num_heads
        cc.num_register_tokens = cfg.num_register_tokens
        cc.use_dino, cc.dino_depth, cc.dino_heads, cc.dino_img_size = int(cfg.use_dino), cfg.dino_depth, cfg.dino_heads, cfg.img_size
        cc.cam_trunk_depth, cc.cam_heads, cc.cam_iters = cfg.cam_trunk_depth, cfg.cam_heads, cam_iters
        cc.dpt_features = cfg.dpt_features
        for i in range(4):
            cc.dpt_out_channels[i] = cfg.dpt_out_channels[i]
            cc.dpt_layers[i] = cfg.dpt_layers[i]
        cc.track_features, cc.track_hidden = cfg.track_features, cfg.track_hidden
        cc.track_corr_levels, cc.track_corr_radius = cfg.track_corr_levels, cfg.track_corr_radius
        cc.track_iters, cc.track_depth, cc.track_heads, cc.track_virtual = cfg.track_iters, cfg.track_depth, cfg.track_heads, cfg.track_virtual
        cc.enable_camera, cc.enable_depth = int(cfg.enable_camera), int(cfg.enable_depth)
        cc.enable_point, cc.enable_track = int(cfg.enable_point), int(cfg.enable_track)
        cc.prec, cc.head_prec = prec, head_prec
        self._h = lib().skimi_vggt_create(C.byref(cc))
        if not self._h:
            raise _lib.SkimiError(lib().skimi_last_error().decode())
        self._ws = {}   # (device, stream) -> workspace
        self.training = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:   # module globals may already be torn down at interpreter exit
            try:
                lib().skimi_vggt_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- reference API --------------------------------------------------------------
    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def cuda(self):
        return self

    def load_state_dict(self, state_dict, strict=True):
        spec = vggt_spec(self.cfg)
        missing = [k for k in spec if k not in state_dict]
        unexpected = [k for k in state_dict if k not in spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        for k, (shape, _kind) in spec.items():
            if k not in state_dict:
                continue
            t = state_dict[k].detach()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(shape)}")
            t = t.to(torch.float32).contiguous()
            check(lib().skimi_vggt_set_weight(self._h, k.encode(), t.data_ptr(), t.numel(), int(t.is_cuda)),
                  f"set_weight({k})")
        check(lib().skimi_vggt_finalize(self._h), "skimi_vggt_finalize")
        if self.cfg.use_dino:   # kept on the host for interpolate_pos_encoding at other input sizes
            self._pos_embed = state_dict["aggregator.patch_embed.pos_embed"].detach().to("cpu", torch.float32)
            self._pos_sizes = set()
        return self

    def _ensure_pos_embed(self, H: int, W: int):
        """vision_transformer.py:180-212 (interpolate_pos_encoding, antialias=True, offset 0):
        a per-resolution constant table, computed once on the host and registered with the
        library.  (The reference names the height `w` and the width `h`; size=(w0, h0) is
        (rows, cols).)"""
        cfg = self.cfg
        if not cfg.use_dino or (H == W == cfg.img_size) or (H, W) in self._pos_sizes:
            return
        pe = self._pos_embed
        N = pe.shape[1] - 1
        M = int(N ** 0.5)
        assert N == M * M
        dim = pe.shape[-1]
        w0, h0 = H // cfg.patch_size, W // cfg.patch_size
        grid = torch.nn.functional.interpolate(pe[:, 1:].reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic",
                                               antialias=True, size=(w0, h0))
        table = torch.cat((pe[:, 0], grid.permute(0, 2, 3, 1).reshape(-1, dim)), dim=0).contiguous()
        check(lib().skimi_vggt_set_pos_embed(self._h, H, W, table.data_ptr(), 0), "skimi_vggt_set_pos_embed")
        self._pos_sizes.add((H, W))

    def rope_positions(self, frames: int, H: int, W: int, device="cuda") -> torch.Tensor:
        """The RoPE position table the forward uses: int32 [frames, P, 2] (y, x) -- PositionGetter + the special-token
        offset (rope.py:39-59, aggregator.py:219-228).  An index path, exposed for the bit-exact parity test."""
        self._ensure_pos_embed(H, W)
        P = 1 + self.cfg.num_register_tokens + (H // self.cfg.patch_size) * (W // self.cfg.patch_size)
        out = torch.empty((frames, P, 2), dtype=torch.int32, device=device)
        check(lib().skimi_vggt_rope_positions(self._h, frames, H, W, out.data_ptr()), "skimi_vggt_rope_positions")
        return out

    def __call__(self, images, query_points=None, **kw):
        return self.forward(images, query_points, **kw)

    def forward(self, images: torch.Tensor, query_points: torch.Tensor = None, want=None, return_tokens=False):
        """images [S,3,H,W] or [B,S,3,H,W] in [0,1]; query_points [N,2] or [B,N,2] pixels.
        `want` optionally restricts which heads run (subset of {"camera","depth","point","track"})."""
        if not images.is_cuda:
            raise _lib.SkimiError("VGGT.forward needs a device tensor (the HIP path is the only path)")
        if images.dim() == 4:   # vggt.py:55-56
            images = images.unsqueeze(0)
        if query_points is not None and query_points.dim() == 2:
            query_points = query_points.unsqueeze(0)
        B, S, Cin, H, W = images.shape
        if Cin != 3:
            raise ValueError(f"Expected 3 input channels, got {Cin}")   # aggregator.py:197-198
        images = images.contiguous().to(torch.float32)
        cfg = self.cfg
        dev = images.device
        want = set(want) if want is not None else {"camera", "depth", "point", "track"}
        f32 = dict(dtype=torch.float32, device=dev)
        outs = VGGTOutputsC()
        preds = {}
        if cfg.enable_camera and "camera" in want:
            lst = torch.empty((self.cam_iters, B, S, 9), **f32)
            pe = torch.empty((B, S, 9), **f32)
            outs.pose_enc_list, outs.pose_enc = ptr(lst), ptr(pe)
            preds["pose_enc"] = pe
            preds["pose_enc_list"] = [lst[i] for i in range(self.cam_iters)]
        if cfg.enable_depth and "depth" in want:
            preds["depth"] = torch.empty((B, S, H, W, 1), **f32)
            preds["depth_conf"] = torch.empty((B, S, H, W), **f32)
            outs.depth, outs.depth_conf = ptr(preds["depth"]), ptr(preds["depth_conf"])
        if cfg.enable_point and "point" in want:
            preds["world_points"] = torch.empty((B, S, H, W, 3), **f32)
            preds["world_points_conf"] = torch.empty((B, S, H, W), **f32)
            outs.world_points, outs.world_points_conf = ptr(preds["world_points"]), ptr(preds["world_points_conf"])
        nq = 0
        if cfg.enable_track and query_points is not None and "track" in want:
            query_points = query_points.contiguous().to(torch.float32)
            nq = query_points.shape[1]
            preds["track"] = torch.empty((B, S, nq, 2), **f32)
            preds["vis"] = torch.empty((B, S, nq), **f32)
            preds["conf"] = torch.empty((B, S, nq), **f32)
            outs.track, outs.vis, outs.conf = ptr(preds["track"]), ptr(preds["vis"]), ptr(preds["conf"])
        if return_tokens:
            P = 1 + cfg.num_register_tokens + (H // cfg.patch_size) * (W // cfg.patch_size)
            preds["tokens_last"] = torch.empty((B, S, P, 2 * cfg.embed_dim), **f32)
            outs.tokens_last = ptr(preds["tokens_last"])
        self._ensure_pos_embed(H, W)
        need = lib().skimi_vggt_workspace_bytes(self._h, B, S, H, W, nq)
        if need == 0:
            raise _lib.SkimiError(lib().skimi_last_error().decode())
        # one workspace per stream: forwards issued from different host threads on different streams
        # (infer.process_multi_view_clip(streams=2)) run concurrently on the device; the handle itself is
        # read-only once a frame shape has been prepared by a first call
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            self._ws.pop(key, None)
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        check(lib().skimi_vggt_forward(self._h, ptr(images), ptr(query_points) if nq else None, B, S, H, W, nq,
                                       C.byref(outs), ptr(ws), ws.numel(), _lib.current_stream()),
              "skimi_vggt_forward")
        if not self.training:
            preds["images"] = images   # vggt.py:93-94
        return preds
