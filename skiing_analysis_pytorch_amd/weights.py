"""Deterministic synthetic weights in the reference's own state_dict formats.

The reference obtains its weights from URLs / files that are not available offline
(vggt/vggt/infer.py:63-66, configs/videopose3d.yaml:18), so parity and benchmarks run on
synthetic weights.  Each tensor is drawn from a torch CPU generator seeded by
crc32(key) ^ seed, so the same dict can be rebuilt anywhere from (spec, seed) alone and the
golden fixtures need to store no weights.  Magnitudes are "trained-like" (O(1) activations,
LayerScale ~0.1-0.3) so that every branch of the path contributes to the outputs.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _normal(key, seed, shape, std, mean=0.0):
    return torch.randn(shape, generator=_gen(key, seed), dtype=torch.float32) * std + mean


def _uniform(key, seed, shape, lo, hi):
    return torch.rand(shape, generator=_gen(key, seed), dtype=torch.float32) * (hi - lo) + lo


# --------------------------------------------------------------------------------------
# VideoPose3D TemporalModel (keys: VideoPose3D/common/model.py:32-33,103,112-122)
# --------------------------------------------------------------------------------------
def vp3d_spec(joints_in=17, in_features=2, joints_out=17, filter_widths=(3, 3, 3), channels=1024):
    spec = OrderedDict()
    cin = joints_in * in_features

    def bn(prefix):
        spec[prefix + ".weight"] = ((channels,), "bn_w")
        spec[prefix + ".bias"] = ((channels,), "bn_b")
        spec[prefix + ".running_mean"] = ((channels,), "bn_m")
        spec[prefix + ".running_var"] = ((channels,), "bn_v")
        spec[prefix + ".num_batches_tracked"] = ((), "count")

    bn("expand_bn")
    spec["shrink.weight"] = ((joints_out * 3, channels, 1), "conv")
    spec["shrink.bias"] = ((joints_out * 3,), "bias")
    spec["expand_conv.weight"] = ((channels, cin, filter_widths[0]), "conv")
    for i in range(1, len(filter_widths)):
        spec[f"layers_conv.{2 * (i - 1)}.weight"] = ((channels, channels, filter_widths[i]), "conv")
        spec[f"layers_conv.{2 * (i - 1) + 1}.weight"] = ((channels, channels, 1), "conv")
    for i in range(1, len(filter_widths)):
        bn(f"layers_bn.{2 * (i - 1)}")
        bn(f"layers_bn.{2 * (i - 1) + 1}")
    return spec


def make_vp3d_state_dict(seed=0, **kw):
    sd = OrderedDict()
    for key, (shape, kind) in vp3d_spec(**kw).items():
        if kind == "conv":
            fan_in = shape[1] * shape[2]
            # He-like scale keeps the ReLU stack at O(1)
            sd[key] = _normal(key, seed, shape, math.sqrt(2.0 / fan_in))
        elif kind == "bias":
            sd[key] = _normal(key, seed, shape, 0.1)
        elif kind == "bn_w":
            sd[key] = _uniform(key, seed, shape, 0.5, 1.5)
        elif kind == "bn_b":
            sd[key] = _normal(key, seed, shape, 0.1)
        elif kind == "bn_m":
            sd[key] = _normal(key, seed, shape, 0.1)
        elif kind == "bn_v":
            sd[key] = _uniform(key, seed, shape, 0.5, 1.5)
        elif kind == "count":
            sd[key] = torch.tensor(0, dtype=torch.long)
        else:  # pragma: no cover
            raise ValueError(kind)
    return sd


def make_keypoints_2d(frames=243, joints=17, w=1920, h=1080, seed=0):
    """Synthetic pixel-space 2D keypoints [T, J, 2]: a smooth random walk inside the frame."""
    g = _gen("keypoints_2d", seed)
    base = torch.rand((1, joints, 2), generator=g) * torch.tensor([w * 0.5, h * 0.5]) + torch.tensor([w * 0.25, h * 0.25])
    steps = torch.randn((frames, joints, 2), generator=g) * 3.0
    return (base + torch.cumsum(steps, dim=0)).to(torch.float32)


# --------------------------------------------------------------------------------------
# VGGT (keys: vggt/vggt/models/{vggt,aggregator}.py, layers/*, heads/*)
# --------------------------------------------------------------------------------------
class VGGTConfig:
    """Shape parameters of the VGGT family.  Defaults = the reference's VGGT() (VGGT-1B:
    vggt/vggt/models/vggt.py:18-27, aggregator.py:51-70); small values build the tiny test models."""

    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                 num_register_tokens=4, patch_embed="dinov2_vitl14_reg", dino_depth=24, dino_heads=16,
                 cam_trunk_depth=4, cam_heads=16, dpt_features=256, dpt_out_channels=(256, 512, 1024, 1024),
                 dpt_layers=(4, 11, 17, 23), track_features=128, track_hidden=384, track_corr_levels=7,
                 track_corr_radius=4, track_iters=4, track_depth=6, track_heads=8, track_virtual=64,
                 enable_camera=True, enable_depth=True, enable_point=True, enable_track=True):
        self.__dict__.update({k: v for k, v in locals().items() if k != "self"})
        self.dpt_out_channels = tuple(dpt_out_channels)
        self.dpt_layers = tuple(dpt_layers)

    @property
    def use_dino(self):
        return "conv" not in self.patch_embed

    def to_dict(self):
        return dict(self.__dict__)


def _block_spec(spec, prefix, C, hidden, qk_norm, head_dim):
    spec[prefix + ".norm1.weight"] = ((C,), "ln_w")
    spec[prefix + ".norm1.bias"] = ((C,), "ln_b")
    spec[prefix + ".attn.qkv.weight"] = ((3 * C, C), "linear")
    spec[prefix + ".attn.qkv.bias"] = ((3 * C,), "bias")
    if qk_norm:
        spec[prefix + ".attn.q_norm.weight"] = ((head_dim,), "ln_w")
        spec[prefix + ".attn.q_norm.bias"] = ((head_dim,), "ln_b")
        spec[prefix + ".attn.k_norm.weight"] = ((head_dim,), "ln_w")
        spec[prefix + ".attn.k_norm.bias"] = ((head_dim,), "ln_b")
    spec[prefix + ".attn.proj.weight"] = ((C, C), "linear")
    spec[prefix + ".attn.proj.bias"] = ((C,), "bias")
    spec[prefix + ".ls1.gamma"] = ((C,), "ls")
    spec[prefix + ".norm2.weight"] = ((C,), "ln_w")
    spec[prefix + ".norm2.bias"] = ((C,), "ln_b")
    spec[prefix + ".mlp.fc1.weight"] = ((hidden, C), "linear")
    spec[prefix + ".mlp.fc1.bias"] = ((hidden,), "bias")
    spec[prefix + ".mlp.fc2.weight"] = ((C, hidden), "linear")
    spec[prefix + ".mlp.fc2.bias"] = ((C,), "bias")
    spec[prefix + ".ls2.gamma"] = ((C,), "ls")


def _dpt_spec(spec, prefix, dim_in, features, out_channels, output_dim, feature_only):
    spec[prefix + ".norm.weight"] = ((dim_in,), "ln_w")
    spec[prefix + ".norm.bias"] = ((dim_in,), "ln_b")
    for i, oc in enumerate(out_channels):
        spec[f"{prefix}.projects.{i}.weight"] = ((oc, dim_in, 1, 1), "conv")
        spec[f"{prefix}.projects.{i}.bias"] = ((oc,), "bias")
    oc = out_channels
    spec[prefix + ".resize_layers.0.weight"] = ((oc[0], oc[0], 4, 4), "convT")
    spec[prefix + ".resize_layers.0.bias"] = ((oc[0],), "bias")
    spec[prefix + ".resize_layers.1.weight"] = ((oc[1], oc[1], 2, 2), "convT")
    spec[prefix + ".resize_layers.1.bias"] = ((oc[1],), "bias")
    spec[prefix + ".resize_layers.3.weight"] = ((oc[3], oc[3], 3, 3), "conv")
    spec[prefix + ".resize_layers.3.bias"] = ((oc[3],), "bias")
    for i in range(4):
        spec[f"{prefix}.scratch.layer{i + 1}_rn.weight"] = ((features, oc[i], 3, 3), "conv")
    for r in (1, 2, 3, 4):
        p = f"{prefix}.scratch.refinenet{r}"
        spec[p + ".out_conv.weight"] = ((features, features, 1, 1), "conv")
        spec[p + ".out_conv.bias"] = ((features,), "bias")
        for u in ((1, 2) if r != 4 else (2,)):
            for c in (1, 2):
                spec[f"{p}.resConfUnit{u}.conv{c}.weight"] = ((features, features, 3, 3), "conv")
                spec[f"{p}.resConfUnit{u}.conv{c}.bias"] = ((features,), "bias")
    if feature_only:
        spec[prefix + ".scratch.output_conv1.weight"] = ((features, features, 3, 3), "conv")
        spec[prefix + ".scratch.output_conv1.bias"] = ((features,), "bias")
    else:
        spec[prefix + ".scratch.output_conv1.weight"] = ((features // 2, features, 3, 3), "conv")
        spec[prefix + ".scratch.output_conv1.bias"] = ((features // 2,), "bias")
        spec[prefix + ".scratch.output_conv2.0.weight"] = ((32, features // 2, 3, 3), "conv")
        spec[prefix + ".scratch.output_conv2.0.bias"] = ((32,), "bias")
        spec[prefix + ".scratch.output_conv2.2.weight"] = ((output_dim, 32, 1, 1), "conv")
        spec[prefix + ".scratch.output_conv2.2.bias"] = ((output_dim,), "head_bias")


def _mha_block_spec(spec, prefix, H, cross):
    spec[prefix + ".norm1.weight"] = ((H,), "ln_w")
    spec[prefix + ".norm1.bias"] = ((H,), "ln_b")
    if cross:
        spec[prefix + ".norm_context.weight"] = ((H,), "ln_w")
        spec[prefix + ".norm_context.bias"] = ((H,), "ln_b")
    spec[prefix + ".norm2.weight"] = ((H,), "ln_w")
    spec[prefix + ".norm2.bias"] = ((H,), "ln_b")
    a = prefix + (".cross_attn" if cross else ".attn")
    spec[a + ".in_proj_weight"] = ((3 * H, H), "linear")
    spec[a + ".in_proj_bias"] = ((3 * H,), "bias")
    spec[a + ".out_proj.weight"] = ((H, H), "linear")
    spec[a + ".out_proj.bias"] = ((H,), "bias")
    spec[prefix + ".mlp.fc1.weight"] = ((4 * H, H), "linear")
    spec[prefix + ".mlp.fc1.bias"] = ((4 * H,), "bias")
    spec[prefix + ".mlp.fc2.weight"] = ((H, 4 * H), "linear")
    spec[prefix + ".mlp.fc2.bias"] = ((H,), "bias")


def vggt_spec(cfg: VGGTConfig):
    """Ordered {key: (shape, kind)} of the reference VGGT state_dict for `cfg`."""
    spec = OrderedDict()
    C = cfg.embed_dim
    hd = C // cfg.num_heads
    hidden = int(C * cfg.mlp_ratio)
    npatch = (cfg.img_size // cfg.patch_size) ** 2
    A = "aggregator"
    spec[A + ".camera_token"] = ((1, 2, 1, C), "token")
    spec[A + ".register_token"] = ((1, 2, cfg.num_register_tokens, C), "token")
    if cfg.use_dino:
        pe = A + ".patch_embed"
        spec[pe + ".cls_token"] = ((1, 1, C), "token")
        spec[pe + ".pos_embed"] = ((1, npatch + 1, C), "pos")
        spec[pe + ".register_tokens"] = ((1, cfg.num_register_tokens, C), "token")
        spec[pe + ".mask_token"] = ((1, C), "token")
        spec[pe + ".patch_embed.proj.weight"] = ((C, 3, cfg.patch_size, cfg.patch_size), "conv")
        spec[pe + ".patch_embed.proj.bias"] = ((C,), "bias")
        for i in range(cfg.dino_depth):
            _block_spec(spec, f"{pe}.blocks.{i}", C, hidden, False, C // cfg.dino_heads)
        spec[pe + ".norm.weight"] = ((C,), "ln_w")
        spec[pe + ".norm.bias"] = ((C,), "ln_b")
    else:
        spec[A + ".patch_embed.proj.weight"] = ((C, 3, cfg.patch_size, cfg.patch_size), "conv")
        spec[A + ".patch_embed.proj.bias"] = ((C,), "bias")
    for i in range(cfg.depth):
        _block_spec(spec, f"{A}.frame_blocks.{i}", C, hidden, True, hd)
    for i in range(cfg.depth):
        _block_spec(spec, f"{A}.global_blocks.{i}", C, hidden, True, hd)
    D = 2 * C
    if cfg.enable_camera:
        H = "camera_head"
        for i in range(cfg.cam_trunk_depth):
            _block_spec(spec, f"{H}.trunk.{i}", D, 4 * D, False, D // cfg.cam_heads)
        spec[H + ".token_norm.weight"] = ((D,), "ln_w")
        spec[H + ".token_norm.bias"] = ((D,), "ln_b")
        spec[H + ".trunk_norm.weight"] = ((D,), "ln_w")
        spec[H + ".trunk_norm.bias"] = ((D,), "ln_b")
        spec[H + ".empty_pose_tokens"] = ((1, 1, 9), "pose_token")
        spec[H + ".embed_pose.weight"] = ((D, 9), "linear")
        spec[H + ".embed_pose.bias"] = ((D,), "bias")
        spec[H + ".poseLN_modulation.1.weight"] = ((3 * D, D), "linear")
        spec[H + ".poseLN_modulation.1.bias"] = ((3 * D,), "bias")
        spec[H + ".pose_branch.fc1.weight"] = ((D // 2, D), "linear")
        spec[H + ".pose_branch.fc1.bias"] = ((D // 2,), "bias")
        spec[H + ".pose_branch.fc2.weight"] = ((9, D // 2), "linear")
        spec[H + ".pose_branch.fc2.bias"] = ((9,), "pose_bias")
    if cfg.enable_point:
        _dpt_spec(spec, "point_head", D, cfg.dpt_features, cfg.dpt_out_channels, 4, False)
    if cfg.enable_depth:
        _dpt_spec(spec, "depth_head", D, cfg.dpt_features, cfg.dpt_out_channels, 2, False)
    if cfg.enable_track:
        T = "track_head"
        _dpt_spec(spec, T + ".feature_extractor", D, cfg.track_features, cfg.dpt_out_channels, 4, True)
        K = T + ".tracker"
        L, Hh = cfg.track_features, cfg.track_hidden
        tdim = 3 * L + 4
        spec[K + ".query_ref_token"] = ((1, 2, tdim), "token1")
        corr_in = cfg.track_corr_levels * (2 * cfg.track_corr_radius + 1) ** 2
        spec[K + ".corr_mlp.fc1.weight"] = ((Hh, corr_in), "linear")
        spec[K + ".corr_mlp.fc1.bias"] = ((Hh,), "bias")
        spec[K + ".corr_mlp.fc2.weight"] = ((L, Hh), "linear")
        spec[K + ".corr_mlp.fc2.bias"] = ((L,), "bias")
        U = K + ".updateformer"
        spec[U + ".virual_tracks"] = ((1, cfg.track_virtual, 1, Hh), "token1")
        spec[U + ".input_norm.weight"] = ((tdim,), "ln_w")
        spec[U + ".input_norm.bias"] = ((tdim,), "ln_b")
        spec[U + ".input_transform.weight"] = ((Hh, tdim), "linear")
        spec[U + ".input_transform.bias"] = ((Hh,), "bias")
        spec[U + ".output_norm.weight"] = ((Hh,), "ln_w")
        spec[U + ".output_norm.bias"] = ((Hh,), "ln_b")
        spec[U + ".flow_head.weight"] = ((L + 2, Hh), "flow")
        spec[U + ".flow_head.bias"] = ((L + 2,), "flow_bias")
        for i in range(cfg.track_depth):
            _mha_block_spec(spec, f"{U}.time_blocks.{i}", Hh, False)
        for i in range(cfg.track_depth):
            _mha_block_spec(spec, f"{U}.space_virtual_blocks.{i}", Hh, False)
        for i in range(cfg.track_depth):
            _mha_block_spec(spec, f"{U}.space_point2virtual_blocks.{i}", Hh, True)
        for i in range(cfg.track_depth):
            _mha_block_spec(spec, f"{U}.space_virtual2point_blocks.{i}", Hh, True)
        spec[K + ".fmap_norm.weight"] = ((L,), "ln_w")
        spec[K + ".fmap_norm.bias"] = ((L,), "ln_b")
        spec[K + ".ffeat_norm.weight"] = ((L,), "ln_w")
        spec[K + ".ffeat_norm.bias"] = ((L,), "ln_b")
        spec[K + ".ffeat_updater.0.weight"] = ((L, L), "linear")
        spec[K + ".ffeat_updater.0.bias"] = ((L,), "bias")
        spec[K + ".vis_predictor.0.weight"] = ((1, L), "linear")
        spec[K + ".vis_predictor.0.bias"] = ((1,), "bias")
        spec[K + ".conf_predictor.0.weight"] = ((1, L), "linear")
        spec[K + ".conf_predictor.0.bias"] = ((1,), "bias")
    return spec


def _fill(key, seed, shape, kind, device=None):
    """One synthetic tensor.  device=None: deterministic CPU generator (parity);
    device='cuda': fast on-device fill of the same distribution (benchmarks only)."""
    if device is not None:
        g = torch.Generator(device=device)
        g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        rn = lambda std: torch.randn(shape, generator=g, dtype=torch.float32, device=device) * std  # noqa: E731
        ru = lambda lo, hi: torch.rand(shape, generator=g, dtype=torch.float32, device=device) * (hi - lo) + lo  # noqa: E731
    else:
        rn = lambda std: _normal(key, seed, shape, std)  # noqa: E731
        ru = lambda lo, hi: _uniform(key, seed, shape, lo, hi)  # noqa: E731
    if kind == "linear":
        return rn(0.64 / math.sqrt(shape[1]))
    if kind in ("conv", "convT"):
        fan_in = shape[1] * shape[2] * shape[3] if kind == "conv" else shape[0]
        return rn(0.9 / math.sqrt(fan_in))
    if kind == "bias":
        return rn(0.02)
    if kind == "ln_w":
        return 1.0 + rn(0.1)
    if kind == "ln_b":
        return rn(0.05)
    if kind == "ls":
        return ru(0.05, 0.3)
    if kind == "token":
        return rn(0.5)
    if kind == "token1":
        return rn(1.0)
    if kind == "pos":
        return rn(0.2)
    if kind == "pose_token":
        return rn(0.1)
    if kind == "pose_bias":
        # keep the FoV outputs (ReLU'd, heads/camera_head.py:135-137) comfortably positive
        t = rn(0.02)
        t[7:] = t[7:] + 1.0
        return t
    if kind == "head_bias":
        return rn(0.05)
    if kind == "flow":
        return rn(0.02 / math.sqrt(shape[1]) * 8)
    if kind == "flow_bias":
        return rn(0.01)
    raise ValueError(kind)


def make_vggt_state_dict(cfg: VGGTConfig, seed=0, device=None, prefix_filter=None):
    sd = OrderedDict()
    for key, (shape, kind) in vggt_spec(cfg).items():
        if prefix_filter is not None and not key.startswith(prefix_filter):
            continue
        sd[key] = _fill(key, seed, shape, kind, device)
    return sd


def make_images(S, H, W, seed=0):
    """[S, 3, H, W] in [0, 1): smooth-ish synthetic frames (low-frequency pattern + noise)."""
    g = _gen("images", seed)
    base = torch.rand((S, 3, H // 14 + 1, W // 14 + 1), generator=g)
    img = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    img = 0.7 * img + 0.3 * torch.rand((S, 3, H, W), generator=g)
    return img.clamp(0, 1).to(torch.float32)
