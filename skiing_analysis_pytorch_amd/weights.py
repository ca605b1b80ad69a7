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
