"""Oracle: VGGT forward (aggregator, camera / DPT / track heads) and the geometry post-proc.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A functional fp32 CPU restatement that
walks a flat reference-format state_dict; no nn.Module tree.  Each function cites the
reference lines it follows.  Pinned by tests/golden/vggt_*.npz, which hold outputs of the
reference's own classes (tools/make_goldens.py).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

_RESNET_MEAN = [0.485, 0.456, 0.406]   # vggt/vggt/models/aggregator.py:21-22
_RESNET_STD = [0.229, 0.224, 0.225]


def _ln(x, sd, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


# ---------------------------------------------------------------------------------------
# RoPE  (vggt/vggt/layers/rope.py)
# ---------------------------------------------------------------------------------------
def rope_tables(npos, dim=32, base=100.0):
    """rope.py:86-117: inv_freq = 1/base^(arange(0,dim,2)/dim); angles = pos x inv_freq -> [npos, dim/2]"""
    exponents = torch.arange(0, dim, 2).float() / dim
    inv_freq = 1.0 / (base ** exponents)
    positions = torch.arange(npos, dtype=inv_freq.dtype)
    angles = torch.einsum("i,j->ij", positions, inv_freq)
    return angles.cos(), angles.sin()


def positions_2d(n_frames, ph, pw, n_special):
    """rope.py:39-59 (cartesian_prod of (y, x)) + aggregator.py:219-228 (+1, zeros for the special
    tokens).  int64 [n_frames, n_special + ph*pw, 2] — index path, bit-exact."""
    pos = torch.cartesian_prod(torch.arange(ph), torch.arange(pw)).view(1, ph * pw, 2).expand(n_frames, -1, -1).clone()
    if n_special > 0:
        pos = pos + 1
        pos = torch.cat([torch.zeros(n_frames, n_special, 2, dtype=pos.dtype), pos], dim=1)
    return pos


def rope2d(t, pos, base=100.0):
    """rope.py:154-188.  t [B, H, N, D]; pos [B, N, 2] int64."""
    D = t.shape[-1] // 2
    cos_t, sin_t = rope_tables(int(pos.max()) + 1, D, base)
    cos_t = torch.cat((cos_t, cos_t), -1)   # rope.py:112
    sin_t = torch.cat((sin_t, sin_t), -1)

    def rot(x):  # rope.py:119-131
        h = x.shape[-1] // 2
        return torch.cat((-x[..., h:], x[..., :h]), dim=-1)

    def one(x, p):  # rope.py:133-152
        c = F.embedding(p, cos_t)[:, None]
        s = F.embedding(p, sin_t)[:, None]
        return x * c + rot(x) * s

    v, h = t.chunk(2, dim=-1)
    return torch.cat((one(v, pos[..., 0]), one(h, pos[..., 1])), dim=-1)


# ---------------------------------------------------------------------------------------
# Transformer block  (vggt/vggt/layers/{attention,block,mlp,layer_scale}.py)
# ---------------------------------------------------------------------------------------
def attention(sd, prefix, x, num_heads, pos=None, qk_norm=False):
    """attention.py:50-72"""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = _lin(x, sd, prefix + ".qkv").reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    if qk_norm:
        q, k = _ln(q, sd, prefix + ".q_norm"), _ln(k, sd, prefix + ".k_norm")
    if pos is not None:
        q, k = rope2d(q, pos), rope2d(k, pos)
    o = F.scaled_dot_product_attention(q, k, v)
    return _lin(o.transpose(1, 2).reshape(B, N, C), sd, prefix + ".proj")


def block(sd, prefix, x, num_heads, pos=None, qk_norm=False, eps=1e-5):
    """block.py:77-98 (eval branch): x += ls1(attn(norm1(x))); x += ls2(mlp(norm2(x)))"""
    a = attention(sd, prefix + ".attn", _ln(x, sd, prefix + ".norm1", eps), num_heads, pos, qk_norm)
    x = x + a * sd[prefix + ".ls1.gamma"]
    h = _lin(F.gelu(_lin(_ln(x, sd, prefix + ".norm2", eps), sd, prefix + ".mlp.fc1")), sd, prefix + ".mlp.fc2")
    return x + h * sd[prefix + ".ls2.gamma"]


# ---------------------------------------------------------------------------------------
# Aggregator  (vggt/vggt/models/aggregator.py, layers/vision_transformer.py, layers/patch_embed.py)
# ---------------------------------------------------------------------------------------
def interpolate_pos_encoding(pos_embed, npatch, H, W, patch):
    """vision_transformer.py:180-212 with interpolate_antialias=True, interpolate_offset=0.0
    (aggregator.py:147-148): identity for the native square size, otherwise bicubic + antialias
    resize of the M x M grid to (H/patch, W/patch); the class position is kept."""
    N = pos_embed.shape[1] - 1
    if npatch == N and H == W:
        return pos_embed
    pe = pos_embed.float()
    dim = pe.shape[-1]
    M = int(math.sqrt(N))
    assert N == M * M
    grid = F.interpolate(pe[:, 1:].reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic", antialias=True,
                         size=(H // patch, W // patch))
    return torch.cat((pe[:, 0].unsqueeze(0), grid.permute(0, 2, 3, 1).view(1, -1, dim)), dim=1)


def dino_patch_tokens(sd, prefix, x, cfg):
    """vision_transformer.py:214-226 (prepare tokens), :252-268 (blocks, final norm, drop cls+reg).
    Square input of the configured size only (interpolate_pos_encoding short-circuit :184-185)."""
    p = cfg["patch_size"]
    t = F.conv2d(x, sd[prefix + ".patch_embed.proj.weight"], sd[prefix + ".patch_embed.proj.bias"], stride=p)
    t = t.flatten(2).transpose(1, 2)                                       # patch_embed.py:73-74
    t = torch.cat((sd[prefix + ".cls_token"].expand(t.shape[0], -1, -1), t), dim=1)
    t = t + interpolate_pos_encoding(sd[prefix + ".pos_embed"], t.shape[1] - 1, x.shape[-2], x.shape[-1], p)
    reg = sd[prefix + ".register_tokens"]
    t = torch.cat((t[:, :1], reg.expand(t.shape[0], -1, -1), t[:, 1:]), dim=1)
    for i in range(cfg["dino_depth"]):
        t = block(sd, f"{prefix}.blocks.{i}", t, cfg["dino_heads"], None, False, eps=1e-6)
    t = _ln(t, sd, prefix + ".norm", 1e-6)
    return t[:, reg.shape[1] + 1:]


def slice_expand_and_flatten(tok, B, S):
    """aggregator.py:308-331: index 0 for frame 0, index 1 for the other S-1 frames."""
    q = tok[:, 0:1].expand(B, 1, *tok.shape[2:])
    o = tok[:, 1:].expand(B, S - 1, *tok.shape[2:])
    return torch.cat([q, o], dim=1).reshape(B * S, *tok.shape[2:])


def aggregator_forward(sd, images, cfg, keep_layers=None):
    """aggregator.py:184-258.  images [B,S,3,H,W] in [0,1] -> {layer: [B,S,P,2C]}, patch_start_idx"""
    B, S, Cin, H, W = images.shape
    if Cin != 3:
        raise ValueError(f"Expected 3 input channels, got {Cin}")
    mean = torch.tensor(_RESNET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(_RESNET_STD).view(1, 1, 3, 1, 1)
    x = ((images - mean) / std).view(B * S, Cin, H, W)
    A = "aggregator"
    p = cfg["patch_size"]
    if "conv" in cfg["patch_embed"]:
        pt = F.conv2d(x, sd[A + ".patch_embed.proj.weight"], sd[A + ".patch_embed.proj.bias"], stride=p)
        pt = pt.flatten(2).transpose(1, 2)
    else:
        pt = dino_patch_tokens(sd, A + ".patch_embed", x, cfg)
    C = pt.shape[-1]
    cam = slice_expand_and_flatten(sd[A + ".camera_token"], B, S)
    reg = slice_expand_and_flatten(sd[A + ".register_token"], B, S)
    tokens = torch.cat([cam, reg, pt], dim=1)
    nsp = 1 + cfg["num_register_tokens"]
    pos = positions_2d(B * S, H // p, W // p, nsp)
    P = tokens.shape[1]
    out = {}
    nh = cfg["num_heads"]
    for i in range(cfg["depth"]):
        tokens = block(sd, f"{A}.frame_blocks.{i}", tokens.view(B * S, P, C), nh, pos.view(B * S, P, 2), True)
        frame_out = tokens.view(B, S, P, C)
        tokens = block(sd, f"{A}.global_blocks.{i}", tokens.view(B, S * P, C), nh, pos.view(B, S * P, 2), True)
        glob_out = tokens.view(B, S, P, C)
        if keep_layers is None or i in keep_layers:
            out[i] = torch.cat([frame_out, glob_out], dim=-1)   # aggregator.py:250-253
    return out, nsp


# ---------------------------------------------------------------------------------------
# Camera head  (vggt/vggt/heads/camera_head.py, head_act.py:12-35)
# ---------------------------------------------------------------------------------------
def camera_head_forward(sd, tokens_last, cfg, num_iterations=4):
    """camera_head.py:73-141.  tokens_last [B,S,P,2C] -> list of [B,S,9]"""
    H = "camera_head"
    pose_tokens = _ln(tokens_last[:, :, 0], sd, H + ".token_norm")
    B, S, D = pose_tokens.shape
    pred = None
    outs = []
    for _ in range(num_iterations):
        if pred is None:
            mod_in = _lin(sd[H + ".empty_pose_tokens"].expand(B, S, -1), sd, H + ".embed_pose")
        else:
            mod_in = _lin(pred, sd, H + ".embed_pose")
        shift, scale, gate = _lin(F.silu(mod_in), sd, H + ".poseLN_modulation.1").chunk(3, dim=-1)
        x = gate * (F.layer_norm(pose_tokens, (D,), None, None, 1e-6) * (1 + scale) + shift) + pose_tokens
        for i in range(cfg["cam_trunk_depth"]):
            x = block(sd, f"{H}.trunk.{i}", x, cfg["cam_heads"], None, False)
        x = _ln(x, sd, H + ".trunk_norm")
        delta = _lin(F.gelu(_lin(x, sd, H + ".pose_branch.fc1")), sd, H + ".pose_branch.fc2")
        pred = delta if pred is None else pred + delta
        # activate_pose: T linear, quat linear, FoV relu (head_act.py:12-35; camera_head.py:135-137)
        outs.append(torch.cat([pred[..., :3], pred[..., 3:7], F.relu(pred[..., 7:])], dim=-1))
    return outs


# ---------------------------------------------------------------------------------------
# DPT head  (vggt/vggt/heads/dpt_head.py, heads/utils.py, head_act.py:61-125)
# ---------------------------------------------------------------------------------------
def _sincos_1d(embed_dim, pos, omega_0=100.0):
    """heads/utils.py:38-64 (float64 on CPU, cast to float at the end)"""
    omega = torch.arange(embed_dim // 2, dtype=torch.double)
    omega /= embed_dim / 2.0
    omega = 1.0 / omega_0 ** omega
    out = torch.einsum("m,d->md", pos.reshape(-1), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1).float()


def uv_pos_embed(w, h, channels, aspect_ratio, ratio=0.1):
    """create_uv_grid (heads/utils.py:67-109) + position_grid_to_embed (:11-35) as called from
    dpt_head.py:249-259.  Returns [C, h, w] already scaled by `ratio`."""
    diag = (aspect_ratio ** 2 + 1.0) ** 0.5
    sx, sy = aspect_ratio / diag, 1.0 / diag
    xs = torch.linspace(-sx * (w - 1) / w, sx * (w - 1) / w, steps=w, dtype=torch.float32)
    ys = torch.linspace(-sy * (h - 1) / h, sy * (h - 1) / h, steps=h, dtype=torch.float32)
    uu, vv = torch.meshgrid(xs, ys, indexing="xy")          # each [h, w]
    flat = torch.stack((uu, vv), dim=-1).reshape(-1, 2)
    emb = torch.cat([_sincos_1d(channels // 2, flat[:, 0]), _sincos_1d(channels // 2, flat[:, 1])], dim=-1)
    return (emb.view(h, w, channels) * ratio).permute(2, 0, 1)


def _interp(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)   # dpt_head.py:459-484


def _rcu(sd, prefix, x):
    """ResidualConvUnit, dpt_head.py:344-386.  The activation is an in-place ReLU applied to the
    INPUT, so the skip that is added back is relu(x) (and the caller's tensor is mutated)."""
    x = F.relu(x)
    out = F.conv2d(x, sd[prefix + ".conv1.weight"], sd[prefix + ".conv1.bias"], padding=1)
    out = F.conv2d(F.relu(out), sd[prefix + ".conv2.weight"], sd[prefix + ".conv2.bias"], padding=1)
    return out + x


def _fusion(sd, prefix, x0, x1=None, size=None):
    """FeatureFusionBlock.forward, dpt_head.py:427-456"""
    out = x0
    if x1 is not None:
        out = out + _rcu(sd, prefix + ".resConfUnit1", x1)
    out = _rcu(sd, prefix + ".resConfUnit2", out)
    if size is None:
        size = (out.shape[-2] * 2, out.shape[-1] * 2)
    out = _interp(out, size)
    return F.conv2d(out, sd[prefix + ".out_conv.weight"], sd[prefix + ".out_conv.bias"])


def inverse_log_transform(y):
    """head_act.py:115-125"""
    return torch.sign(y) * torch.expm1(torch.abs(y))


def dpt_forward(sd, prefix, tokens, H, W, patch_start_idx, cfg, activation="inv_log", feature_only=False,
                down_ratio=1, pos_embed=True):
    """DPTHead._forward_impl, dpt_head.py:172-247 (frame chunking, :140-170, does not change results).
    tokens: {layer: [B,S,P,2C]}"""
    p = cfg["patch_size"]
    ph, pw = H // p, W // p
    feats = []
    for di, layer in enumerate(cfg["dpt_layers"]):
        x = tokens[layer][:, :, patch_start_idx:]
        B, S = x.shape[:2]
        x = _ln(x.reshape(B * S, -1, x.shape[-1]), sd, prefix + ".norm")
        x = x.permute(0, 2, 1).reshape(B * S, x.shape[-1], ph, pw)
        x = F.conv2d(x, sd[f"{prefix}.projects.{di}.weight"], sd[f"{prefix}.projects.{di}.bias"])
        if pos_embed:
            x = x + uv_pos_embed(pw, ph, x.shape[1], W / H)[None]
        if di == 0:
            x = F.conv_transpose2d(x, sd[prefix + ".resize_layers.0.weight"], sd[prefix + ".resize_layers.0.bias"], stride=4)
        elif di == 1:
            x = F.conv_transpose2d(x, sd[prefix + ".resize_layers.1.weight"], sd[prefix + ".resize_layers.1.bias"], stride=2)
        elif di == 3:
            x = F.conv2d(x, sd[prefix + ".resize_layers.3.weight"], sd[prefix + ".resize_layers.3.bias"], stride=2, padding=1)
        feats.append(x)
    sc = prefix + ".scratch"
    l1, l2, l3, l4 = (F.conv2d(f, sd[f"{sc}.layer{i + 1}_rn.weight"], None, padding=1) for i, f in enumerate(feats))
    out = _fusion(sd, sc + ".refinenet4", l4, None, size=l3.shape[2:])     # dpt_head.py:261-291
    out = _fusion(sd, sc + ".refinenet3", out, l3, size=l2.shape[2:])
    out = _fusion(sd, sc + ".refinenet2", out, l2, size=l1.shape[2:])
    out = _fusion(sd, sc + ".refinenet1", out, l1)
    out = F.conv2d(out, sd[sc + ".output_conv1.weight"], sd[sc + ".output_conv1.bias"], padding=1)
    out = _interp(out, (int(ph * p / down_ratio), int(pw * p / down_ratio)))
    if pos_embed:
        out = out + uv_pos_embed(out.shape[-1], out.shape[-2], out.shape[1], W / H)[None]
    if feature_only:
        return out.view(B, S, *out.shape[1:])
    out = F.conv2d(out, sd[sc + ".output_conv2.0.weight"], sd[sc + ".output_conv2.0.bias"], padding=1)
    out = F.conv2d(F.relu(out), sd[sc + ".output_conv2.2.weight"], sd[sc + ".output_conv2.2.bias"])
    fmap = out.permute(0, 2, 3, 1)                                         # head_act.py:61-112
    xyz, conf = fmap[..., :-1], fmap[..., -1]
    if activation == "exp":
        pts = torch.exp(xyz)
    elif activation == "inv_log":
        pts = inverse_log_transform(xyz)
    else:
        raise ValueError(activation)
    conf = 1 + conf.exp()
    return pts.view(B, S, *pts.shape[1:]), conf.view(B, S, *conf.shape[1:])


# ---------------------------------------------------------------------------------------
# Track head  (vggt/vggt/heads/track_head.py, track_modules/*)
# ---------------------------------------------------------------------------------------
def bilinear_sampler(inp, coords, padding_mode="border"):
    """track_modules/utils.py:124-195 (align_corners=True; coords in pixels (x, y))"""
    sizes = inp.shape[2:]
    scale = torch.tensor([2 / max(s - 1, 1) for s in reversed(sizes)], dtype=coords.dtype)
    grid = coords * scale - 1
    return F.grid_sample(inp, grid, align_corners=True, padding_mode=padding_mode)


def sample_features4d(inp, coords):
    """track_modules/utils.py:198-223: [B,C,H,W], [B,R,2] -> [B,R,C]"""
    B = inp.shape[0]
    feats = bilinear_sampler(inp, coords.unsqueeze(2))
    return feats.permute(0, 2, 1, 3).reshape(B, -1, feats.shape[1] * feats.shape[3])


def get_2d_embedding(xy, C):
    """track_modules/utils.py:90-121 with cat_coords=False"""
    x, y = xy[:, :, 0:1], xy[:, :, 1:2]
    div = (torch.arange(0, C, 2, dtype=torch.float32) * (1000.0 / C)).reshape(1, 1, C // 2)
    B, N, _ = xy.shape
    pe_x = torch.zeros(B, N, C)
    pe_y = torch.zeros(B, N, C)
    pe_x[:, :, 0::2], pe_x[:, :, 1::2] = torch.sin(x * div), torch.cos(x * div)
    pe_y[:, :, 0::2], pe_y[:, :, 1::2] = torch.sin(y * div), torch.cos(y * div)
    return torch.cat([pe_x, pe_y], dim=2)


def sincos_pos_embed_2d(embed_dim, gh, gw):
    """track_modules/utils.py:18-87: [1, D, gh, gw]; omega base 10000, float64 -> float"""
    grid_h = torch.arange(gh, dtype=torch.float)
    grid_w = torch.arange(gw, dtype=torch.float)
    grid = torch.stack(torch.meshgrid(grid_w, grid_h, indexing="xy"), dim=0).reshape(2, 1, gh, gw)

    def one(d, pos):
        omega = torch.arange(d // 2, dtype=torch.double)
        omega /= d / 2.0
        omega = 1.0 / 10000 ** omega
        out = torch.einsum("m,d->md", pos.reshape(-1), omega)
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1)[None].float()

    emb = torch.cat([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], dim=2)
    return emb.reshape(1, gh, gw, -1).permute(0, 3, 1, 2)


def _mha(sd, prefix, q_in, kv_in, heads):
    """nn.MultiheadAttention(batch_first=True) forward, no mask (modules.py:150,184-186)"""
    E = q_in.shape[-1]
    w, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(kv_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(kv_in, w[2 * E:], b[2 * E:])
    B, Nq, _ = q.shape
    hd = E // heads
    sp = lambda t: t.reshape(B, -1, heads, hd).transpose(1, 2)  # noqa: E731
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, Nq, E)
    return F.linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


def _tmlp(sd, prefix, x):
    return _lin(F.gelu(_lin(x, sd, prefix + ".fc1")), sd, prefix + ".fc2")


def attn_block(sd, prefix, x, heads):
    """AttnBlock.forward, modules.py:155-169: x = norm1(x) FIRST, then the residual adds onto the
    normed x (not a standard pre-LN block)."""
    x = _ln(x, sd, prefix + ".norm1")
    x = x + _mha(sd, prefix + ".attn", x, x, heads)
    return x + _tmlp(sd, prefix + ".mlp", _ln(x, sd, prefix + ".norm2"))


def cross_attn_block(sd, prefix, x, context, heads):
    """CrossAttnBlock.forward, modules.py:190-204 (same normed-x residual)."""
    x = _ln(x, sd, prefix + ".norm1")
    context = _ln(context, sd, prefix + ".norm_context")
    x = x + _mha(sd, prefix + ".cross_attn", x, context, heads)
    return x + _tmlp(sd, prefix + ".mlp", _ln(x, sd, prefix + ".norm2"))


def update_former(sd, prefix, x, cfg):
    """EfficientUpdateFormer.forward, track_modules/blocks.py:90-134.  x [B, N, T, D_in]"""
    heads, nv = cfg["track_heads"], cfg["track_virtual"]
    tokens = _lin(_ln(x, sd, prefix + ".input_norm"), sd, prefix + ".input_transform")
    init = tokens
    B, _, T, _ = tokens.shape
    tokens = torch.cat([tokens, sd[prefix + ".virual_tracks"].repeat(B, 1, T, 1)], dim=1)
    N = tokens.shape[1]
    for i in range(cfg["track_depth"]):
        tt = attn_block(sd, f"{prefix}.time_blocks.{i}", tokens.contiguous().view(B * N, T, -1), heads)
        tokens = tt.view(B, N, T, -1)
        st = tokens.permute(0, 2, 1, 3).contiguous().view(B * T, N, -1)
        pt, vt = st[:, : N - nv], st[:, N - nv:]
        vt = cross_attn_block(sd, f"{prefix}.space_virtual2point_blocks.{i}", vt, pt, heads)
        vt = attn_block(sd, f"{prefix}.space_virtual_blocks.{i}", vt, heads)
        pt = cross_attn_block(sd, f"{prefix}.space_point2virtual_blocks.{i}", pt, vt, heads)
        tokens = torch.cat([pt, vt], dim=1).view(B, T, N, -1).permute(0, 2, 1, 3)
    tokens = tokens[:, : N - nv] + init
    return _lin(_ln(tokens, sd, prefix + ".output_norm"), sd, prefix + ".flow_head")


def tracker_forward(sd, prefix, query_points, fmaps, cfg, iters=4, stride=2, max_scale=518):
    """BaseTrackerPredictor.forward, track_modules/base_track_predictor.py:82-209 and
    CorrBlock (blocks.py:137-236).  fmaps [B,S,C,HH,WW]; query_points [B,N,2] pixels."""
    B, N, _ = query_points.shape
    _, S, C, HH, WW = fmaps.shape
    fmaps = _ln(fmaps.permute(0, 1, 3, 4, 2), sd, prefix + ".fmap_norm").permute(0, 1, 4, 2, 3)
    qp = query_points / float(stride)
    coords = qp.clone().reshape(B, 1, N, 2).repeat(1, S, 1, 1)
    qfeat = sample_features4d(fmaps[:, 0], coords[:, 0])
    track_feats = qfeat.unsqueeze(1).repeat(1, S, 1, 1)
    coords_backup = coords.clone()
    levels, r = cfg["track_corr_levels"], cfg["track_corr_radius"]
    pyr = [fmaps]
    cur = fmaps
    for _ in range(levels - 1):
        b_, s_, c_, h_, w_ = cur.shape
        cur = F.avg_pool2d(cur.reshape(b_ * s_, c_, h_, w_), 2, stride=2)
        cur = cur.reshape(b_, s_, c_, cur.shape[-2], cur.shape[-1])
        pyr.append(cur)
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)
    L = cfg["track_features"]
    tdim = 3 * L + 4
    preds = []
    for _ in range(iters):
        samples = []
        for li, fm in enumerate(pyr):
            _, _, _, h_, w_ = fm.shape
            corrs = torch.matmul(track_feats, fm.view(B, S, C, h_ * w_)) / math.sqrt(C)
            cl = coords.reshape(B * S * N, 1, 1, 2) / (2 ** li) + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
            smp = bilinear_sampler(corrs.reshape(B * S * N, 1, h_, w_), cl, padding_mode="zeros")
            samples.append(smp.view(B, S, N, -1))
        fcorrs = torch.cat(samples, dim=-1)
        fc = _tmlp(sd, prefix + ".corr_mlp", fcorrs.permute(0, 2, 1, 3).reshape(B * N, S, -1))
        flows = (coords - coords[:, 0:1]).permute(0, 2, 1, 3).reshape(B * N, S, 2)
        femb = torch.cat([get_2d_embedding(flows, L // 2), flows / max_scale, flows / max_scale], dim=-1)
        tf = track_feats.permute(0, 2, 1, 3).reshape(B * N, S, L)
        tin = torch.cat([femb, fc, tf], dim=2)
        pe = sample_features4d(sincos_pos_embed_2d(tdim, HH, WW).expand(B, -1, -1, -1), coords[:, 0])
        x = tin + pe.reshape(B * N, tdim).unsqueeze(1)
        qrt = sd[prefix + ".query_ref_token"]
        x = x + torch.cat([qrt[:, 0:1], qrt[:, 1:2].expand(-1, S - 1, -1)], dim=1)
        dlt = update_former(sd, prefix + ".updateformer", x.view(B, N, S, tdim), cfg).reshape(B * N, S, -1)
        dc, df = dlt[:, :, :2], dlt[:, :, 2:].reshape(B * N * S, L)
        gn = F.group_norm(df, 1, sd[prefix + ".ffeat_norm.weight"], sd[prefix + ".ffeat_norm.bias"])
        tf = F.gelu(_lin(gn, sd, prefix + ".ffeat_updater.0")) + tf.reshape(B * N * S, L)
        track_feats = tf.reshape(B, N, S, L).permute(0, 2, 1, 3)
        coords = coords + dc.reshape(B, N, S, 2).permute(0, 2, 1, 3)
        coords[:, 0] = coords_backup[:, 0]
        preds.append(coords * stride)
    flat = track_feats.reshape(B * S * N, L)
    vis = torch.sigmoid(_lin(flat, sd, prefix + ".vis_predictor.0")).reshape(B, S, N)
    conf = torch.sigmoid(_lin(flat, sd, prefix + ".conf_predictor.0")).reshape(B, S, N)
    return preds, vis, conf


# ---------------------------------------------------------------------------------------
# Full model  (vggt/vggt/models/vggt.py:29-96)
# ---------------------------------------------------------------------------------------
def vggt_forward(sd, images, cfg, query_points=None):
    if images.dim() == 4:
        images = images.unsqueeze(0)
    if query_points is not None and query_points.dim() == 2:
        query_points = query_points.unsqueeze(0)
    B, S, _, H, W = images.shape
    keep = set(cfg["dpt_layers"]) | {cfg["depth"] - 1}
    tokens, psi = aggregator_forward(sd, images, cfg, keep)
    out = {}
    if cfg.get("enable_camera", True):
        lst = camera_head_forward(sd, tokens[cfg["depth"] - 1], cfg)
        out["pose_enc"], out["pose_enc_list"] = lst[-1], lst
    if cfg.get("enable_depth", True):
        d, c = dpt_forward(sd, "depth_head", tokens, H, W, psi, cfg, activation="exp")
        out["depth"], out["depth_conf"] = d, c
    if cfg.get("enable_point", True):
        pts, c = dpt_forward(sd, "point_head", tokens, H, W, psi, cfg, activation="inv_log")
        out["world_points"], out["world_points_conf"] = pts, c
    if cfg.get("enable_track", True) and query_points is not None:
        tcfg = dict(cfg)
        tcfg["dpt_features"] = cfg["track_features"]
        fm = dpt_forward(sd, "track_head.feature_extractor", tokens, H, W, psi, _with(cfg, dpt_features=cfg["track_features"]),
                         feature_only=True, down_ratio=2, pos_embed=False)
        tr, vis, conf = tracker_forward(sd, "track_head.tracker", query_points, fm, cfg, iters=cfg["track_iters"])
        out["track"], out["vis"], out["conf"] = tr[-1], vis, conf
    return out


def _with(cfg, **kw):
    c = dict(cfg)
    c.update(kw)
    return c


# ---------------------------------------------------------------------------------------
# Geometry post-proc  (vggt/vggt/utils/{pose_enc,rotation,geometry}.py, vggt/vggt/infer.py)
# ---------------------------------------------------------------------------------------
def quat_to_mat(q):
    """rotation.py:14-44 (XYZW, scalar last; two_s = 2/sum(q^2))"""
    i, j, k, r = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def pose_encoding_to_extri_intri(pose, image_size_hw):
    """pose_enc.py:62-124"""
    T, quat, fov_h, fov_w = pose[..., :3], pose[..., 3:7], pose[..., 7], pose[..., 8]
    E = torch.cat([quat_to_mat(quat), T[..., None]], dim=-1)
    H, W = image_size_hw
    K = torch.zeros(pose.shape[:2] + (3, 3))
    K[..., 0, 0] = (W / 2.0) / torch.tan(fov_w / 2.0)
    K[..., 1, 1] = (H / 2.0) / torch.tan(fov_h / 2.0)
    K[..., 0, 2] = W / 2
    K[..., 1, 2] = H / 2
    K[..., 2, 2] = 1.0
    return E, K


def unproject_depth_map_to_point_map(depth, E, K):
    """geometry.py:15-117 (numpy, per frame): cam = ((u-cx)d/fx, (v-cy)d/fy, d) -> float32;
    world = cam . R_c2w^T + t_c2w with [R|t]^-1 = [R^T | -R^T t]"""
    depth, E, K = np.asarray(depth), np.asarray(E), np.asarray(K)
    out = []
    for f in range(depth.shape[0]):
        d = depth[f].squeeze(-1) if depth[f].ndim == 3 else depth[f]
        Hh, Ww = d.shape
        u, v = np.meshgrid(np.arange(Ww), np.arange(Hh))
        cam = np.stack(((u - K[f, 0, 2]) * d / K[f, 0, 0], (v - K[f, 1, 2]) * d / K[f, 1, 1], d), axis=-1).astype(np.float32)
        R, t = E[f, :3, :3], E[f, :3, 3]
        Rc = R.T
        tc = -(Rc @ t[:, None])[:, 0]
        out.append(np.dot(cam, Rc.T) + tc)
    return np.stack(out, axis=0)


def triangulate_point(P_list, x_list):
    """vggt/triangulate.py:19-34 (DLT), written for V views: rows u*P[2]-P[0], v*P[2]-P[1] per view;
    solution = last right-singular vector of A, dehomogenised.  V = 2 is the reference's function."""
    A = []
    for P, (u, v) in zip(P_list, x_list):
        A.append(u * P[2] - P[0])
        A.append(v * P[2] - P[1])
    _, _, Vt = np.linalg.svd(np.stack(A, axis=0))
    X = Vt[-1]
    return (X / X[3])[:3]


def triangulate_one_frame(K, R, T, kpts):
    """vggt/triangulate.py:38-71 without the I/O: K, R [V,3,3], T [V,3], kpts [V,J,2] -> X3d [J,3] float32.
    make_P = K @ [R | t] (triangulate.py:13-16)."""
    Ps = [K[v] @ np.concatenate([R[v], T[v].reshape(3, 1)], axis=1) for v in range(K.shape[0])]
    J = kpts.shape[1]
    X3d = np.zeros((J, 3), dtype=np.float32)
    for j in range(J):
        X3d[j] = triangulate_point(Ps, [kpts[v, j] for v in range(K.shape[0])])
    return X3d
