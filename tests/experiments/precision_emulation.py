#!/usr/bin/env python
"""EXPERIMENT (test infrastructure, not product, not collected by pytest): which operand precision does the
aggregator need for the 3D joints to stay within north_star's 1e-3?

Runs the oracle's aggregator + camera head (oracle/vggt_oracle.py, the pinned restatement of the reference) with
torch on the GPU in fp32 and EMULATES operand rounding per class of operation: an MFMA with bf16 / fp16 operands
and fp32 accumulation is, to first order, an fp32 product of operands rounded to that format.  No HIP kernel of
the product is involved: this decides which kernels are worth writing.  Classes:

  group  dino | frame | global            (DINOv2 blocks, frame-attention blocks, global-attention blocks)
  op     qkv | qk | pv | proj | fc1 | fc2 (Linear operands; q,k of QK^T; P,V of PV)
  side   a | w                            (activation operand / weight operand of a Linear)

A policy maps (group, op, side) -> format in {f32, bf16, f16, bf16x2, f16x2} (x2 = hi + lo, what the bf16x3
kernels carry), optionally with the special-token rows (camera + register tokens) of the activation side
kept in another format.

Output: for each policy max |pose_enc - pose_enc_fp32|, the MPJPE of the 8-view DLT joints on the ring rig
(oracle/joints_check.py) and on the model's native (degenerate) scene -> JSON on stdout / --out.

    python tests/experiments/precision_emulation.py --out gpurun_out/precision_emulation.json
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import joints_check, vggt_oracle as vo  # noqa: E402
from skiing_analysis_pytorch_amd import weights as W  # noqa: E402

S_VIEWS, IMG = 8, 518


def rnd(x, fmt):
    if fmt == "f32":
        return x
    if fmt == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if fmt == "f16":
        return x.to(torch.float16).to(torch.float32)
    if fmt == "bf16x2":
        hi = x.to(torch.bfloat16).to(torch.float32)
        return hi + (x - hi).to(torch.bfloat16).to(torch.float32)
    if fmt == "f16x2":
        hi = x.to(torch.float16).to(torch.float32)
        return hi + (x - hi).to(torch.float16).to(torch.float32)
    raise ValueError(fmt)


class Policy:
    """fmt(group, op, side) with most-specific-first lookup; `special` = format of the special-token rows of
    every activation operand (None: same as the other rows)."""

    def __init__(self, name, default="f32", rules=None, special=None, n_special=5):
        self.name, self.default, self.special, self.n_special = name, default, special, n_special
        self.rules = {(k[0] if isinstance(k, tuple) and len(k) == 1 else k): v for k, v in (rules or {}).items()}

    def fmt(self, group, op, side):
        for key in ((group, op, side), (group, op), (group, side), (op, side), (group,), (op,), (side,)):
            k = key if len(key) > 1 else key[0]
            if k in self.rules:
                return self.rules[k]
        return self.default


CTX = {"policy": None, "S": S_VIEWS, "P": None}


def _group_of(prefix):
    if ".patch_embed.blocks." in prefix:
        return "dino"
    if ".frame_blocks." in prefix:
        return "frame"
    if ".global_blocks." in prefix:
        return "global"
    return None


def _round_rows(x, fmt, special_fmt, P, n_special):
    """x [..., N, C] token rows; rows with (n mod P) < n_special are the special tokens"""
    y = rnd(x, fmt)
    if special_fmt is not None and special_fmt != fmt:
        N = x.shape[-2]
        idx = torch.arange(N, device=x.device) % P < n_special
        y = torch.where(idx.view(*([1] * (x.dim() - 2)), N, 1), rnd(x, special_fmt), y)
    return y


def _lin_emul(x, sd, prefix, group, op):
    pol = CTX["policy"]
    a = _round_rows(x, pol.fmt(group, op, "a"), pol.special, CTX["P"], pol.n_special)
    w = rnd(sd[prefix + ".weight"], pol.fmt(group, op, "w"))
    return F.linear(a, w, sd.get(prefix + ".bias"))


def _attention_emul(sd, prefix, x, num_heads, pos, qk_norm, group):
    pol = CTX["policy"]
    B, N, C = x.shape
    hd = C // num_heads
    qkv = _lin_emul(x, sd, prefix + ".qkv", group, "qkv").reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    if qk_norm:
        q, k = vo._ln(q, sd, prefix + ".q_norm"), vo._ln(k, sd, prefix + ".k_norm")
    if pos is not None:
        q, k = vo.rope2d(q, pos), vo.rope2d(k, pos)
    P, ns, sp = CTX["P"], pol.n_special, pol.special
    fq, fk = pol.fmt(group, "qk", "a"), pol.fmt(group, "qk", "w")
    fp, fv = pol.fmt(group, "pv", "a"), pol.fmt(group, "pv", "w")
    if fq == fk == fp == fv == "f32" and sp in (None, "f32"):
        o = F.scaled_dot_product_attention(q, k, v)
    else:
        scale = hd ** -0.5
        qr = _round_rows(q * scale, fq, sp, P, ns)        # the kernels fold the scale into q before its rounding
        kr = _round_rows(k, fk, sp, P, ns)
        vr = _round_rows(v, fv, sp, P, ns)
        o = torch.empty_like(q)
        for b in range(B):
            for h0 in range(0, num_heads, 4):
                s = qr[b, h0:h0 + 4] @ kr[b, h0:h0 + 4].transpose(-1, -2)
                s = s - s.amax(dim=-1, keepdim=True)
                p = torch.exp(s)
                den = p.sum(dim=-1, keepdim=True)             # fp32 row sums of the unrounded p, as the kernels keep them
                p = _round_rows(p, fp, sp, P, ns)                # rows of p = query rows
                o[b, h0:h0 + 4] = (p @ vr[b, h0:h0 + 4]) / den
    return _lin_emul(o.transpose(1, 2).reshape(B, N, C), sd, prefix + ".proj", group, "proj")


_orig_block = vo.block


def _block_emul(sd, prefix, x, num_heads, pos=None, qk_norm=False, eps=1e-5):
    group = _group_of(prefix)
    if group is None or CTX["policy"] is None:
        return _orig_block(sd, prefix, x, num_heads, pos, qk_norm, eps)
    a = _attention_emul(sd, prefix + ".attn", vo._ln(x, sd, prefix + ".norm1", eps), num_heads, pos, qk_norm, group)
    x = x + a * sd[prefix + ".ls1.gamma"]
    h = _lin_emul(vo._ln(x, sd, prefix + ".norm2", eps), sd, prefix + ".mlp.fc1", group, "fc1")
    h = _lin_emul(F.gelu(h), sd, prefix + ".mlp.fc2", group, "fc2")
    return x + h * sd[prefix + ".ls2.gamma"]


def policies():
    ops = ("qkv", "qk", "pv", "proj", "fc1", "fc2")
    out = [Policy("fp32 (GPU torch; noise floor vs itself = 0)"),
           Policy("bf16 everywhere (the bench mode)", default="bf16"),
           Policy("f16 everywhere", default="f16"),
           Policy("bf16x2 everywhere (what the bf16x3 kernels carry)", default="bf16x2")]
    for fmt in ("bf16", "f16"):
        for g in ("dino", "frame", "global"):
            out.append(Policy(f"{fmt} in the {g} blocks only", rules={(g,): fmt}))
        for op in ops:
            out.append(Policy(f"{fmt} in {op} only (all groups)", rules={(op,): fmt}))
    # weights exact (hi + lo), activations single
    out.append(Policy("bf16 activations, bf16x2 weights (2 MFMAs per product)", default="bf16", rules={("w",): "bf16x2", ("qk", "w"): "bf16", ("pv", "w"): "bf16"}))
    out.append(Policy("f16 activations, f16x2 weights (2 MFMAs per product)", default="f16", rules={("w",): "f16x2", ("qk", "w"): "f16", ("pv", "w"): "f16"}))
    # special-token rows exact
    out.append(Policy("bf16, special-token rows bf16x2", default="bf16", special="bf16x2"))
    out.append(Policy("f16, special-token rows f16x2", default="f16", special="f16x2"))
    out.append(Policy("bf16 activations + bf16x2 weights, special-token rows bf16x2", default="bf16", special="bf16x2",
                      rules={("w",): "bf16x2", ("qk", "w"): "bf16", ("pv", "w"): "bf16"}))
    # mixes
    out.append(Policy("f16 linears, bf16 attention (qk, pv)", default="f16", rules={("qk",): "bf16", ("pv",): "bf16"}))
    out.append(Policy("bf16 linears, f16 attention (qk, pv)", default="bf16", rules={("qk",): "f16", ("pv",): "f16"}))
    out.append(Policy("f16 everywhere, bf16 dino", default="f16", rules={("dino",): "bf16"}))
    out.append(Policy("f16 dino + frame, bf16x2 global", default="f16", rules={("global",): "bf16x2"}))
    out.append(Policy("bf16 dino, bf16x2 frame + global", default="bf16x2", rules={("dino",): "bf16"}))
    out.append(Policy("f16 dino, bf16x2 frame + global", default="bf16x2", rules={("dino",): "f16"}))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--seeds", type=int, default=2, help="number of image seeds (time steps) per policy")
    ap.add_argument("--only", default=None, help="substring filter on policy names")
    ap.add_argument("--tiny", action="store_true", help="CPU-sized config (plumbing check of this script)")
    args = ap.parse_args()
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    torch.backends.cuda.matmul.allow_tf32 = False
    if args.tiny:
        g = np.load(ROOT / "tests" / "golden" / "vggt_tiny_dino.npz")
        cfg = W.VGGTConfig(**json.loads(str(g["cfg_json"])))
        img_hw, S = (int(g["H"]), int(g["W"])), int(g["S"])
    else:
        cfg = W.VGGTConfig()
        img_hw, S = (IMG, IMG), S_VIEWS
    d = cfg.to_dict()
    sd = W.make_vggt_state_dict(cfg, seed=0)
    sd = {k: v.to(dev) for k, v in sd.items() if k.startswith(("aggregator", "camera_head"))}
    CTX["S"] = S
    CTX["P"] = 1 + cfg.num_register_tokens + (img_hw[0] // cfg.patch_size) * (img_hw[1] // cfg.patch_size)
    vo.block = _block_emul
    imgs = []
    for sseed in range(args.seeds):
        gen = torch.Generator(device="cpu").manual_seed(5 + sseed)
        imgs.append(torch.rand((1, S, 3, *img_hw), generator=gen, device="cpu").to(dev))

    def pose_of(policy):
        CTX["policy"] = policy
        outs = []
        with torch.no_grad(), torch.device(dev):
            for im in imgs:
                tokens, _ = vo.aggregator_forward(sd, im, d, {d["depth"] - 1})
                outs.append(vo.camera_head_forward(sd, tokens[d["depth"] - 1], d)[-1])
        return torch.cat(outs).cpu()       # [T, S, 9]

    pols = [p for p in policies() if args.only is None or args.only in p.name]
    t0 = time.time()
    ref = pose_of(Policy("fp32"))
    print(f"# fp32 reference pass: {time.time() - t0:.1f} s on {dev}", file=sys.stderr, flush=True)
    ring, kps_ring, joints_ring = joints_check.ring_rig_scene(ref, img_hw, seed=5)
    kps_nat, Xw, joints_nat = joints_check.keypoints_from_oracle_cameras(ref, img_hw, seed=5)

    def dlt(pe, kps):
        E, K = vo.pose_encoding_to_extri_intri(pe.float().cpu(), img_hw)
        j = np.stack([vo.triangulate_one_frame(K[t].double().numpy(), E[t, :, :3, :3].double().numpy(), E[t, :, :3, 3].double().numpy(),
                                               kps[t].double().numpy()) for t in range(pe.shape[0])])
        return j

    rows = []
    for p in pols:
        t0 = time.time()
        pe = pose_of(p)
        err = (pe - ref).abs()
        row = {"policy": p.name, "pose_enc_max_abs_err": err.max().item(),
               "pose_enc_err_T_quat_fov": [err[..., :3].max().item(), err[..., 3:7].max().item(), err[..., 7:].max().item()],
               "mpjpe_ring_rig": joints_check.mpjpe(dlt(joints_check.ring_rig_test_pose_enc(ring, pe, ref), kps_ring), joints_ring),
               "mpjpe_native_scene": joints_check.mpjpe(dlt(pe, kps_nat), joints_nat),
               "seconds": time.time() - t0}
        row["within_1e-3_on_ring"] = row["mpjpe_ring_rig"] <= 1e-3
        rows.append(row)
        print(json.dumps(row), flush=True)
    res = {"what": "operand-rounding emulation (torch fp32 on the GPU) of the aggregator's precision classes; VGGT-1B synthetic weights, "
                   f"{S} views x {img_hw[0]}x{img_hw[1]}, {args.seeds} time steps; reference = the same code without rounding",
           "rows": rows}
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
