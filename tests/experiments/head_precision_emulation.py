#!/usr/bin/env python
"""EXPERIMENT (test infrastructure, not product): operand precision of the DPT depth / point heads.

The reference runs these heads under autocast(enabled=False) (models/vggt.py:65), i.e. fp32 -- on its own GPU path
with cuDNN's default TF32 convolutions (11-bit significands).  This runs the oracle's DPT heads with torch on the GPU
in fp32 and emulates rounding of the operands of every convolution / transposed convolution (what a one-MFMA
contraction with fp16 / bf16 operands and fp32 accumulation does), on tokens from the fp16-operand aggregator
(tests/experiments/precision_emulation.py's 'f16 linears, bf16 attention' policy) and reports the relative error
|a - b| / (|b| + 1) of depth, world points and confidences against the all-fp32 run.

    python tests/experiments/head_precision_emulation.py --out gpurun_out/head_precision_emulation.json
"""
from __future__ import annotations

import argparse
import json
import sys
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import precision_emulation as pe  # noqa: E402
from oracle import vggt_oracle as vo  # noqa: E402
from skiing_analysis_pytorch_amd import weights as W  # noqa: E402

FMT = {"fmt": "f32"}


class _RoundingF(types.SimpleNamespace):
    """torch.nn.functional with the contraction operands of conv2d / conv_transpose2d rounded to FMT['fmt']"""

    def __getattr__(self, name):
        return getattr(F, name)

    @staticmethod
    def conv2d(x, w, b=None, **kw):
        return F.conv2d(pe.rnd(x, FMT["fmt"]), pe.rnd(w, FMT["fmt"]), b, **kw)

    @staticmethod
    def conv_transpose2d(x, w, b=None, **kw):
        return F.conv_transpose2d(pe.rnd(x, FMT["fmt"]), pe.rnd(w, FMT["fmt"]), b, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    cfg = W.VGGTConfig()
    d = cfg.to_dict()
    sd = {k: v.to(dev) for k, v in W.make_vggt_state_dict(cfg, seed=0).items() if not k.startswith("track_head")}
    S, IMG = 8, 518
    pe.CTX["S"], pe.CTX["P"] = S, 1 + cfg.num_register_tokens + (IMG // 14) ** 2
    gen = torch.Generator(device="cpu").manual_seed(5)
    img = torch.rand((1, S, 3, IMG, IMG), generator=gen).to(dev)
    keep = set(d["dpt_layers"])
    tok = {}
    real_F = vo.F
    with torch.no_grad(), torch.device(dev):
        for name, pol in (("fp32", None), ("f16agg", pe.Policy("f16 linears, bf16 attention", default="f16", rules={("qk",): "bf16", ("pv",): "bf16"}))):
            pe.CTX["policy"] = pol
            vo.block = pe._block_emul if pol is not None else pe._orig_block
            tok[name], psi = vo.aggregator_forward(sd, img, d, keep)
        vo.block = pe._orig_block
        rows = []

        def heads(tokens, fmt):
            FMT["fmt"] = fmt
            vo.F = _RoundingF() if fmt != "f32" else real_F
            try:
                dep, dc = vo.dpt_forward(sd, "depth_head", tokens, IMG, IMG, psi, d, activation="exp")
                pts, pc = vo.dpt_forward(sd, "point_head", tokens, IMG, IMG, psi, d, activation="inv_log")
            finally:
                vo.F = real_F
            return {"depth": dep, "depth_conf": dc, "world_points": pts, "world_points_conf": pc}

        ref = heads(tok["fp32"], "f32")
        for tname in ("fp32", "f16agg"):
            for fmt in ("f32", "f16", "bf16", "bf16x2"):
                got = heads(tok[tname], fmt)
                row = {"aggregator": "fp32" if tname == "fp32" else "f16 linears + bf16 attention (PREC_F16)", "head_conv_operands": fmt}
                for k, v in got.items():
                    rel = (v - ref[k]).abs() / (ref[k].abs() + 1.0)
                    row[k] = {"max": rel.max().item(), "median": rel.median().item(), "p999": rel.flatten().float().kthvalue(int(0.999 * rel.numel())).values.item()}
                row["every_output_within_1e-3"] = all(row[k]["max"] <= 1e-3 for k in got)
                rows.append(row)
                print(json.dumps(row), flush=True)
    res = {"what": "operand-rounding emulation of the DPT depth / point heads' convolutions (torch fp32 on the GPU); VGGT-1B synthetic weights, 8 views x "
                   "518x518; relative error |a - b| / (|b| + 1) against the all-fp32 run", "rows": rows}
    if args.out:
        Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
