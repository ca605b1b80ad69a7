"""How far does the fp32 CPU oracle's own track output move under a 1-ulp perturbation of its input?

The track head refines coordinates over 4 iterations; each iteration embeds the per-frame flow with
frequencies up to 1000 rad/px (get_2d_embedding, vggt/vggt/heads/track_modules/utils.py:107) and samples a
9x9 correlation window bilinearly around the current estimate, so rounding noise of iteration k re-enters
iteration k+1 amplified.  This script measures that amplification on the oracle itself: the same VGGT-1B
forward (synthetic weights, S views x 518 x 518, 17 queries) on `images` and on `images` with every pixel moved
by at most one fp32 ulp.  The result bounds what ANY other fp32 implementation (different summation order)
can be expected to reproduce, and is the justification of the pixel tolerance in
tests/test_bench_shape_gpu.py / tests/test_vggt_gpu.py.

    python tools/track_sensitivity.py [S]        # CPU only, ~1.5 min per forward at S = 8 on 8 cores
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import vggt_oracle  # noqa: E402
from skiing_analysis_pytorch_amd import weights as W  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = W.VGGTConfig(enable_depth=False, enable_point=False)
sd = W.make_vggt_state_dict(cfg, seed=0)
g = torch.Generator().manual_seed(1234)
images = torch.rand((1, S, 3, 518, 518), generator=g)
queries = torch.rand((1, 17, 2), generator=g) * (518 - 80) + 40
# one ulp up or down, at random
up = torch.nextafter(images, torch.full_like(images, 2.0))
dn = torch.nextafter(images, torch.full_like(images, -1.0))
pert = torch.where(torch.rand(images.shape, generator=g) < 0.5, up, dn).clamp(0.0, 1.0)
with torch.no_grad():
    a = vggt_oracle.vggt_forward(sd, images, cfg.to_dict(), query_points=queries)
    b = vggt_oracle.vggt_forward(sd, pert, cfg.to_dict(), query_points=queries)
d = (a["track"] - b["track"]).abs()
res = {"S": S, "input_perturbation": "every pixel +-1 fp32 ulp (<= 6e-8)",
       "pose_enc_max_abs_diff": float((a["pose_enc"] - b["pose_enc"]).abs().max()),
       "track_px_diff_max": float(d.max()), "track_px_diff_median": float(d.median()),
       "track_px_diff_p99": float(np.quantile(d.numpy(), 0.99)),
       "vis_diff_max": float((a["vis"] - b["vis"]).abs().max()), "conf_diff_max": float((a["conf"] - b["conf"]).abs().max())}
print(json.dumps(res))
out = ROOT / "profiles" / "r02_track_sensitivity.json"
out.write_text(json.dumps(res, indent=1) + "\n")
