"""TemporalModel lifter timing, steady-state calls with fixed buffers: the weight-streaming path
(vp3d_stream.hip, small batches) against the generic GEMM chain (SKIMI_VP3D_STREAM=0 in a second process),
RF 27 (269 -> 243 frames) and RF 243 (485 -> 243), with the achieved fraction of the HBM roofline on
SURVEY §8(d)'s algorithmic bytes (fp32 weights once + B x (input + output)).

    python tools/mb_vp3d.py            # current path
    SKIMI_VP3D_STREAM=0 python tools/mb_vp3d.py
"""
import os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vp3d, weights as W
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
from tools.microbench import timeit

print("SKIMI_VP3D_STREAM =", os.environ.get("SKIMI_VP3D_STREAM", "(default 1)"))
for fw, lin in (([3, 3, 3], 269), ([3, 3, 3, 3, 3], 485)):
    sd = W.make_vp3d_state_dict(seed=0, filter_widths=fw)
    m = vp3d.TemporalModel(17, 2, 17, fw, prec=PREC_BF16X3)
    m.load_state_dict(sd)
    wbytes = sum(v.numel() * 4 for k, v in sd.items() if k.endswith("weight") and v.dim() == 3)
    for B in (1, 2, 4, 8, 64):
        x = torch.randn(B, lin, 17, 2, device="cuda")
        out = torch.empty(B, 243, 17, 3, device="cuda")
        t = timeit(lambda: m(x, out=out))
        alg = wbytes + B * (lin * 34 + 243 * 51) * 4
        print(f"RF {m.receptive_field():3d} B={B:2d}: {t*1e6:8.1f} us/call {t*1e6/B:7.1f} us/clip  {alg/t/1e9:7.1f} GB/s = {alg/t/8e12*100:5.2f} % of 8 TB/s",
              flush=True)
