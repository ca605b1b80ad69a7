"""TemporalModel (RF 27) lifter timing, steady-state calls with fixed buffers.
(A hipGraph replay of the 12-node chain was tried and measured no faster: 104 vs 96 us at B=1.)"""
import os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vp3d, weights as W
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3, PREC_BF16
from tools.microbench import timeit
for prec, name in ((PREC_BF16X3, "bf16x3"), (PREC_BF16, "bf16")):
    m = vp3d.TemporalModel(17, 2, 17, [3, 3, 3], prec=prec)
    m.load_state_dict(W.make_vp3d_state_dict(seed=0, filter_widths=[3, 3, 3]))
    for B in (1, 8, 64):
        x = torch.randn(B, 243, 17, 2, device="cuda")
        out = torch.empty(B, 217, 17, 3, device="cuda")
        t = timeit(lambda: m(x, out=out))
        print(f"{name} B={B}: {t*1e6:8.1f} us/call {t*1e6/B:7.1f} us/clip  {B*243/t:10.0f} frames/s", flush=True)
