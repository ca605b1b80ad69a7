"""A few TemporalModel forwards for rocprofv3 (--kernel-trace --stats): RF 27, B = 1 clip of 269 -> 243 frames
(the weight-streaming path), then B = 64 (the LDS-DMA path)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vp3d, weights as W
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
m = vp3d.TemporalModel(17, 2, 17, [3, 3, 3], prec=PREC_BF16X3)
m.load_state_dict(W.make_vp3d_state_dict(seed=0, filter_widths=[3, 3, 3]))
batches = [int(v) for v in sys.argv[1:]] or [1, 64]
for B in batches:
    x = torch.randn(B, 269, 17, 2, device="cuda")
    out = torch.empty(B, 243, 17, 3, device="cuda")
    for _ in range(20):
        m(x, out=out)
    torch.cuda.synchronize()
