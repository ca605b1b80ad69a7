"""A few TemporalModel (RF 27) forwards for rocprofv3: B = 1 and B = 64 clips of 243 frames."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vp3d, weights as W
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
m = vp3d.TemporalModel(17, 2, 17, [3, 3, 3], prec=PREC_BF16X3)
m.load_state_dict(W.make_vp3d_state_dict(seed=0, filter_widths=[3, 3, 3]))
for B in (1, 64):
    x = torch.randn(B, 243, 17, 2, device="cuda")
    out = torch.empty(B, 217, 17, 3, device="cuda")
    for _ in range(5):
        m(x, out=out)
    torch.cuda.synchronize()
