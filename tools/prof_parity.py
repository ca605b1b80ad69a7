"""Three parity-mode (bf16x3 everywhere) forwards of B (argv[1], default 4) 8-view 518 x 518 steps with all heads, for rocprofv3 --kernel-trace."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vggt, weights as W
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
cfg = W.VGGTConfig()
m = vggt.VGGT(config=cfg, prec=PREC_BF16X3, head_prec=PREC_BF16X3)
m.load_state_dict(W.make_vggt_state_dict(cfg, seed=0, device="cuda"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
img = torch.rand(B, 8, 3, 518, 518, device="cuda")
q = torch.rand(B, 17, 2, device="cuda") * 400 + 50
for _ in range(3):
    m(img, query_points=q)
torch.cuda.synchronize()
