# upsample-into-records kernel at the two DPT shapes of the bench batch (32 frames)
import sys, torch, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vggt, weights as Wt
from skiing_analysis_pytorch_amd._lib import PREC_BF16, PREC_BF16X3
from tools.microbench import timeit
cfg = Wt.VGGTConfig(enable_track=False, enable_camera=False)
m = vggt.VGGT(config=cfg, prec=PREC_BF16, head_prec=PREC_BF16X3); m.load_state_dict(Wt.make_vggt_state_dict(cfg, seed=3, device="cuda"))
img = torch.rand(4, 8, 3, 518, 518, device="cuda")
m(img, want={"depth"}); torch.cuda.synchronize()
t = timeit(lambda: m(img, want={"depth", "point"}), iters=3, warm=1)
print(f"forward (depth + point heads, no camera): {t*1e3:.1f} ms")
