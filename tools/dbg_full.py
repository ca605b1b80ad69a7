import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from skiing_analysis_pytorch_amd import vggt, weights as W, _lib
from skiing_analysis_pytorch_amd._lib import PREC_BF16, PREC_BF16X3
def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)
dev = torch.device("cuda", 0)
cfg = W.VGGTConfig()
log("create")
m = vggt.VGGT(config=cfg, prec=PREC_BF16, head_prec=PREC_BF16X3)
t=time.time(); sd = W.make_vggt_state_dict(cfg, seed=0, device=dev); torch.cuda.synchronize(); log("weights", time.time()-t)
t=time.time(); m.load_state_dict(sd); torch.cuda.synchronize(); log("loaded", time.time()-t)
del sd; torch.cuda.empty_cache()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
img = torch.rand((1, S, 3, 518, 518), device=dev)
for want in ({"camera"}, {"camera","depth"}, {"camera","depth","point"}):
    for it in range(2):
        t=time.time(); out = m(img, want=want); torch.cuda.synchronize(); log("forward", sorted(want), it, f"{(time.time()-t)*1e3:.1f} ms")
log("ws MB", m._ws.numel()/1e6)
