"""Timing ablations of the halo-window conv kernel (csrc/conv_win.hip) at the extractor's largest launch (32 x 296 x 296 x 128,
fp16): needs a `SKIMI_ABLATIONS=1 python -m skiing_analysis_pytorch_amd.build` library (the variants are not in the product).
SKIMI_CONV_WIN_ABL bits: 1 no weight DMA in the loop, 2 no window DMA, 4 no MFMA, 8 no fragment reads, 16 no barrier.
Round 3: 1151 us full; 979 / 975 without the weight / window DMA; 870 without MFMAs; 916 without fragment reads; 1088 without
barriers; 588 with an empty loop (prologue, epilogue and the launch's 1.4 GB of HBM traffic)."""
import os, sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
os.environ["SKIMI_ENV_DYNAMIC"] = "1"; os.environ["SKIMI_CONV_WIN"] = "2"
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_F16, ACT_RELU
from tools.microbench import timeit
n, H, W, C = 32, 296, 296, 128
x = torch.randn((n, H, W, C), device="cuda").to(torch.float16)
w = (torch.randn((128, 9 * C), device="cuda") / (9 * C) ** 0.5).to(torch.float16)
b = torch.randn(128, device="cuda")
out = torch.empty((n * H * W, 128), device="cuda", dtype=torch.float16)
conv = dict(N=n, H=H, W=W, C=C, KH=3, KW=3, stride=1, pad=1, dil=1, OH=H, OW=W)
for abl in (0, 1, 2, 3, 4, 8, 12, 16, 7, 15, 31):
    os.environ["SKIMI_CONV_WIN_ABL"] = str(abl)
    t = timeit(lambda: ops.gemm(x.reshape(-1, C), w, prec=PREC_F16, bias=b, act=ACT_RELU, out=out, conv=conv))
    print(f"abl {abl:2d}: {t*1e6:8.1f} us", flush=True)
