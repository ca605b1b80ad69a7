"""The track extractor's 3 x 3 convolutions (32 frames, 128 output channels, fp16 operands) on the halo-window kernel
(csrc/conv_win.hip, SKIMI_CONV_WIN=2) against the generic implicit-gather kernel (SKIMI_CONV_WIN=0), interleaved."""
import os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["SKIMI_ENV_DYNAMIC"] = "1"
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_F16, ACT_RELU
from tools.microbench import timeit

for n, H, W, C in ((32, 148, 148, 128), (32, 296, 296, 128), (32, 74, 74, 128), (32, 148, 148, 256), (32, 37, 37, 128)):
    x = torch.randn((n, H, W, C), device="cuda").to(torch.float16)
    w = (torch.randn((128, 9 * C), device="cuda") / (9 * C) ** 0.5).to(torch.float16)
    b = torch.randn(128, device="cuda")
    out = torch.empty((n * H * W, 128), device="cuda", dtype=torch.float16)
    conv = dict(N=n, H=H, W=W, C=C, KH=3, KW=3, stride=1, pad=1, dil=1, OH=H, OW=W)
    fl = 2.0 * n * H * W * 9 * C * 128
    res = {}
    for rnd in range(2):
        for mode in ("0", "2"):
            os.environ["SKIMI_CONV_WIN"] = mode
            t = timeit(lambda: ops.gemm(x.reshape(-1, C), w, prec=PREC_F16, bias=b, act=ACT_RELU, out=out, conv=conv))
            res.setdefault(mode, []).append(t)
    t0, t2 = min(res["0"]), min(res["2"])
    print(f"{n} x {H} x {W} x {C}: generic {t0*1e6:8.1f} us = {fl/t0/1e12:5.0f} TFLOP/s | window {t2*1e6:8.1f} us = {fl/t2/1e12:5.0f} TFLOP/s", flush=True)
