"""The DPT heads' last 3 x 3 convolution (128 -> 32 at 518 x 518, fp32-accurate: bf16 hi / lo planes) on the direct halo-tile
kernel (csrc/conv_direct.hip) at the bench batch: 32 frames."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import _lib
from skiing_analysis_pytorch_amd._lib import check, lib, ptr
from tools.microbench import timeit
F_, H, W, C = 32, 518, 518, 128
rec = torch.randn((F_ * H * W, 2 * C), device="cuda").to(torch.bfloat16)      # [pixel][hi C | lo C]
w = torch.randn((32, C, 3, 3), device="cuda") / (9 * C) ** 0.5
packed = torch.empty(2 * 32 * C * 9, dtype=torch.bfloat16, device="cuda")
st = _lib.current_stream()
check(lib().skimi_conv3x3_n32_pack(ptr(w), ptr(packed), C, st), "pack")
bias = torch.randn(32, device="cuda")
out = torch.empty((F_, H, W, 32), device="cuda")
# the C-ABI entry takes separate planes; pixel records are the internal form (px_stride 2C) -- time the planes form
hi = rec[:, :C].contiguous(); lo = rec[:, C:].contiguous()
t = timeit(lambda: check(lib().skimi_conv3x3_n32(ptr(hi), ptr(lo), ptr(packed), ptr(bias), ptr(out), F_, H, W, C, 1, st), "conv"), iters=10)
fl = 2.0 * F_ * H * W * 9 * C * 32
print(f"conv_direct 128 -> 32 @ 518^2 x {F_}: {t*1e6:.0f} us = {fl/t/1e12:.0f} TFLOP/s fp32-equivalent ({3*fl/t/1e12:.0f} of MFMA issue)")
