"""DPT output conv (3x3, 128 -> 32, fp32-accurate) at 518x518: implicit-gather GEMM vs a plain GEMM of the same M, N, K."""
import sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
from tools.microbench import timeit
D = "cuda"
n, H, W, C, Co, k = 8, 518, 518, 128, 32, 3
x = torch.randn(n * H * W, C, device=D)
w = torch.randn(Co, k * k * C, device=D) / math.sqrt(k * k * C)
o = torch.empty(n * H * W, Co, device=D)
fl = 2.0 * n * H * W * Co * k * k * C
for sm in (False, True):
    conv = dict(N=n, H=H, W=W, C=C, KH=k, KW=k, stride=1, pad=1, dil=1, OH=H, OW=W, slice_major=sm)
    t = timeit(lambda: ops.gemm(x, w, prec=PREC_BF16X3, conv=conv, out=o, act=1), iters=5)
    print(f"conv3x3 128->32 @518^2 x{n} slice_major={sm}: {t*1e6:.0f} us  {fl/t/1e12:.0f} TF/s fp32-equivalent", flush=True)
# plain GEMM with the same M, N, K (A rows contiguous: no redundant gather)
M = n * H * W // 9 * 9 // 9   # keep memory modest: M/9 rows of K = 9*C contiguous
a2 = x[: M * 9].reshape(M, 9 * C)
o2 = torch.empty(M, Co, device=D)
t = timeit(lambda: ops.gemm(a2, w, prec=PREC_BF16X3, out=o2, act=1), iters=5)
print(f"plain GEMM M={M} N=32 K=1152: {t*1e6:.0f} us  {2.0*M*Co*9*C/t/1e12:.0f} TF/s   (x9 rows -> {t*9*1e6:.0f} us for the conv's M)", flush=True)
