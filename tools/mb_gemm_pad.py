"""Does a padded leading dimension (no power-of-two row stride) change the gemm256 DMA rate?"""
import os, sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16
from tools.microbench import timeit
for (M, N, K) in [(8192, 8192, 8192), (43968, 4096, 1024), (43968, 1024, 4096)]:
    for pad in (0, 64, 192):
        ab = torch.randn(M, K + pad, device="cuda").to(torch.bfloat16)
        wb = (torch.randn(N, K + pad, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
        a, w = ab[:, :K], wb[:, :K]
        b = torch.randn(N, device="cuda")
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, w, prec=PREC_BF16, bias=b, out=o))
        print(f"PP={os.environ.get('SKIMI_GEMM256_PP','0')} ABL={os.environ.get('SKIMI_GEMM256_ABL','0')} M={M} N={N} K={K} pad={pad}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:6.0f} TF/s", flush=True)
