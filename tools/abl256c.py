"""gemm256 epilogue-only timing (SKIMI_GEMM256_ABL=3) vs grid size: per-CU or aggregate store limit?"""
import os, sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16
from tools.microbench import timeit
K = 1024
for M, N in [(256, 4096), (1024, 4096), (4096, 4096), (8192, 4096), (16384, 4096), (43968, 3072)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, w, prec=PREC_BF16, bias=b, out=o))
    tiles = ((M + 255) // 256) * (N // 256)
    print(os.environ.get("SKIMI_GEMM256_ABL", "0"), M, N, "tiles", tiles, f"{t*1e6:.1f} us", f"{M*N*2/t/1e12:.2f} TB/s out", flush=True)
