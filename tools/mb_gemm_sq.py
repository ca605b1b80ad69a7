"""Square bf16 GEMM timing (plain epilogue) for the gemm256 main-loop variants."""
import os, sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16
from tools.microbench import timeit
for S in (4096, 8192):
    a = torch.randn(S, S, device="cuda").to(torch.bfloat16)
    w = (torch.randn(S, S, device="cuda") / math.sqrt(S)).to(torch.bfloat16)
    b = torch.randn(S, device="cuda")
    o = torch.empty(S, S, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, w, prec=PREC_BF16, bias=b, out=o))
    t2 = timeit(lambda: torch.matmul(a, w.T))
    print(f"PP={os.environ.get('SKIMI_GEMM256_PP','0')} ABL={os.environ.get('SKIMI_GEMM256_ABL','0')} {S}^3: {t*1e6:8.1f} us {2*S**3/t/1e12:6.0f} TF/s | hipBLASLt {t2*1e6:8.1f} us {2*S**3/t2/1e12:6.0f} TF/s", flush=True)
