import os, sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16
from tools.microbench import timeit
M, N, K = 8192, 8192, 8192
a=torch.randn(M,K,device="cuda").to(torch.bfloat16); w=(torch.randn(N,K,device="cuda")/math.sqrt(K)).to(torch.bfloat16)
o=torch.empty(M,N,device="cuda",dtype=torch.bfloat16)
t=timeit(lambda: ops.gemm(a,w,prec=PREC_BF16,out=o))
print("DEEP", os.environ.get("SKIMI_GEMM256_DEEP","1"), "ABL", os.environ.get("SKIMI_GEMM256_ABL","0"), f"{t*1e6:.1f} us {2*M*N*K/t/1e12:.0f} TF/s  {t*1e6/4/128:.3f} us per K-tile", flush=True)
