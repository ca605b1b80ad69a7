# epilogue share of the bench shapes: SKIMI_GEMM256_ABL=8 skips the epilogue
import os, sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16, ACT_GELU
from tools.microbench import timeit
D="cuda"; M=43968
for (N,K,kind) in [(3072,1024,"qkv"),(1024,1024,"proj"),(4096,1024,"fc1"),(1024,4096,"fc2")]:
    a=torch.randn(M,K,device=D).to(torch.bfloat16); w=(torch.randn(N,K,device=D)/math.sqrt(K)).to(torch.bfloat16)
    b=torch.randn(N,device=D); g=torch.rand(N,device=D); r=torch.randn(M,N,device=D)
    if kind in ("qkv","fc1"):
        o=torch.empty(M,N,device=D,dtype=torch.bfloat16)
        f=lambda: ops.gemm(a,w,prec=PREC_BF16,bias=b,act=ACT_GELU if kind=="fc1" else 0,out=o)
    else:
        f=lambda: ops.gemm(a,w,prec=PREC_BF16,bias=b,gamma=g,resid=r,out=r)
    t=timeit(f)
    print(f"W4={os.environ.get('SKIMI_GEMM256_W4','0')} ABL={os.environ.get('SKIMI_GEMM256_ABL','0')} {kind:4s} N={N} K={K}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:6.0f} TF/s", flush=True)
