import sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16, PREC_BF16X3
from tools.microbench import timeit
D="cuda"
import os
os.environ["SKIMI_ENV_DYNAMIC"]="1"
for (n,H,W,C,Co,k) in [(32,148,148,256,256,3),(32,74,74,256,256,3),(8,296,296,256,256,3),(8,296,296,256,128,3),(8,37,37,2048,256,1)]:
    x=torch.randn(n*H*W,C,device=D); w=torch.randn(Co,k*k*C,device=D)/math.sqrt(k*k*C)
    conv=dict(N=n,H=H,W=W,C=C,KH=k,KW=k,stride=1,pad=k//2,dil=1,OH=H,OW=W) if k>1 else None
    fl=2.0*n*H*W*Co*k*k*C
    o=torch.empty(n*H*W,Co,device=D)
    t=timeit(lambda: ops.gemm(x,w,prec=PREC_BF16X3,conv=conv,out=o),iters=5)
    ws=ops.split_records(w); sc=torch.empty(ops.x3_scratch_numel(x.shape[0],C),device=D)
    t4=timeit(lambda: ops.gemm(x,w,prec=PREC_BF16X3,conv=conv,out=o,w_split=ws,x3_scratch=sc),iters=5)
    xb=x.to(torch.bfloat16); wb=w.to(torch.bfloat16); ob=torch.empty(n*H*W,Co,device=D,dtype=torch.bfloat16)
    t3=timeit(lambda: ops.gemm(xb,wb,prec=PREC_BF16,conv=conv,out=ob),iters=5)
    print(f"conv{k} {H}x{W} {C}->{Co}: generic x3 {t*1e6:.0f} us {fl/t/1e12:.0f} TF/s | LDS-DMA records kernel {t4*1e6:.0f} us {fl/t4/1e12:.0f} TF/s | bf16 {t3*1e6:.0f} us {fl/t3/1e12:.0f} TF/s", flush=True)
