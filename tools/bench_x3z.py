# data dependence (power) of the bf16x3 conv kernels: random vs zero operands, both loops
import os, sys, math, torch
os.environ["SKIMI_ENV_DYNAMIC"]="1"
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
from tools.microbench import timeit
D="cuda"
n,H,W,C,Co,k = 32,148,148,256,256,3
conv=dict(N=n,H=H,W=W,C=C,KH=k,KW=k,stride=1,pad=1,dil=1,OH=H,OW=W)
fl=2.0*n*H*W*Co*k*k*C
o=torch.empty(n*H*W,Co,device=D)
sc=torch.empty(ops.x3_scratch_numel(n*H*W,C),device=D)
for kind in ("randn","zeros","randn","zeros"):
    x=torch.randn(n*H*W,C,device=D) if kind=="randn" else torch.zeros(n*H*W,C,device=D)
    w=torch.randn(Co,k*k*C,device=D)/math.sqrt(k*k*C) if kind=="randn" else torch.zeros(Co,k*k*C,device=D)
    ws=ops.split_records(w)
    hi=torch.empty(n*H*W,C,device=D,dtype=torch.bfloat16); lo=torch.empty_like(hi)
    ts=timeit(lambda: ops.split_records(x), iters=5)
    for rep in range(2):
        t=timeit(lambda: ops.gemm(x,w,prec=PREC_BF16X3,conv=conv,out=o,w_split=ws,x3_scratch=sc),iters=5)
        print(f"{kind} run {rep}: {t*1e6:.0f} us total, split pass ~{ts*1e6:.0f} us -> conv kernel ~{(t-ts)*1e6:.0f} us = {3*fl/(t-ts)/1e12:.0f} TF/s of MFMA work", flush=True)
