# ablation timing of the bf16x3 single-stream loop on the 3x3 conv gather (32 x 148 x 148, 256 -> 256) (SKIMI_X3_ABL: 1 no staging,
# 2 no fragment reads, 3 MFMA only, 5 reads only, 6 staging only)
import os, sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
from tools.microbench import timeit
D="cuda"
n,H,W,C,k=32,148,148,256,3
M,N,K=n*H*W,256,k*k*C
kind=os.environ.get("KIND","randn")
mk = (lambda *s: torch.randn(*s,device=D)) if kind=="randn" else (lambda *s: torch.zeros(*s,device=D))
w=mk(N,K)/math.sqrt(K); ws=ops.split_records(w); x=mk(M,C)
conv=dict(N=n,H=H,W=W,C=C,KH=k,KW=k,stride=1,pad=1,dil=1,OH=H,OW=W,slice_major=True)
sc=torch.empty(ops.x3_scratch_numel(M,C),device=D); o=torch.empty(M,N,device=D)
t=timeit(lambda: ops.gemm(x,w,prec=PREC_BF16X3,conv=conv,out=o,w_split=ws,x3_scratch=sc),iters=5)
ts=timeit(lambda: ops.split_records(x),iters=5)
tiles=(M+255)//256; rounds=-(-tiles//256)
print(f"{kind} ABL={os.environ.get('SKIMI_X3_ABL','0')}: kernel {(t-ts)*1e6:.0f} us = {3*2.0*M*N*K/(t-ts)/1e12:.0f} TF/s MFMA work; {(t-ts)*1e6/rounds/(K/32):.2f} us per K-tile", flush=True)
