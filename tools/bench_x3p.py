# bf16x3 LDS-DMA kernel: plain-row contraction vs the 3x3 conv gather of the same size
import os, sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
from tools.microbench import timeit
D="cuda"
n,H,W,C,Co,k = 32,148,148,256,256,3
M=n*H*W; K=k*k*C
for kind in ("randn","zeros"):
    mk = (lambda *s: torch.randn(*s,device=D)) if kind=="randn" else (lambda *s: torch.zeros(*s,device=D))
    w=mk(Co,K)/math.sqrt(K); ws=ops.split_records(w)
    o=torch.empty(M,Co,device=D)
    # conv gather
    x=mk(M,C); sc=torch.empty(ops.x3_scratch_numel(M,C),device=D)
    conv=dict(N=n,H=H,W=W,C=C,KH=k,KW=k,stride=1,pad=1,dil=1,OH=H,OW=W,slice_major=True)
    t=timeit(lambda: ops.gemm(x,w,prec=PREC_BF16X3,conv=conv,out=o,w_split=ws,x3_scratch=sc),iters=5)
    ts=timeit(lambda: ops.split_records(x),iters=5)
    fl=2.0*M*Co*K
    print(f"{kind} conv3x3 gather : {t*1e6:.0f} us, split {ts*1e6:.0f} -> kernel {(t-ts)*1e6:.0f} us = {3*fl/(t-ts)/1e12:.0f} TF/s MFMA work", flush=True)
    del x, sc
    # plain rows, same M, N, K (A is 9x bigger)
    Mp = M // 4
    xp=mk(Mp,K); scp=torch.empty(ops.x3_scratch_numel(Mp,K),device=D); op=torch.empty(Mp,Co,device=D)
    t=timeit(lambda: ops.gemm(xp,w,prec=PREC_BF16X3,out=op,w_split=ws,x3_scratch=scp),iters=5)
    ts=timeit(lambda: ops.split_records(xp),iters=5)
    flp=2.0*Mp*Co*K
    print(f"{kind} plain rows M={Mp}: {t*1e6:.0f} us, split {ts*1e6:.0f} -> kernel {(t-ts)*1e6:.0f} us = {3*flp/(t-ts)/1e12:.0f} TF/s MFMA work", flush=True)
    del xp, scp
