# interleaved A/B timing of gemm256 variants inside one process (SKIMI_ENV_DYNAMIC=1):
#   python tools/ab_gemm.py "W4=0" "W4=1" "W4=1,MT3=0"
import os, sys, math, torch
os.environ["SKIMI_ENV_DYNAMIC"] = "1"
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16, ACT_GELU
from tools.microbench import timeit
variants = sys.argv[1:] or ["W4=0", "W4=1"]
D = "cuda"; M = int(os.environ.get("AB_M", 43968))
KEYS = set()
def setenv(v):
    for k in KEYS: os.environ.pop(k, None)
    for kv in v.split(","):
        if not kv: continue
        k, x = kv.split("="); k = "SKIMI_GEMM256_" + k; KEYS.add(k); os.environ[k] = x
shapes = [(3072,1024,"qkv"),(1024,1024,"proj"),(4096,1024,"fc1"),(1024,4096,"fc2")]
fns = {}
for (N,K,kind) in shapes:
    a=torch.randn(M,K,device=D).to(torch.bfloat16); w=(torch.randn(N,K,device=D)/math.sqrt(K)).to(torch.bfloat16)
    b=torch.randn(N,device=D); g=torch.rand(N,device=D); r=torch.randn(M,N,device=D)
    if kind in ("qkv","fc1"):
        o=torch.empty(M,N,device=D,dtype=torch.bfloat16)
        fns[kind]=(lambda a=a,w=w,b=b,o=o,kind=kind: ops.gemm(a,w,prec=PREC_BF16,bias=b,act=ACT_GELU if kind=="fc1" else 0,out=o))
    else:
        fns[kind]=(lambda a=a,w=w,b=b,g=g,r=r: ops.gemm(a,w,prec=PREC_BF16,bias=b,gamma=g,resid=r,out=r))
res = {(v,k): [] for v in variants for (_,_,k) in shapes}
for rep in range(4):
    for (N,K,kind) in shapes:
        for v in variants:
            setenv(v)
            res[(v,kind)].append(timeit(fns[kind], iters=10, warm=2))
for (N,K,kind) in shapes:
    for v in variants:
        ts = sorted(res[(v,kind)]); t = ts[len(ts)//2]
        print(f"{kind:4s} N={N} K={K} [{v:16s}]: median {t*1e6:7.1f} us {2*M*N*K/t/1e12:6.0f} TF/s   (min {ts[0]*1e6:.1f})", flush=True)
