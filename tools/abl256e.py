# power / data-dependence check: the same 8192^3 contraction on random, constant and zero operands
import os, sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16
from tools.microbench import timeit
M, N, K = 8192, 8192, 8192
o=torch.empty(M,N,device="cuda",dtype=torch.bfloat16)
for kind in ("randn", "ones", "zeros", "randn"):
    if kind == "randn":
        a=torch.randn(M,K,device="cuda").to(torch.bfloat16); w=(torch.randn(N,K,device="cuda")/math.sqrt(K)).to(torch.bfloat16)
    elif kind == "ones":
        a=torch.ones(M,K,device="cuda",dtype=torch.bfloat16); w=torch.ones(N,K,device="cuda",dtype=torch.bfloat16)
    else:
        a=torch.zeros(M,K,device="cuda",dtype=torch.bfloat16); w=torch.zeros(N,K,device="cuda",dtype=torch.bfloat16)
    t=timeit(lambda: ops.gemm(a,w,prec=PREC_BF16,out=o), iters=30, warm=5)
    t2=timeit(lambda: torch.matmul(a,w.T), iters=30, warm=5)
    print(f"W4={os.environ.get('SKIMI_GEMM256_W4','0')} {kind:6s}: ours {t*1e6:7.1f} us {2*M*N*K/t/1e12:5.0f} TF/s | hipBLASLt {t2*1e6:7.1f} us {2*M*N*K/t2/1e12:5.0f} TF/s", flush=True)
