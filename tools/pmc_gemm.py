"""One gemm256 launch set for PMC collection (rocprofv3 --pmc ... -- python tools/pmc_gemm.py M N K)."""
import sys, math, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16
M, N, K = (int(x) for x in sys.argv[1:4])
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
b = torch.randn(N, device="cuda")
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(a, w, prec=PREC_BF16, bias=b, out=o)
torch.cuda.synchronize()
