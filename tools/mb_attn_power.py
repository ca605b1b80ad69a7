# global-attention launch of the bench (batch 4, seq 10992, 16 heads x 64) on random vs zero vs constant operands:
# how much of the kernel's time is the chip's power limit?
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from tools.microbench import timeit
B, S, Hh = 4, 10992, 16
fl = 4.0 * B * Hh * S * S * 64
for kind in ("randn", "zeros", "randn*0.1", "zeros"):
    if kind == "zeros": qkv = torch.zeros(B * S, 3 * Hh * 64, device="cuda", dtype=torch.bfloat16)
    else: qkv = (torch.randn(B * S, 3 * Hh * 64, device="cuda") * (0.1 if "0.1" in kind else 1.0)).to(torch.bfloat16)
    t = timeit(lambda: ops.attention(qkv, B, S, Hh, 64), iters=5, warm=2)
    print(f"{kind:10s}: {t*1e6:7.0f} us  {fl/t/1e12:6.0f} TF/s", flush=True)
