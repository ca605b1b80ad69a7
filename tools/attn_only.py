import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
qkv = torch.randn(10992, 3*16*64, device="cuda").to(torch.bfloat16)
for _ in range(3):
    ops.attention(qkv, 1, 10992, 16, 64)
torch.cuda.synchronize()
