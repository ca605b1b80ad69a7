"""Two global-attention launches (batch 4 x seq 10992) for PMC collection."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
qkv = torch.randn(4 * 10992, 3 * 16 * 64, device="cuda").to(torch.bfloat16)
for _ in range(2):
    ops.attention(qkv, 4, 10992, 16, 64)
torch.cuda.synchronize()
