import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from tools.microbench import timeit
tokens, heads = 43968, 16
qkv = torch.randn(tokens, 3 * heads * 64, device="cuda").to(torch.bfloat16)
w = [torch.randn(64, device="cuda") for _ in range(4)]
pos = torch.stack([torch.arange(tokens) % 38, (torch.arange(tokens) * 7) % 38], -1).to(torch.int32).cuda()
cos_t = torch.rand(38, 16, device="cuda"); sin_t = torch.rand(38, 16, device="cuda")
t = timeit(lambda: ops.qknorm_rope_(qkv, heads, *w, 1e-5, pos, cos_t, sin_t))
print(f"qknorm_rope bf16 {tokens}x{heads}: {t*1e6:.1f} us  {tokens*heads*2*64*2*2/t/1e12:.2f} TB/s")
