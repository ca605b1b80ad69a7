"""Global-attention timing (seq 10992, 16 heads x 64) incl. ablations via SKIMI_ATTN_ABL."""
import os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from tools.microbench import timeit
for (b, seq) in [(1, 10992), (4, 10992), (32, 1374)]:
    qkv = torch.randn(b * seq, 3 * 16 * 64, device="cuda").to(torch.bfloat16)
    t = timeit(lambda: ops.attention(qkv, b, seq, 16, 64))
    fl = 4.0 * b * 16 * seq * seq * 64
    print(f"ABL={os.environ.get('SKIMI_ATTN_ABL','0')} batch {b} seq {seq}: {t*1e6:8.1f} us {fl/t/1e12:6.0f} TF/s", flush=True)
