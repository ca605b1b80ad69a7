#!/usr/bin/env python
"""Kernel micro-benchmarks on the GPU box (TF/s of the MFMA kernels, us of the lifter)."""
import math
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops, vp3d, weights as W  # noqa: E402
from skiing_analysis_pytorch_amd._lib import ACT_GELU, PREC_BF16, PREC_BF16X3  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm():
    for (M, N, K) in [(10992, 3072, 1024), (10992, 1024, 1024), (10992, 4096, 1024), (10992, 1024, 4096), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(N, K, device=DEV) / math.sqrt(K)
        ab, wb = a.to(torch.bfloat16), w.to(torch.bfloat16)
        out_b = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        out_f = torch.empty(M, N, device=DEV)
        t = timeit(lambda: ops.gemm(ab, wb, prec=PREC_BF16, out=out_b))
        print(f"gemm bf16  {M}x{N}x{K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s")
        t = timeit(lambda: ops.gemm(a, wb, prec=PREC_BF16, out=out_b))
        print(f"gemm f32A  {M}x{N}x{K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s")
        t = timeit(lambda: ops.gemm(a, w, prec=PREC_BF16X3, out=out_f), iters=5)
        print(f"gemm x3    {M}x{N}x{K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s (useful)")
        # torch (hipBLASLt) for scale
        t = timeit(lambda: torch.matmul(ab, wb.T))
        print(f"torch bf16 {M}x{N}x{K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s")


def bench_attn():
    for (batch, seq) in [(8, 1374), (1, 10992), (1, 2748)]:
        heads, hd = 16, 64
        qkv = torch.randn(batch * seq, 3 * heads * hd, device=DEV).to(torch.bfloat16)
        t = timeit(lambda: ops.attention(qkv, batch, seq, heads, hd))
        fl = 4.0 * batch * seq * seq * heads * hd
        print(f"attn bf16 b{batch} n{seq}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
        x = qkv.reshape(batch, seq, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = x[0].contiguous(), x[1].contiguous(), x[2].contiguous()
        t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
        print(f"torch sdpa b{batch} n{seq}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
    qkv = torch.randn(2748, 3 * 16 * 64, device=DEV)
    t = timeit(lambda: ops.attention(qkv, 1, 2748, 16, 64), iters=5)
    print(f"attn f32 n2748: {t*1e6:8.1f} us  {4.0*2748*2748*1024/t/1e12:7.2f} TF/s")


def bench_vp3d():
    for fw in ([3, 3, 3], [3, 3, 3, 3, 3]):
        for prec, nm in ((PREC_BF16X3, "x3"), (PREC_BF16, "bf16")):
            m = vp3d.TemporalModel(17, 2, 17, fw, prec=prec)
            m.load_state_dict(W.make_vp3d_state_dict(seed=0, filter_widths=fw))
            rf = m.receptive_field()
            for B in (1, 2):
                x = torch.randn(B, 243 + rf - 1, 17, 2, device=DEV)
                t = timeit(lambda: m(x), iters=50)
                print(f"vp3d rf{rf} {nm} B={B}: {t*1e6:8.1f} us/clip-call  {243/t:10.0f} frames/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "vp3d"]
    print(torch.cuda.get_device_name(0))
    for w in which:
        {"gemm": bench_gemm, "attn": bench_attn, "vp3d": bench_vp3d}[w]()
