"""MXFP8 Linear layers at the bench shapes (M = 4 time steps x 8 views x 1374 tokens = 43968 rows) against the bf16
256-row kernels: quantisation pass, contraction (128 / 256 tiles via SKIMI_FP8_TILE in separate runs), bf16 GEMM."""
import os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import PREC_BF16, ACT_GELU
from tools.microbench import timeit
M = 43968
print("SKIMI_FP8_TILE =", os.environ.get("SKIMI_FP8_TILE", "(auto)"))
for name, N, K in (("qkv", 3072, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5)
    wb = w.to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aq, asx = ops.quant_mx(a)
    wq, wsx = ops.quant_mx(w)
    fl = 2.0 * M * N * K
    tq = timeit(lambda: ops.quant_mx(a))
    t8 = timeit(lambda: ops.gemm_fp8(aq, asx, wq, wsx, K, bias=bias, out=out))
    t16 = timeit(lambda: ops.gemm(a, wb, prec=PREC_BF16, bias=bias, out=out, out_dtype=torch.bfloat16))
    print(f"{name}: quant {tq*1e6:7.1f} us ({(M*K*3+M*K//32)/tq/1e12:.2f} TB/s) | fp8 gemm {t8*1e6:7.1f} us = {fl/t8/1e12:6.0f} TFLOP/s | bf16 gemm {t16*1e6:7.1f} us = {fl/t16/1e12:6.0f} TFLOP/s", flush=True)

# quantisation fused into the producers (round 2): LayerNorm -> MXFP8 and fc1's GELU epilogue -> MXFP8
x = torch.randn(M, 1024, device="cuda")
g = torch.ones(1024, device="cuda"); b = torch.zeros(1024, device="cuda")
t_ln = timeit(lambda: ops.layernorm(x, g, b, 1e-5, out_dtype=torch.bfloat16))
xb = ops.layernorm(x, g, b, 1e-5, out_dtype=torch.bfloat16)
t_q = timeit(lambda: ops.quant_mx(xb))
t_lnmx = timeit(lambda: ops.layernorm_mx(x, g, b, 1e-5))
print(f"LayerNorm(1024): -> bf16 {t_ln*1e6:.1f} us + quant {t_q*1e6:.1f} us | -> MXFP8 {t_lnmx*1e6:.1f} us", flush=True)
K, N = 1024, 4096
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = torch.randn(N, K, device="cuda") / K ** 0.5
bias = torch.randn(N, device="cuda")
aq, asx = ops.quant_mx(a); wq, wsx = ops.quant_mx(w)
hid = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
t_b = timeit(lambda: ops.gemm_fp8(aq, asx, wq, wsx, K, bias=bias, act=ACT_GELU, out=hid))
t_hq = timeit(lambda: ops.quant_mx(hid))
t_m = timeit(lambda: ops.gemm_fp8(aq, asx, wq, wsx, K, bias=bias, act=ACT_GELU, out_mx=True))
print(f"fc1 + GELU: -> bf16 {t_b*1e6:.1f} us + quant {t_hq*1e6:.1f} us | -> MXFP8 {t_m*1e6:.1f} us", flush=True)
