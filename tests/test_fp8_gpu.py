"""GPU: the MXFP8 path of BASELINE config 5 (csrc/gemm_fp8.hip) -- quantisation bit-exact against the NumPy
restatement of the OCP formats (oracle/mxfp8_oracle.py; byte / integer work), the scaled-MFMA contraction against
the float64 product of the dequantised operands, and the PREC_FP8 VGGT against the reference's golden outputs
with its error stated (the reference has no fp8 path: parity of this mode is reported, not pinned to 1e-3)."""
import json

import numpy as np
import pytest
import torch

from oracle import mxfp8_oracle as mx
from skiing_analysis_pytorch_amd import ops, vggt, weights as W
from skiing_analysis_pytorch_amd._lib import ACT_GELU, PREC_BF16, PREC_BF16X3, PREC_FP8

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,K", [(37, 200), (130, 1024), (5, 32)])
def test_quant_mx_bit_exact(dtype, rows, K):
    g = torch.Generator().manual_seed(rows * 1000 + K)
    x = torch.randn((rows, K), generator=g) * torch.exp(torch.randn((rows, 1), generator=g) * 3)   # rows of very different magnitude
    x[0, :32] = 0.0                                 # an all-zero block
    x[1, 3] = 448.0 * 4                             # amax exactly on a power-of-two boundary of amax / 448
    x[2, 7] = 3.0e38 if dtype == torch.float32 else 3.0e38   # near the top of the fp32 / bf16 range
    x[3, :] *= 1e-30                                # tiny values
    x = x.to(dtype)
    Kal = (K + 7) // 8 * 8
    xd = torch.zeros((rows, Kal), dtype=dtype)
    xd[:, :K] = x
    q, s = ops.quant_mx(xd.cuda()[:, :K]) if Kal == K else ops.quant_mx(xd[:, :K].contiguous().cuda()) if K % 8 == 0 else (None, None)
    if q is None:
        pytest.skip("rows must be 16-byte aligned")
    qr, sr = mx.mx_quantize(x.to(torch.float32).numpy())
    assert np.array_equal(s.cpu().numpy(), sr)
    assert np.array_equal(q.cpu().numpy(), qr)


# the last four shapes fill the chip with 256-row tiles and take the single-stream loop (gemm256w4_fp8_kernel): ragged M and N,
# 1, 2, 5 and 8 K-tiles (the peeled iterations of its software pipeline)
@pytest.mark.parametrize("M,N,K", [(300, 260, 384), (128, 128, 128), (1000, 3072, 1024), (77, 64, 4096),
                                   (5000, 2304, 1024), (4100, 2500, 640), (4096, 4096, 128), (4200, 2560, 256)])
def test_gemm_fp8_matches_dequantised_product(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g) * (1 + 3 * torch.rand((M, 1), generator=g))
    w = torch.randn((N, K), generator=g) / K ** 0.5
    bias = torch.randn((N,), generator=g)
    aq, asx = ops.quant_mx(a.cuda())
    wq, wsx = ops.quant_mx(w.cuda())
    ref = mx.mx_dequantize(aq.cpu().numpy(), asx.cpu().numpy()) @ mx.mx_dequantize(wq.cpu().numpy(), wsx.cpu().numpy()).T
    scale = np.abs(ref).max()
    out = ops.gemm_fp8(aq, asx, wq, wsx, K, bias=bias.cuda())
    torch.cuda.synchronize()
    # fp32 accumulation inside the MFMA (its internal summation order / rounding of a 64-long step is not ours): observed 2e-5
    assert np.abs(out.cpu().numpy() - (ref + bias.numpy())).max() < 1e-4 * scale
    # bf16 output, GELU epilogue
    o2 = ops.gemm_fp8(aq, asx, wq, wsx, K, bias=bias.cuda(), act=ACT_GELU, out_dtype=torch.bfloat16)
    r2 = torch.nn.functional.gelu(torch.from_numpy(ref + bias.numpy())).numpy()
    assert np.abs(o2.float().cpu().numpy() - r2).max() < 1e-2 * max(1.0, np.abs(r2).max())
    # LayerScale + residual, in place (fc2 of block.py:77-98)
    gamma = torch.rand((N,), generator=g) * 0.2
    x = torch.randn((M, N), generator=g)
    xd = x.cuda()
    ops.gemm_fp8(aq, asx, wq, wsx, K, bias=bias.cuda(), gamma=gamma.cuda(), resid=xd, out=xd)
    r3 = (ref + bias.numpy()) * gamma.numpy() + x.numpy()
    assert np.abs(xd.cpu().numpy() - r3).max() < 1e-4 * scale + 1e-6
    # the quantised product itself is an fp8 approximation of a @ w.T: block-scaled e4m3 ~ 2^-4 per element
    exact = a.double().numpy() @ w.double().numpy().T
    rel = np.linalg.norm(ref - exact) / np.linalg.norm(exact)
    assert rel < 5e-2, rel


def test_vggt_fp8_mode_against_reference(golden_dir):
    """PREC_FP8 (MXFP8 qkv / fc1 / fc2 in every block, everything else as the bf16 mode) on the tiny golden: finite,
    close to the bf16 mode, and its distance to the fp32 reference stated (fp8 operands: a few 1e-2 relative)."""
    gd = np.load(golden_dir / "vggt_tiny_conv.npz")
    cfg = W.VGGTConfig(**json.loads(str(gd["cfg_json"])))
    sd = W.make_vggt_state_dict(cfg, seed=int(gd["seed"]))
    images = W.make_images(int(gd["S"]), int(gd["H"]), int(gd["W"]), seed=int(gd["images_seed"])).cuda()
    outs = {}
    for name, prec in (("fp8", PREC_FP8), ("bf16", PREC_BF16)):
        m = vggt.VGGT(config=cfg, prec=prec, head_prec=PREC_BF16X3)
        m.load_state_dict(sd)
        outs[name] = m(images, want={"camera", "depth"}, return_tokens=True)
    ref = gd["tokens_last"]
    rel8 = np.linalg.norm(outs["fp8"]["tokens_last"].cpu().numpy() - ref) / np.linalg.norm(ref)
    rel16 = np.linalg.norm(outs["bf16"]["tokens_last"].cpu().numpy() - ref) / np.linalg.norm(ref)
    pe8 = np.abs(outs["fp8"]["pose_enc"].cpu().numpy() - gd["pose_enc"]).max()
    print(f"tokens rel err vs fp32 reference: fp8 {rel8:.3e}, bf16 {rel16:.3e}; pose_enc max abs err fp8 {pe8:.3e}")
    assert torch.isfinite(outs["fp8"]["depth"]).all()
    assert rel16 < 3e-2 and rel8 < 0.2 and rel8 > rel16       # fp8 is the coarser arithmetic, and it is really in use
    assert pe8 < 0.3
