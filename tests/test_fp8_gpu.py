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


@pytest.mark.parametrize("rows,C", [(4099, 1024), (37, 256), (513, 768), (64, 2048)])
def test_layernorm_mx_equals_quantised_layernorm(rows, C):
    """LayerNorm -> MXFP8 in one kernel writes the bytes skimi_quant_mx makes of the same kernel's fp32 result (the
    arithmetic before the quantisation is the same instruction sequence), and those follow the NumPy restatement."""
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn((rows, C), generator=g) * (1 + 5 * torch.rand((rows, 1), generator=g)) + torch.randn((rows, 1), generator=g)).cuda()
    x[1] = 0.0                                     # constant row: LayerNorm gives beta
    gamma = (1 + 0.2 * torch.randn((C,), generator=g)).cuda()
    beta = (0.1 * torch.randn((C,), generator=g)).cuda()
    beta[32:64] = 0.0                              # an all-zero block on the constant row
    gamma[32:64] *= 1.0
    q, s = ops.layernorm_mx(x, gamma, beta, 1e-5)
    y = ops.layernorm(x, gamma, beta, 1e-5)
    q2, s2 = ops.quant_mx(y)
    assert torch.equal(s, s2) and torch.equal(q, q2)
    qr, sr = mx.mx_quantize(y.cpu().numpy())
    assert np.array_equal(s.cpu().numpy(), sr) and np.array_equal(q.cpu().numpy(), qr)
    assert s[1, 1].item() == 0 and not q[1, 32:64].any()


@pytest.mark.parametrize("M,N,K", [(5000, 4096, 1024), (4100, 2560, 640), (4096, 4096, 128), (5000, 2176, 256)])   # last: half a column tile
def test_gemm_fp8_mx_output(M, N, K):
    """fc1's epilogue writes GELU(. + bias) as the next GEMM's MXFP8 operand: against the NumPy quantisation of the
    float64 result.  The fp32 accumulator differs from float64 by ~1e-6 relative, so a value within that of a rounding
    boundary of e4m3 (or a block maximum within it of a power of two) may land on the neighbouring code: the dequantised
    results agree within one e4m3 step (2^-3 relative) on those and exactly elsewhere."""
    g = torch.Generator().manual_seed(M + N + K + 1)
    a = torch.randn((M, K), generator=g) * (1 + 3 * torch.rand((M, 1), generator=g))
    w = torch.randn((N, K), generator=g) / K ** 0.5
    bias = torch.randn((N,), generator=g)
    aq, asx = ops.quant_mx(a.cuda())
    wq, wsx = ops.quant_mx(w.cuda())
    ref = mx.mx_dequantize(aq.cpu().numpy(), asx.cpu().numpy()) @ mx.mx_dequantize(wq.cpu().numpy(), wsx.cpu().numpy()).T
    r = torch.nn.functional.gelu(torch.from_numpy(ref + bias.numpy().astype(np.float64))).numpy().astype(np.float32)
    q, s = ops.gemm_fp8(aq, asx, wq, wsx, K, bias=bias.cuda(), act=ACT_GELU, out_mx=True)
    torch.cuda.synchronize()
    assert q.shape == (M, N) and s.shape == (M, N // 32)
    # the same kernel's bf16 output is the rounded fp32 value: the MX bytes must be a quantisation of something within bf16
    # rounding of it -- and, tighter, dequantise to the oracle's quantisation of the float64 result
    got = mx.mx_dequantize(q.cpu().numpy(), s.cpu().numpy())
    qr, sr = mx.mx_quantize(r)
    want = mx.mx_dequantize(qr, sr)
    same_scale = s.cpu().numpy() == sr
    assert same_scale.mean() > 0.999
    blk = np.abs(r).reshape(M, N // 32, 32).max(-1)
    step = np.repeat(np.maximum(blk, 1e-30) * 2.0 ** -2, 32, axis=1).reshape(M, N)   # > one e4m3 step at the block's top binade, either scale
    diff = np.abs(got - want)
    assert (diff <= step).all()
    assert (diff == 0).mean() > 0.995
    # and the bytes are a valid operand of the next contraction (fc2): its result is the product of their dequantisation
    w2 = torch.randn((256, N), generator=g) / N ** 0.5
    w2q, w2s = ops.quant_mx(w2.cuda())
    o_f = ops.gemm_fp8(q, s, w2q, w2s, N)
    o_r = got.astype(np.float64) @ mx.mx_dequantize(w2q.cpu().numpy(), w2s.cpu().numpy()).T
    assert np.abs(o_f.cpu().numpy() - o_r).max() < 1e-4 * np.abs(o_r).max()


@pytest.mark.parametrize("batch,seq,heads", [(2, 300, 2), (1, 1374, 4), (3, 64, 6)])
def test_attention_mx_output(batch, seq, heads):
    """The attention kernel's MXFP8 output rows (skimi_attention_out with SKIMI_FP8MX: proj's operand under PREC_FP8,
    attention.py:60-64) are the quantisation of its fp32 quotient: against the float64 attention of the same bf16 q / k / v
    within the e4m3 step, and against the oracle quantiser applied to the kernel's own bf16 rows (those differ by the
    bf16 rounding in front of the quantiser only: same scale byte except at a power-of-two edge, payload within one code)."""
    hd, C = 64, heads * 64
    g = torch.Generator().manual_seed(seq + heads)
    qkv = (torch.randn(batch * seq, 3 * C, generator=g) * 1.5).to(torch.bfloat16).cuda()
    q8, s8 = ops.attention(qkv, batch, seq, heads, hd, out_dtype="fp8mx")
    o16 = ops.attention(qkv, batch, seq, heads, hd)
    assert q8.shape == (batch * seq, C) and s8.shape == (batch * seq, C // 32)
    got = mx.mx_dequantize(q8.cpu().numpy(), s8.cpu().numpy())
    x = qkv.double().view(batch, seq, 3, heads, hd).permute(2, 0, 3, 1, 4)
    p = torch.softmax(x[0] @ x[1].transpose(-1, -2) / 8.0, dim=-1)
    ref = (p @ x[2]).permute(0, 2, 1, 3).reshape(batch * seq, C).cpu().numpy()
    scale = np.exp2(s8.cpu().numpy().astype(np.float64) - 127.0).repeat(32, axis=1)
    # e4m3: 3 mantissa bits (half a step = 2^-4 relative), subnormal step 2^-9 of the block scale; the kernel's own bf16
    # probabilities and output rounding stay within 1.5e-2 of the row maximum (test_ops_gpu.py bounds the bf16 kernel at 2e-2)
    tol = 2.0 ** -4 * np.abs(ref) + scale * 2.0 ** -10 + 1.5e-2 * np.abs(ref).max(axis=1, keepdims=True)
    assert (np.abs(got - ref) <= tol).all(), float((np.abs(got - ref) - tol).max())
    qr, sr = mx.mx_quantize(o16.float().cpu().numpy())
    same_scale = (sr == s8.cpu().numpy())
    assert same_scale.mean() > 0.98 and np.abs(sr.astype(np.int32) - s8.cpu().numpy().astype(np.int32)).max() <= 1
    want = mx.mx_dequantize(qr, sr)
    blk = same_scale.repeat(32, axis=1)
    step = 2.0 ** -3 * np.maximum(np.abs(got), np.abs(want)) + scale * 2.0 ** -9
    assert (np.abs(got - want)[blk] <= step[blk]).all()
    assert (q8.cpu().numpy() == qr)[blk].mean() > 0.9


def test_attention_mx_output_refuses_other_shapes():
    """MXFP8 rows come from the 64-query bf16 kernel only: fp32 q / k / v, another head width or an odd number of heads
    (a row of 32-channel blocks that is not a whole number of 128-column K-tiles) are refused, not mis-written."""
    from skiing_analysis_pytorch_amd._lib import SkimiError
    for dtype, heads, hd in ((torch.float32, 2, 64), (torch.bfloat16, 2, 48), (torch.bfloat16, 3, 64)):
        qkv = torch.zeros(64, 3 * heads * hd, dtype=dtype, device="cuda")
        with pytest.raises(SkimiError):
            ops.attention(qkv, 1, 64, heads, hd, out_dtype="fp8mx")


def test_vggt_fp8_mode_against_reference(golden_dir):
    """PREC_FP8 (MXFP8 qkv / proj / fc1 / fc2 in every block, everything else as the bf16 mode) on the tiny golden: finite,
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


def test_vggt_fp8_fused_quantisation_path(monkeypatch):
    """The block path of SKIMI_PREC_FP8 at a size where the quantisation rides in the producers (LayerNorm -> MXFP8, the attention kernel's
    output rows -> MXFP8, fc1's GELU epilogue -> MXFP8 on the 256-row loop: 8 x 1374 = 10992 token rows x hidden 1024 = 172 tiles): against the fp32 CPU
    oracle, and against the same model with the fused path switched off (SKIMI_FP8_W4=0: separate quantisation passes of the
    bf16 activations).  The two fp8 variants differ by the bf16 rounding in front of the quantiser only."""
    from oracle import vggt_oracle
    cfg = W.VGGTConfig(img_size=518, embed_dim=256, depth=2, num_heads=4, patch_embed="conv", cam_trunk_depth=1, cam_heads=4,
                       dpt_layers=(0, 0, 1, 1), enable_depth=False, enable_point=False, enable_track=False)
    sd = W.make_vggt_state_dict(cfg, seed=5)
    images = W.make_images(8, 518, 518, seed=6)
    keep = {cfg.depth - 1}
    with torch.no_grad():
        tok, _ = vggt_oracle.aggregator_forward({k: v.float() for k, v in sd.items()}, images.unsqueeze(0), cfg.to_dict(), keep)
    ref = tok[cfg.depth - 1][0].numpy()
    m = vggt.VGGT(config=cfg, prec=PREC_FP8, head_prec=PREC_BF16X3)
    m.load_state_dict(sd)
    fused = m(images.cuda(), want={"camera"}, return_tokens=True)["tokens_last"].cpu().numpy()
    monkeypatch.setenv("SKIMI_FP8_W4", "0")        # re-read per launch: conftest sets SKIMI_ENV_DYNAMIC=1
    monkeypatch.setenv("SKIMI_ATTN_MX", "0")       # proj's operand: a quantisation pass over the attention's bf16 rows
    plain = m(images.cuda(), want={"camera"}, return_tokens=True)["tokens_last"].cpu().numpy()
    monkeypatch.delenv("SKIMI_FP8_W4")
    monkeypatch.delenv("SKIMI_ATTN_MX")
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    print(f"tokens rel err vs oracle: fused {rel(fused.reshape(ref.shape), ref):.3e}, separate passes {rel(plain.reshape(ref.shape), ref):.3e}; "
          f"fused vs separate {rel(fused, plain):.3e}")
    assert np.isfinite(fused).all() and not np.array_equal(fused, plain)
    assert rel(fused.reshape(ref.shape), ref) < 0.1 and rel(plain.reshape(ref.shape), ref) < 0.1
    assert rel(fused, plain) < 3e-2
