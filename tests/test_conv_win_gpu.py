"""GPU: the halo-window 3 x 3 convolution kernel (csrc/conv_win.hip: 16-bit single-term operands -> 128 channels, what the
track head's DPT feature extractor runs, dpt_head.py:261-291 under the autocast of vggt.py:85-91) against torch's conv2d on
the same rounded operands -- bit-exact on small integers (an index path: window gather, tap order, swizzles, ragged tiles),
fp32-accumulate tolerance on random data -- and against the generic implicit-gather kernel it replaces."""
import math

import pytest
import torch
import torch.nn.functional as F

from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import ACT_NONE, ACT_RELU, PREC_BF16, PREC_F16

pytestmark = pytest.mark.gpu


def _conv(x16, w16, prec, monkeypatch, mode, **kw):
    n, H, W_, C = x16.shape
    conv = dict(N=n, H=H, W=W_, C=C, KH=3, KW=3, stride=1, pad=1, dil=1, OH=H, OW=W_)
    monkeypatch.setenv("SKIMI_CONV_WIN", str(mode))   # re-read per launch: conftest sets SKIMI_ENV_DYNAMIC=1
    out = ops.gemm(x16.reshape(-1, C), w16, prec=prec, conv=conv, **kw)
    monkeypatch.delenv("SKIMI_CONV_WIN")
    return out


def _ref(x16, w16, bias=None):
    n, H, W_, C = x16.shape
    w = w16.double().reshape(128, 3, 3, C).permute(0, 3, 1, 2)
    r = F.conv2d(x16.double().permute(0, 3, 1, 2), w, None if bias is None else bias.double(), padding=1)
    return r.permute(0, 2, 3, 1).reshape(-1, 128)


@pytest.mark.parametrize("prec,dt", [(PREC_BF16, torch.bfloat16), (PREC_F16, torch.float16)])
@pytest.mark.parametrize("n,H,W_,C", [(2, 16, 16, 64), (3, 37, 41, 64), (1, 50, 18, 128), (2, 9, 70, 256), (1, 33, 20, 192)])
def test_conv_win_exact_on_integers(prec, dt, n, H, W_, C, monkeypatch):
    """small integers: every product and partial sum is exact in fp32, so the result must equal conv2d bit for bit whatever
    the summation order -- whole tiles, ragged right / bottom tiles, images smaller than a tile, 2 .. 8 channel slices"""
    g = torch.Generator().manual_seed(n * 1000 + H)
    x = torch.randint(-4, 5, (n, H, W_, C), generator=g).to(dt).cuda()
    w = torch.randint(-2, 3, (128, 9 * C), generator=g).to(dt).cuda()
    b = torch.randint(-8, 9, (128,), generator=g).float().cuda()
    out = _conv(x, w, prec, monkeypatch, 2, bias=b)
    ref = _ref(x, w, b)
    assert out.shape == ref.shape and torch.equal(out.double(), ref)
    assert torch.equal(out, _conv(x, w, prec, monkeypatch, 0, bias=b))   # the generic kernel agrees (and is a different launch)


@pytest.mark.parametrize("prec,dt", [(PREC_BF16, torch.bfloat16), (PREC_F16, torch.float16)])
def test_conv_win_random_data_and_epilogues(prec, dt, monkeypatch):
    """random operands: fp32 accumulation against float64 on the same 16-bit operands; the ResidualConvUnit epilogues of the
    extractor (bias + ReLU -> 16-bit rows; bias + 16-bit residual + ReLU -> 16-bit rows, dpt_head.py:376-380)"""
    n, H, W_, C = 2, 45, 52, 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn((n, H, W_, C), generator=g).to(dt).cuda()
    w = (torch.randn((128, 9 * C), generator=g) / math.sqrt(9 * C)).to(dt).cuda()
    b = torch.randn(128, generator=g).cuda()
    ref = _ref(x, w, b)
    out = _conv(x, w, prec, monkeypatch, 2, bias=b)
    assert (out.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    gen = _conv(x, w, prec, monkeypatch, 0, bias=b)
    assert (out - gen).abs().max().item() < 2e-5 * ref.abs().max().item()
    # 16-bit output rows with ReLU: the same values rounded once
    o16 = _conv(x, w, prec, monkeypatch, 2, bias=b, act=ACT_RELU, out_dtype=dt)
    assert o16.dtype == dt and torch.equal(o16, torch.relu(out).to(dt))
    # residual + post-activation
    r1 = torch.randn((n * H * W_, 128), generator=g).to(dt).cuda()
    r2 = torch.randn((n * H * W_, 128), generator=g).to(dt).cuda()
    o_res = _conv(x, w, prec, monkeypatch, 2, bias=b, resid=r1, post_act=ACT_RELU, out_dtype=dt)
    assert torch.equal(o_res, torch.relu(out + r1.float()).to(dt))
    o_gen = _conv(x, w, prec, monkeypatch, 0, bias=b, resid=r1, post_act=ACT_RELU, out_dtype=dt)
    assert (o_res.float() - o_gen.float()).abs().max().item() <= 2.0 ** -7 * o_gen.float().abs().max().item()


def test_conv_win_is_picked_for_the_extractor_shapes(monkeypatch):
    """default mode: the kernel takes the launches it is meant for (148 x 148 x 32 frames and up) and leaves small or ragged
    maps to the generic kernel -- checked through the results being identical either way on integers"""
    g = torch.Generator().manual_seed(9)
    for n, H, W_ in ((4, 148, 148), (32, 37, 37)):
        x = torch.randint(-3, 4, (n, H, W_, 128), generator=g).to(torch.float16).cuda()
        w = torch.randint(-2, 3, (128, 9 * 128), generator=g).to(torch.float16).cuda()
        a = _conv(x, w, PREC_F16, monkeypatch, 1)
        assert torch.equal(a.double(), _ref(x, w))
