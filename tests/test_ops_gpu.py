"""GPU parity of the generic C-ABI ops against plain torch fp32 references of the same op."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from skiing_analysis_pytorch_amd import ops
from skiing_analysis_pytorch_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, PREC_BF16, PREC_BF16X3

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 104), (1374, 3072, 1024), (8, 6144, 2048), (267, 1024, 3072)])
@pytest.mark.parametrize("prec", [PREC_BF16X3, PREC_BF16])
def test_gemm_plain(M, N, K, prec):
    a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3)
    scratch = torch.empty(M * N, dtype=torch.float32, device=DEV)
    ref = a @ w.T + b
    wd = w if prec == PREC_BF16X3 else w.to(torch.bfloat16)
    out = ops.gemm(a, wd, prec=prec, bias=b, splitk_scratch=scratch)
    torch.cuda.synchronize()
    tol = 2e-5 if prec == PREC_BF16X3 else 1e-2
    assert _rel(out, ref) < tol
    if prec == PREC_BF16:  # bf16 A and bf16 out
        out2 = ops.gemm(a.to(torch.bfloat16), wd, prec=prec, bias=b, out_dtype=torch.bfloat16)
        assert _rel(out2.float(), ref) < 2e-2


def test_gemm_exact_integers_asymmetric():
    # exact small-integer data catches any swapped fragment / transposed C map
    M, N, K = 96, 160, 64
    a = torch.arange(M * K, device=DEV, dtype=torch.float32).reshape(M, K) % 7 - 3
    w = (torch.arange(N * K, device=DEV, dtype=torch.float32).reshape(N, K) * 3 % 11) - 5
    for prec, wd in ((PREC_BF16X3, w), (PREC_BF16, w.to(torch.bfloat16))):
        out = ops.gemm(a, wd, prec=prec)
        assert torch.equal(out, a @ w.T)


@pytest.mark.parametrize("splitk", [2, 5])
def test_gemm_splitk(splitk):
    M, N, K = 100, 192, 1024
    a, w, b = _rand(M, K, seed=4), _rand(N, K, seed=5, scale=1 / 32), _rand(N, seed=6)
    scratch = torch.empty(M * N, dtype=torch.float32, device=DEV)
    out = ops.gemm(a, w, bias=b, act=ACT_RELU, splitk_scratch=scratch, force_splitk=splitk)
    assert _rel(out, F.relu(a @ w.T + b)) < 2e-5


def test_gemm_epilogue_gelu_layerscale_residual():
    M, N, K = 300, 256, 128
    a, w, b, g, r = _rand(M, K, seed=7), _rand(N, K, seed=8, scale=0.1), _rand(N, seed=9), _rand(N, seed=10), _rand(M, N, seed=11)
    out = ops.gemm(a, w, bias=b, act=ACT_GELU)
    assert _rel(out, F.gelu(a @ w.T + b)) < 2e-5
    out = ops.gemm(a, w, bias=b, gamma=g, resid=r)
    assert _rel(out, r + g * (a @ w.T + b)) < 2e-5
    # in-place residual stream update (out aliases resid), as the transformer blocks do
    r2 = r.clone()
    ops.gemm(a, w, bias=b, gamma=g, resid=r2, out=r2)
    assert _rel(r2, r + g * (a @ w.T + b)) < 2e-5
    # odd N: scalar epilogue path
    w51, b51 = _rand(51, K, seed=12, scale=0.1), _rand(51, seed=13)
    assert _rel(ops.gemm(a, w51, bias=b51), a @ w51.T + b51) < 2e-5


@pytest.mark.parametrize("prec", [PREC_BF16X3, PREC_BF16])
@pytest.mark.parametrize("cfg", [dict(H=37, W=37, C=64, Co=96, k=3, s=1, p=1, d=1), dict(H=37, W=37, C=128, Co=64, k=3, s=2, p=1, d=1),
                                 dict(H=1, W=90, C=64, Co=64, k=3, s=1, p=0, d=9), dict(H=20, W=24, C=64, Co=32, k=1, s=1, p=0, d=1)])
def test_gemm_conv_gather(cfg, prec):
    H, W_, Cc, Co, k, s, p, d = (cfg[x] for x in ("H", "W", "C", "Co", "k", "s", "p", "d"))
    kh = 1 if H == 1 else k
    n = 2
    x = _rand(n, H, W_, Cc, seed=20)                      # channels-last
    w = _rand(Co, Cc, kh, k, seed=21, scale=1 / math.sqrt(Cc * kh * k))
    b = _rand(Co, seed=22)
    pad = (0 if H == 1 else p, p)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=s, padding=pad, dilation=(1 if H == 1 else d, d))
    OH, OW = ref.shape[2], ref.shape[3]
    wp = w.permute(0, 2, 3, 1).reshape(Co, kh * k * Cc).contiguous()   # [Co, ky, kx, Cin]
    if prec == PREC_BF16:
        wp = wp.to(torch.bfloat16)
    conv = dict(N=n, H=H, W=W_, C=Cc, KH=kh, KW=k, stride=s, pad=p if H > 1 else 0, dil=d, OH=OH, OW=OW)
    if H == 1:
        conv["pad"] = 0
    out = ops.gemm(x.reshape(-1, Cc), wp, prec=prec, bias=b, conv=conv)
    ref_cl = ref.permute(0, 2, 3, 1).reshape(-1, Co)
    assert _rel(out, ref_cl) < (2e-5 if prec == PREC_BF16X3 else 1e-2)


@pytest.mark.parametrize("cfg", [dict(H=37, W=37, C=64, Co=96, k=3, s=1, p=1), dict(H=19, W=23, C=128, Co=32, k=3, s=2, p=1),
                                 dict(H=30, W=30, C=256, Co=256, k=3, s=1, p=1), dict(H=50, W=47, C=256, Co=256, k=3, s=1, p=1),
                                 dict(H=61, W=40, C=96, Co=128, k=3, s=1, p=1), dict(H=121, W=117, C=64, Co=256, k=3, s=2, p=1)])
def test_gemm_conv_gather_slice_major_k(cfg, x3_small_shapes):
    """a_mode 2: the same 3x3 gather with K ordered [Cin/32][ky][kx][32] (weights packed to match); the
    shapes with >= 4096 output pixels and Cout >= 96 also go through the LDS-DMA bf16x3 kernels (256- and
    128-column tiles, ragged last tile rows, stride 2)."""
    H, W_, Cc, Co, k, s, p = (cfg[x] for x in ("H", "W", "C", "Co", "k", "s", "p"))
    n = 2
    x = _rand(n, H, W_, Cc, seed=20)
    w = _rand(Co, Cc, k, k, seed=21, scale=1 / math.sqrt(Cc * k * k))
    b = _rand(Co, seed=22)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=s, padding=p)
    OH, OW = ref.shape[2], ref.shape[3]
    # [Co, Cin/32, ky, kx, 32]
    wp = w.reshape(Co, Cc // 32, 32, k, k).permute(0, 1, 3, 4, 2).reshape(Co, k * k * Cc).contiguous()
    conv = dict(N=n, H=H, W=W_, C=Cc, KH=k, KW=k, stride=s, pad=p, dil=1, OH=OH, OW=OW, slice_major=True)
    ref_cl = ref.permute(0, 2, 3, 1).reshape(-1, Co)
    out = ops.gemm(x.reshape(-1, Cc), wp, prec=PREC_BF16X3, bias=b, conv=conv)
    assert _rel(out, ref_cl) < 2e-5
    if Co >= 96:
        scratch = torch.empty(ops.x3_scratch_numel(x.numel() // Cc, Cc), dtype=torch.float32, device=DEV)
        out2 = ops.gemm(x.reshape(-1, Cc), wp, prec=PREC_BF16X3, bias=b, conv=conv, w_split=ops.split_records(wp), x3_scratch=scratch)
        assert _rel(out2, ref_cl) < 2e-5
    # plain bf16 tiles are 64 deep: slice-major weights are refused, not mis-read
    with pytest.raises(Exception):
        ops.gemm(x.reshape(-1, Cc).to(torch.bfloat16), wp.to(torch.bfloat16), prec=PREC_BF16, bias=b, conv=conv)


@pytest.mark.parametrize("F_,H,W_,Cc,relu", [(2, 16, 16, 32, False), (1, 37, 53, 128, True), (3, 518 // 7, 70, 64, True)])
def test_conv3x3_n32_direct(F_, H, W_, Cc, relu):
    """LDS halo-tile direct convolution of the DPT output stage: interior tiles, ragged right / bottom
    tiles, zero padding at every border, several 32-channel slices."""
    x = _rand(F_, H, W_, Cc, seed=30)
    w = _rand(32, Cc, 3, 3, seed=31, scale=1 / math.sqrt(Cc * 9))
    b = _rand(32, seed=32)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)
    if relu:
        ref = F.relu(ref)
    out = ops.conv3x3_n32(x, w, b, relu=relu)
    assert out.shape == (F_, H, W_, 32)
    assert _rel(out, ref.permute(0, 2, 3, 1)) < 2e-5


def test_gemm_pixel_shuffle_convtranspose():
    n, H, W_, Cc, Co, s = 2, 9, 7, 64, 32, 4
    x = _rand(n, H, W_, Cc, seed=30)
    w = _rand(Cc, Co, s, s, seed=31, scale=0.1)          # ConvTranspose2d weight [Cin, Cout, kH, kW]
    b = _rand(Co, seed=32)
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2), w, b, stride=s).permute(0, 2, 3, 1)
    wp = w.permute(2, 3, 1, 0).reshape(s * s * Co, Cc).contiguous()   # [(a,b,co), Cin]
    out = ops.gemm(x.reshape(-1, Cc), wp, bias=b.repeat(s * s), pixel_shuffle=(s, Co, n, H, W_))
    assert _rel(out, ref) < 2e-5


@pytest.mark.parametrize("C", [64, 128, 384, 388, 1024, 2048])
def test_layernorm(C):
    x, g, b = _rand(333, C, seed=40), _rand(C, seed=41), _rand(C, seed=42)
    out = ops.layernorm(x, g, b, 1e-5)
    ref = F.layer_norm(x, (C,), g, b, 1e-5)
    assert (out - ref).abs().max().item() < 2e-5
    assert (ops.layernorm(x, None, None, 1e-6) - F.layer_norm(x, (C,), None, None, 1e-6)).abs().max().item() < 2e-5
    ob = ops.layernorm(x, g, b, 1e-5, out_dtype=torch.bfloat16)
    assert (ob.float() - ref).abs().max().item() < 4e-2


def test_layernorm_concat():
    a, b2 = _rand(100, 1024, seed=43), _rand(100, 1024, seed=44)
    g, b = _rand(2048, seed=45), _rand(2048, seed=46)
    out = ops.layernorm(a, g, b, 1e-5, x2=b2)
    ref = F.layer_norm(torch.cat([a, b2], -1), (2048,), g, b, 1e-5)
    assert (out - ref).abs().max().item() < 2e-5


def _rope_tables(npos, base=100.0):
    # restated from vggt/vggt/layers/rope.py:86-117 (fp32)
    exponents = torch.arange(0, 32, 2).float() / 32
    inv_freq = 1.0 / (base ** exponents)
    ang = torch.einsum("i,j->ij", torch.arange(npos, dtype=inv_freq.dtype), inv_freq)
    return ang.cos().contiguous(), ang.sin().contiguous()


def _rope_ref(t, pos, cos_t, sin_t):
    # t [B, H, N, 64]; pos [B, N, 2]
    def rot(x):
        h = x.shape[-1] // 2
        return torch.cat((-x[..., h:], x[..., :h]), -1)

    def one(x, p):
        c = torch.cat((cos_t, cos_t), -1)[p][:, None]
        s = torch.cat((sin_t, sin_t), -1)[p][:, None]
        return x * c + rot(x) * s

    v, h = t.chunk(2, -1)
    return torch.cat((one(v, pos[..., 0]), one(h, pos[..., 1])), -1)


@pytest.mark.parametrize("tokens,heads", [(200, 4), (201, 3), (5000, 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_qknorm_rope(dtype, tokens, heads):
    qkv = _rand(tokens, 3 * heads * 64, seed=50)
    qw, qb, kw, kb = (_rand(64, seed=51 + i) for i in range(4))
    pos = torch.stack([torch.arange(tokens) % 38, (torch.arange(tokens) * 7) % 38], -1).to(torch.int32)
    cos_t, sin_t = _rope_tables(38)
    x = qkv.to(dtype).float().cpu().reshape(tokens, 3, heads, 64)
    q = F.layer_norm(x[:, 0], (64,), qw.cpu(), qb.cpu(), 1e-5)
    k = F.layer_norm(x[:, 1], (64,), kw.cpu(), kb.cpu(), 1e-5)
    pl = pos.long()[None]
    q = _rope_ref(q.permute(1, 0, 2)[None], pl, cos_t, sin_t)[0].permute(1, 0, 2)
    k = _rope_ref(k.permute(1, 0, 2)[None], pl, cos_t, sin_t)[0].permute(1, 0, 2)
    ref = torch.stack([q, k, x[:, 2]], 1).reshape(tokens, -1)
    buf = qkv.to(dtype).clone()
    ops.qknorm_rope_(buf, heads, qw, qb, kw, kb, 1e-5, pos.to(DEV), cos_t.to(DEV), sin_t.to(DEV))
    # fp32: absolute; bf16: the stored result is rounded to 8 significant bits (half an ulp = 2^-9 relative)
    err = (buf.float().cpu() - ref).abs()
    bound = 2e-5 if dtype == torch.float32 else (5e-3 + 4e-3 * ref.abs())
    assert bool((err <= bound).all())
    # v untouched, bit for bit
    assert torch.equal(buf.reshape(tokens, 3, -1)[:, 2], qkv.to(dtype).reshape(tokens, 3, -1)[:, 2])


@pytest.mark.parametrize("batch,seq,heads,hd", [(2, 77, 3, 64), (1, 300, 2, 64), (3, 8, 4, 128), (2, 81, 8, 48), (1, 1374, 2, 64)])
def test_attention_f32(batch, seq, heads, hd):
    qkv = _rand(batch * seq, 3 * heads * hd, seed=60)
    out = ops.attention(qkv, batch, seq, heads, hd)
    x = qkv.reshape(batch, seq, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(x[0].cpu(), x[1].cpu(), x[2].cpu()).transpose(1, 2).reshape(batch * seq, -1)
    assert (out.cpu() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("batch,seq,heads", [(2, 77, 3), (1, 300, 2), (2, 1374, 4), (1, 2748, 2), (1, 64, 1), (1, 128, 1),
                                              (3, 1, 2), (1, 65, 1), (2, 257, 2), (1, 513, 3)])
def test_attention_bf16(batch, seq, heads):
    hd = 64
    qkv = _rand(batch * seq, 3 * heads * hd, seed=61).to(torch.bfloat16)
    out = ops.attention(qkv, batch, seq, heads, hd)
    x = qkv.float().cpu().reshape(batch, seq, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(x[0], x[1], x[2]).transpose(1, 2).reshape(batch * seq, -1)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 2e-2, err


@pytest.mark.parametrize("variant", ["0"])
def test_attention_bf16_other_kernels(variant):
    """The default is the 64-query kernel; SKIMI_ATTN_Q64=0 (the first, 32-query kernel) stays selectable for
    A/B timing and must stay correct.  The switch is read once per process, so it runs in a child process."""
    import os, subprocess, sys
    code = (
        "import torch, torch.nn.functional as F\n"
        "from skiing_analysis_pytorch_amd import ops\n"
        "worst = 0.0\n"
        "for batch, seq, heads in [(2, 77, 3), (1, 1374, 2), (2, 257, 2), (1, 64, 1)]:\n"
        "    g = torch.Generator().manual_seed(61)\n"
        "    qkv = torch.randn(batch * seq, 3 * heads * 64, generator=g).to(torch.bfloat16).cuda()\n"
        "    out = ops.attention(qkv, batch, seq, heads, 64)\n"
        "    x = qkv.float().cpu().reshape(batch, seq, 3, heads, 64).permute(2, 0, 3, 1, 4)\n"
        "    ref = F.scaled_dot_product_attention(x[0], x[1], x[2]).transpose(1, 2).reshape(batch * seq, -1)\n"
        "    worst = max(worst, (out.float().cpu() - ref).abs().max().item())\n"
        "print('WORST', worst)\n")
    env = dict(os.environ, SKIMI_ATTN_Q64=variant)
    root = str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    worst = float(r.stdout.strip().split("WORST")[-1])
    assert worst < 2e-2, worst


def test_attention_bf16_online_softmax_rescale():
    # force the running max to jump at a late tile: one key row aligned with the queries
    seq, hd = 512, 64
    g = torch.Generator().manual_seed(7)
    q = torch.randn(seq, hd, generator=g) * 0.5
    k = torch.randn(seq, hd, generator=g) * 0.5
    v = torch.randn(seq, hd, generator=g)
    k[400] = q[5] * 8.0     # spike late in the key sequence for query 5
    k[70] = q[300] * 8.0
    qkv = torch.cat([q, k, v], -1).to(torch.bfloat16).to(DEV)
    out = ops.attention(qkv, 1, seq, 1, hd)
    x = qkv.float().cpu()
    ref = F.scaled_dot_product_attention(x[None, None, :, :64], x[None, None, :, 64:128], x[None, None, :, 128:])[0, 0]
    assert (out.float().cpu() - ref).abs().max().item() < 3e-2


@pytest.mark.parametrize("seq,spikes", [(512, ((5, 400, 16.0), (300, 70, 16.0))), (500, ((7, 490, 16.0), (100, 499, 48.0), (450, 64, 24.0))),
                                        (1374, ((0, 1373, 40.0), (1373, 700, 20.0), (640, 1300, 16.0)))])
def test_attention_bf16_reference_moves_late(seq, spikes):
    """The 64-query kernel exponentiates against a standing row reference and looks only at the tile's partial row sum (no row
    maximum after the first tile).  Keys whose score runs 2^40 .. beyond 2^128 (inf) above everything seen before trip that
    test; the tile is then recomputed from LDS and the reference moved -- here in interior tiles, in a ragged last tile (the
    -inf mask is re-applied) and for the very last key, against fp32 SDPA."""
    hd = 64
    g = torch.Generator().manual_seed(seq)
    q = torch.randn(seq, hd, generator=g) * 0.5
    k = torch.randn(seq, hd, generator=g) * 0.5
    v = torch.randn(seq, hd, generator=g)
    for qi, ki, c in spikes:
        k[ki] = q[qi] * c          # score |q|^2 c / 8 ~ 2 c nats = 2.9 c in the kernel's log2 units
    qkv = torch.cat([q, k, v], -1).to(torch.bfloat16).to(DEV)
    out = ops.attention(qkv, 1, seq, 1, hd).float().cpu()
    x = qkv.float().cpu()
    ref = F.scaled_dot_product_attention(x[None, None, :, :64], x[None, None, :, 64:128], x[None, None, :, 128:])[0, 0]
    assert torch.isfinite(out).all()
    # the spiked keys have 16-48 x the norm of the others, so every row's softmax is dominated by a few of them: bf16 q / k / P
    # rounding is worth up to 4e-2 here whichever way the reference moves (the round-2 kernel, which tracked the row maximum,
    # measures 3.9e-2 on the second case)
    assert (out - ref).abs().max().item() < 6e-2
    for qi, ki, c in spikes:       # the spiked rows collapse onto their key's value row
        assert (out[qi] - x[ki, 128:]).abs().max().item() < 3e-2


_GEMM256_VARIANTS = {  # SKIMI_GEMM256_* switches (gemm256_launch): "auto" is what the library picks
    "auto": {},
    "two_phase_192": {"SKIMI_GEMM256_MT3": "1"},
    "ping_pong": {"SKIMI_GEMM256_MT3": "0", "SKIMI_GEMM256_W4": "0"},
    "single_stream": {"SKIMI_GEMM256_MT3": "0", "SKIMI_GEMM256_W4": "1"},
    "two_phase_256": {"SKIMI_GEMM256_MT3": "0", "SKIMI_GEMM256_W4": "0", "SKIMI_GEMM256_PP": "0"},
}


@pytest.mark.parametrize("variant", list(_GEMM256_VARIANTS))
@pytest.mark.parametrize("M,N,K", [(2048, 512, 64), (4096, 1024, 128), (4300, 768, 192), (2100, 768, 1024),
                                   (10992, 1024, 4096),
                                   (16384, 1024, 64), (16384, 1024, 128), (16300, 1000, 192), (16384, 1024, 1024)])
def test_gemm256_lds_dma_path(M, N, K, variant, monkeypatch):
    """bf16 x bf16 plain-row shapes with M >= 2048 take the 256x256 LDS-DMA kernels: the two-phase
    loop (192- or 256-row tiles), the ping-pong loop or the single-stream 4-wave loop.  Every loop
    is run on every shape (K of 1, 2, 3 and many K-tiles; ragged M and N)."""
    for k, v in _GEMM256_VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    a = (_rand(M, K, seed=70)).to(torch.bfloat16)
    w = (_rand(N, K, seed=71, scale=1 / math.sqrt(K))).to(torch.bfloat16)
    b, g, r = _rand(N, seed=72), _rand(N, seed=73), _rand(M, N, seed=74)
    ref = a.float() @ w.float().T
    out = ops.gemm(a, w, prec=PREC_BF16, bias=b, act=ACT_GELU, out_dtype=torch.bfloat16)
    assert _rel(out.float(), F.gelu(ref + b)) < 1e-2
    out = ops.gemm(a, w, prec=PREC_BF16, bias=b, out_dtype=torch.bfloat16)
    assert _rel(out.float(), ref + b) < 1e-2
    out = ops.gemm(a, w, prec=PREC_BF16, bias=b, gamma=g, resid=r)
    assert _rel(out, r + g * (ref + b)) < 1e-5 + 2e-3
    # exact integer data: any fragment / swizzle / row-map error shows as a wrong integer
    ai = ((torch.arange(M * K, device=DEV).reshape(M, K) * 7 + 3) % 9 - 4).to(torch.bfloat16)
    wi = ((torch.arange(N * K, device=DEV).reshape(N, K) * 5 + 1) % 7 - 3).to(torch.bfloat16)
    assert torch.equal(ops.gemm(ai, wi, prec=PREC_BF16), ai.float() @ wi.float().T)


def test_gelu_epilogue_matches_erf_gelu():
    """The packed two-value GELU of the bf16 epilogue (gemm_epilogue.h: gelu_erf2) against erf-GELU
    over the whole input range, through an identity contraction: the output differs from the exact
    value by bf16 rounding only."""
    M, K = 4096, 64
    x = torch.linspace(-9.0, 9.0, M * K, device=DEV).reshape(M, K)
    # one-hot rows pick x[m, n % K] exactly (bf16 inputs: compare against the rounded input)
    a = x.to(torch.bfloat16)
    w = torch.zeros(512, K, device=DEV)
    w[torch.arange(512), torch.arange(512) % K] = 1.0
    out = ops.gemm(a, w.to(torch.bfloat16), prec=PREC_BF16, act=ACT_GELU, bias=torch.zeros(512, device=DEV),
                   out_dtype=torch.bfloat16).float()
    xin = a.float()[:, torch.arange(512, device=DEV) % K]
    ref = F.gelu(xin.double()).float()
    # bf16 rounding of the result: half an ulp = 2^-9 relative
    assert ((out - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-6).all()


@pytest.fixture
def x3_small_shapes(monkeypatch):
    """The LDS-DMA bf16x3 kernels are only picked for launches of >= 160 tiles; let test shapes through."""
    monkeypatch.setenv("SKIMI_X3_MIN_TILES", "1")


def _x3(a, w, **kw):
    """bf16x3 fast path: weights also as pre-split records + scratch for the activation records."""
    conv = kw.get("conv")
    rows, cc = (conv["N"] * conv["H"] * conv["W"], conv["C"]) if conv else a.shape
    scratch = torch.empty(ops.x3_scratch_numel(rows, cc), dtype=torch.float32, device=DEV)
    return ops.gemm(a, w, prec=PREC_BF16X3, w_split=ops.split_records(w), x3_scratch=scratch, **kw)


def test_split_records_reconstructs():
    x = _rand(1000, 200, seed=79) * 7          # ragged last slice: 200 = 6 * 32 + 8
    p = ops.split_records(x)                    # [rows, 7, 2, 32]
    assert p.shape == (1000, 7, 2, 32)
    hi = p[:, :, 0].reshape(1000, -1).float()
    lo = p[:, :, 1].reshape(1000, -1).float()
    assert torch.equal(hi[:, :200], x.to(torch.bfloat16).float())
    assert (((hi + lo)[:, :200] - x).abs() / x.abs().clamp_min(1e-20)).max().item() < 2 ** -15
    assert (hi[:, 200:] == 0).all() and (lo[:, 200:] == 0).all()


def test_split_planes_reconstructs():
    x = _rand(1000, 256, seed=80) * 7
    p = ops.split_planes(x)
    rec = p[0].float() + p[1].float()
    assert ((rec - x).abs() / x.abs().clamp_min(1e-20)).max().item() < 2 ** -15
    assert torch.equal(p[0], x.to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(4096, 256, 256), (5000, 128, 1024), (10952, 256, 2048), (4100, 512, 392), (4200, 128, 32),
                                   (4300, 96, 64), (4097, 128, 96), (4500, 1024, 128), (4096, 256, 32), (4096, 256, 64)])
def test_gemm_x3dma_plain(M, N, K, x3_small_shapes):
    a, w, b, r = _rand(M, K, seed=81), _rand(N, K, seed=82, scale=1 / math.sqrt(K)), _rand(N, seed=83), _rand(M, N, seed=84)
    ref = a @ w.T + b
    assert _rel(_x3(a, w, bias=b), ref) < 2e-5
    assert _rel(_x3(a, w, bias=b, act=ACT_RELU, resid=r), F.relu(ref) + r) < 2e-5
    ai = torch.arange(M * K, device=DEV, dtype=torch.float32).reshape(M, K) % 13 - 6
    wi = (torch.arange(N * K, device=DEV, dtype=torch.float32).reshape(N, K) * 3 % 11) - 5
    assert torch.equal(_x3(ai, wi), ai @ wi.T)


@pytest.mark.parametrize("cfg", [dict(H=74, W=74, C=64, Co=256, k=3, s=1, p=1), dict(H=75, W=73, C=128, Co=128, k=3, s=2, p=1),
                                 dict(H=70, W=70, C=32, Co=256, k=3, s=1, p=1), dict(H=80, W=61, C=64, Co=128, k=3, s=1, p=1),
                                 dict(H=66, W=70, C=32, Co=100, k=3, s=1, p=1)])
def test_gemm_x3dma_conv(cfg, x3_small_shapes):
    H, W_, Cc, Co, k, s, p = (cfg[x] for x in ("H", "W", "C", "Co", "k", "s", "p"))
    n = 2
    x = _rand(n, H, W_, Cc, seed=85)
    w = _rand(Co, Cc, k, k, seed=86, scale=1 / math.sqrt(Cc * k * k))
    b = _rand(Co, seed=87)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=s, padding=p)
    OH, OW = ref.shape[2], ref.shape[3]
    wp = w.permute(0, 2, 3, 1).reshape(Co, k * k * Cc).contiguous()
    conv = dict(N=n, H=H, W=W_, C=Cc, KH=k, KW=k, stride=s, pad=p, dil=1, OH=OH, OW=OW)
    out = _x3(x.reshape(-1, Cc), wp, bias=b, conv=conv)
    assert out.shape[0] >= 2048
    assert _rel(out, ref.permute(0, 2, 3, 1).reshape(-1, Co)) < 2e-5


@pytest.mark.parametrize("Co", [256, 128])
def test_gemm_records_chain(Co, x3_small_shapes):
    """conv -> conv with the intermediate handed over as bf16x3 records written by the first conv's epilogue
    (out_records, with and without the fp32 copy) instead of fp32 + a split pass: same results."""
    n, H, W_, Cc = 2, 70, 65, 64           # 9100 rows: interior tiles + a ragged last one
    x = _rand(n, H, W_, Cc, seed=91)
    w1 = _rand(Co, Cc, 3, 3, seed=92, scale=1 / math.sqrt(Cc * 9))
    w2 = _rand(256, Co, 3, 3, seed=93, scale=1 / math.sqrt(Co * 9))
    b1, b2 = _rand(Co, seed=94), _rand(256, seed=95)
    r = _rand(n * H * W_, Co, seed=96)
    ref1 = F.relu(F.relu(F.conv2d(x.permute(0, 3, 1, 2), w1, b1, padding=1)).permute(0, 2, 3, 1).reshape(-1, Co) + r)
    ref2 = F.conv2d(ref1.reshape(n, H, W_, Co).permute(0, 3, 1, 2), w2, b2, padding=1).permute(0, 2, 3, 1).reshape(-1, 256)
    pack = lambda w: w.reshape(w.shape[0], w.shape[1] // 32, 32, 3, 3).permute(0, 1, 3, 4, 2).reshape(w.shape[0], -1).contiguous()
    wp1, wp2 = pack(w1), pack(w2)
    conv1 = dict(N=n, H=H, W=W_, C=Cc, KH=3, KW=3, stride=1, pad=1, dil=1, OH=H, OW=W_, slice_major=True)
    conv2 = dict(conv1, C=Co)
    from skiing_analysis_pytorch_amd._lib import ACT_RELU as RELU
    for keep_fp32 in (True, False):
        rec = ops.records_buffer(n * H * W_, Co)
        rec.fill_(float("nan"))
        sc = torch.empty(ops.x3_scratch_numel(n * H * W_, Cc), dtype=torch.float32, device=DEV)
        out1 = torch.empty(n * H * W_, Co, device=DEV) if keep_fp32 else None
        y1 = ops.gemm(x.reshape(-1, Cc), wp1, prec=PREC_BF16X3, bias=b1, act=RELU, resid=r, post_act=RELU, conv=conv1,
                      w_split=ops.split_records(wp1), x3_scratch=sc, out=out1, out_records=rec, records_only=not keep_fp32)
        if keep_fp32:
            assert _rel(y1, ref1) < 2e-5
            assert torch.equal(rec[:-128].reshape(-1, Co // 32, 2, 32), ops.split_records(y1))
        assert (rec[-128:] == 0).all()
        y2 = ops.gemm(None, wp2, prec=PREC_BF16X3, bias=b2, conv=conv2, w_split=ops.split_records(wp2), a_records=rec)
        assert _rel(y2, ref2) < 3e-5


def test_gemm_x3dma_pixel_shuffle(x3_small_shapes):
    n, H, W_, Cc, Co, s = 3, 37, 37, 256, 256, 2
    x = _rand(n, H, W_, Cc, seed=88)
    w = _rand(Cc, Co, s, s, seed=89, scale=0.1)
    b = _rand(Co, seed=90)
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2), w, b, stride=s).permute(0, 2, 3, 1)
    wp = w.permute(2, 3, 1, 0).reshape(s * s * Co, Cc).contiguous()
    out = _x3(x.reshape(-1, Cc), wp, bias=b.repeat(s * s), pixel_shuffle=(s, Co, n, H, W_))
    assert _rel(out, ref) < 2e-5


@pytest.mark.parametrize("prec", [PREC_BF16X3, PREC_BF16])
def test_gemm_narrow_n_tile(prec):
    """N <= 64 with many rows takes the 128 x 64 tile (the DPT 128 -> 32 output conv)."""
    n, H, W_, Cc, Co = 2, 120, 130, 128, 32
    x = _rand(n, H, W_, Cc, seed=95)
    w = _rand(Co, Cc, 3, 3, seed=96, scale=1 / math.sqrt(Cc * 9))
    b = _rand(Co, seed=97)
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)).permute(0, 2, 3, 1).reshape(-1, Co)
    wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Cc).contiguous()
    if prec == PREC_BF16:
        wp = wp.to(torch.bfloat16)
    conv = dict(N=n, H=H, W=W_, C=Cc, KH=3, KW=3, stride=1, pad=1, dil=1, OH=H, OW=W_)
    out = ops.gemm(x.reshape(-1, Cc), wp, prec=prec, bias=b, act=ACT_RELU, conv=conv)
    assert _rel(out, ref) < (2e-5 if prec == PREC_BF16X3 else 1e-2)
