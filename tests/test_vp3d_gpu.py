"""GPU parity of the HIP TemporalModel lifter through the C-ABI against the oracle and the
reference's own outputs (tests/golden/vp3d_*.npz)."""
import numpy as np
import pytest
import torch

from oracle import vp3d_oracle
from skiing_analysis_pytorch_amd import vp3d, weights as W
from skiing_analysis_pytorch_amd._lib import PREC_BF16, PREC_BF16X3

pytestmark = pytest.mark.gpu

TOL_JOINTS = 1e-3   # north_star: output 3D joints within 1e-3 of the reference CPU path


def _model(fw, causal, prec):
    m = vp3d.TemporalModel(17, 2, 17, fw, causal=causal, channels=1024, prec=prec)
    m.load_state_dict(W.make_vp3d_state_dict(seed=0, filter_widths=fw))
    return m


@pytest.mark.parametrize("name", ["rf27", "rf27_causal", "rf243", "rf81_w5"])
def test_vp3d_matches_reference_golden(golden_dir, name):
    g = np.load(golden_dir / f"vp3d_{name}.npz")
    fw = [int(v) for v in g["filter_widths"]]
    causal = bool(g["causal"])
    m = _model(fw, causal, PREC_BF16X3)
    assert m.receptive_field() == int(g["receptive_field"])
    for aug in (0, 1):
        x = torch.from_numpy(g[f"batch2d_aug{aug}"]).cuda()
        raw = m(x).cpu().numpy()
        assert raw.shape == g[f"raw_aug{aug}"].shape
        err = np.abs(raw - g[f"raw_aug{aug}"]).max()
        assert err < 2e-4, f"{name} aug{aug}: max abs err {err}"
        pred = vp3d.lift_clip(m, g["keypoints_px"], int(g["w"]), int(g["h"]), augment=bool(aug))
        mp = vp3d_oracle.mpjpe(pred, g[f"pred_aug{aug}"])
        assert mp < TOL_JOINTS and np.abs(pred - g[f"pred_aug{aug}"]).max() < TOL_JOINTS


def test_vp3d_vs_oracle_other_seed():
    fw = [3, 3, 3]
    sd = W.make_vp3d_state_dict(seed=5, filter_widths=fw)
    m = vp3d.TemporalModel(17, 2, 17, fw, prec=PREC_BF16X3)
    m.load_state_dict(sd)
    for frames in (1, 27, 100):        # ragged clip lengths incl. a single frame
        kp = W.make_keypoints_2d(frames=frames, seed=9).numpy()
        ref = vp3d_oracle.lift_clip(sd, kp, 1920, 1080, fw)
        out = vp3d.lift_clip(m, kp, 1920, 1080)
        assert out.shape == (frames, 17, 3)
        assert np.abs(out - ref).max() < 2e-4


def test_vp3d_bf16_mode_tolerance():
    # plain bf16 operands: documents what the fast mode costs (not the parity mode)
    fw = [3, 3, 3]
    sd = W.make_vp3d_state_dict(seed=0, filter_widths=fw)
    m = vp3d.TemporalModel(17, 2, 17, fw, prec=PREC_BF16)
    m.load_state_dict(sd)
    kp = W.make_keypoints_2d(frames=243, seed=1).numpy()
    ref = vp3d_oracle.lift_clip(sd, kp, 1920, 1080, fw)
    out = vp3d.lift_clip(m, kp, 1920, 1080)
    rel = np.linalg.norm(out - ref) / np.linalg.norm(ref)
    assert rel < 3e-2, rel


@pytest.mark.parametrize("fw,causal", [([3, 3, 3], False), ([3, 3, 3, 3, 3], False), ([3, 3, 3], True)])
def test_vp3d_large_batch_takes_the_lds_dma_kernels(fw, causal):
    """From a few dozen clips per call the block convolutions run on the LDS-DMA bf16x3 kernels (dilated 1-D
    gather, tap-major K, row-remapped residual in the 1x1): every clip of a 64-clip batch equals its
    single-clip result."""
    m = _model(fw, causal, PREC_BF16X3)
    rf = m.receptive_field()
    frames = rf + 216
    x = torch.randn(64, frames, 17, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    big = m(x)
    assert big.shape == (64, frames - rf + 1, 17, 3)
    for i in (0, 17, 63):
        one = m(x[i:i + 1].contiguous())
        # different accumulation orders (LDS-DMA kernel vs split-K atomics) on outputs of magnitude ~20
        assert (big[i] - one[0]).abs().max().item() < 5e-5 * one[0].abs().max().item()


def test_vp3d_out_buffer_reuse():
    """`out=` (beyond the reference signature) writes into a caller buffer; shape errors are loud."""
    fw = [3, 3, 3]
    m = _model(fw, False, PREC_BF16X3)
    x = torch.randn(2, 60, 17, 2, device="cuda")
    out = torch.empty(2, 60 - 26, 17, 3, device="cuda")
    r = m(x, out=out)
    assert r.data_ptr() == out.data_ptr()
    assert (out.cpu() - m(x).cpu()).abs().max().item() < 2e-4   # split-K float atomics: last bits vary
    with pytest.raises(Exception):
        m(x, out=torch.empty(2, 10, 17, 3, device="cuda"))


def test_vp3d_short_input_rejected():
    from skiing_analysis_pytorch_amd import _lib

    m = _model([3, 3, 3], False, PREC_BF16X3)
    with pytest.raises(_lib.SkimiError):
        m(torch.zeros(1, 26, 17, 2, device="cuda"))


@pytest.mark.parametrize("fw,causal", [([3, 3, 3], False), ([3, 3, 3, 3, 3], False), ([3, 3, 3], True), ([3, 5, 3], False)])
def test_vp3d_small_batches_on_the_streaming_path(fw, causal, monkeypatch):
    """B = 1 .. a few clips run on the weight-streaming kernels (vp3d_stream.hip: one launch per convolution, row splits
    per clip, activations loaded once for all taps; filter width 5 and the large dilations of RF 243 take the per-tap
    gather kernel): every clip of a small batch equals its single-clip result, both agree with the oracle, and the
    generic GEMM chain (SKIMI_VP3D_STREAM=0 in a fresh handle's process is not possible here, so the oracle is the
    common reference).  The streaming path has no atomics: two runs are bit-identical."""
    from oracle import vp3d_oracle

    sd = W.make_vp3d_state_dict(seed=0, filter_widths=fw)
    m = vp3d.TemporalModel(17, 2, 17, fw, causal=causal, prec=PREC_BF16X3)
    m.load_state_dict(sd)
    rf = m.receptive_field()
    # B = 2 (the reference's flip-TTA call), 4, 6: the dilated convs of launches of 257 .. 512 workgroups run as co-resident
    # pairs of 4-wave workgroups (vp3d_stream.hip: NW = 4); 5 row tiles per workgroup for the 1 x 1 convs
    for B, frames in ((1, rf), (1, rf + 242), (2, rf + 242), (2, rf + 57), (3, rf + 100), (4, rf + 242), (5, rf + 17), (6, rf + 130)):
        x = torch.randn(B, frames, 17, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(B * 7 + frames))
        out = m(x)
        assert out.shape == (B, frames - rf + 1, 17, 3)
        assert torch.equal(out, m(x))                                      # deterministic
        with torch.no_grad():
            ref = vp3d_oracle.temporal_model_forward(sd, x.cpu(), fw, causal)
        assert (out.cpu() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
        for i in range(B):
            one = m(x[i:i + 1].contiguous())
            assert (out[i] - one[0]).abs().max().item() < 1e-5 * max(1.0, one.abs().max().item())
