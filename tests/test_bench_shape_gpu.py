"""Parity AT THE BENCHMARKED SHAPE: VGGT-1B, S = 8 views x 518 x 518 (global attention over 10 992
tokens), B = 2 time steps per call (batch > 1 and the long sequence both live), all four heads
(17 query points per step: BASELINE config 3), synthetic weights shared with the fp32 CPU oracle.

  * PREC_BF16X3 everywhere (the parity mode): every output <= 1e-3 against the oracle, and the MPJPE
    (VideoPose3D/common/loss.py:11-17) of the 8-view DLT joints from the HIP cameras against the joints
    from the oracle's cameras on identical 2D keypoints <= 1e-3.
  * PREC_BF16 aggregator (the benchmark mode = the reference's own GPU autocast precision,
    vggt/vggt/infer.py:78-84): the same MPJPE is measured and bounded; it does NOT meet 1e-3 (bf16
    operands put ~5e-3 on pose_enc, which the synthetic model's cameras -- FoV > pi, a 0.9-unit
    baseline seen from 3 units -- amplify ~10x), which is why bench.py reports both modes.

The oracle runs once per module (two S = 8 steps, about a minute and a half on the GPU box's 16 cores).
"""
import os

import numpy as np
import pytest
import torch

from oracle import joints_check, vggt_oracle
from skiing_analysis_pytorch_amd import geometry, vggt, weights as W
from skiing_analysis_pytorch_amd._lib import PREC_BF16, PREC_BF16X3, PREC_F16

pytestmark = pytest.mark.gpu

B, S, IMG, NQ = 2, 8, 518, 17


@pytest.fixture(scope="module")
def bench_shape():
    cfg = W.VGGTConfig()                                    # VGGT-1B, all heads
    sd = W.make_vggt_state_dict(cfg, seed=0, device="cuda")
    cpu_sd = {k: v.cpu() for k, v in sd.items()}
    g = torch.Generator().manual_seed(1234)
    images = torch.rand((B, S, 3, IMG, IMG), generator=g)
    queries = torch.rand((B, NQ, 2), generator=g) * (IMG - 80) + 40      # view-0 keypoints in 518-space
    try:
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        pass
    with torch.no_grad():
        ref = vggt_oracle.vggt_forward(cpu_sd, images, cfg.to_dict(), query_points=queries)
    kps, Xw, joints_ref = joints_check.keypoints_from_oracle_cameras(ref["pose_enc"], (IMG, IMG), joints=17, seed=5)
    assert joints_check.conditioning_error(Xw, joints_ref) < 1e-4      # the DLT systems are well conditioned
    # the ring rig (oracle/joints_check.py): the same metric on a well-conditioned standard scene -- the reference
    # cameras are a ring of 8 around the skier, the cameras under test are ring + (pose_enc - oracle's pose_enc)
    ring, kps_ring, joints_ring = joints_check.ring_rig_scene(ref["pose_enc"], (IMG, IMG), joints=17, seed=5)
    return dict(cfg=cfg, sd=sd, images=images, queries=queries, ref=ref, kps=kps, joints_ref=joints_ref,
                ring=ring, kps_ring=kps_ring, joints_ring=joints_ring)


def _ring_mpjpe(out, s):
    """MPJPE on the ring rig, cameras and DLT through the product's device geometry"""
    pe = joints_check.ring_rig_test_pose_enc(s["ring"], out["pose_enc"], s["ref"]["pose_enc"]).cuda()
    return joints_check.mpjpe(_joints({"pose_enc": pe}, s["kps_ring"]), s["joints_ring"])


def _joints(out, kps):
    E, K = geometry.pose_encoding_to_extri_intri(out["pose_enc"], (IMG, IMG))
    return geometry.triangulate_joints(K, E[..., :3, :3].contiguous(), E[..., :3, 3].contiguous(), kps.cuda()).cpu().numpy()


def _rel(got, ref):
    return ((got.cpu() - ref).abs() / (ref.abs() + 1.0)).max().item()


def test_parity_mode_every_output_at_bench_shape(bench_shape):
    s = bench_shape
    m = vggt.VGGT(config=s["cfg"], prec=PREC_BF16X3, head_prec=PREC_BF16X3)
    m.load_state_dict(s["sd"])
    out = m(s["images"].cuda(), query_points=s["queries"].cuda())
    torch.cuda.synchronize()
    # run-to-run determinism of the parity mode: the K splits of its skinny contractions are summed
    # in a fixed order (one slab plane per split, gemm.hip), no float atomics anywhere -> a second run gives the same bits
    keys = ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf", "track", "vis", "conf")
    first = {k: out[k].clone() for k in keys}
    out = m(s["images"].cuda(), query_points=s["queries"].cuda())
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(first[k], out[k]), f"parity mode: {k} differs between two runs on the same inputs"
    ref = s["ref"]
    for i in range(4):
        assert (out["pose_enc_list"][i].cpu() - ref["pose_enc_list"][i]).abs().max().item() < 1e-3
    for k in ("depth", "depth_conf", "world_points", "world_points_conf"):
        assert out[k].shape == ref[k].shape
        assert _rel(out[k], ref[k]) < 1e-3, k
    # the second time step of the batch is its own scene: it must not be a copy of the first
    assert (out["pose_enc"][0] - out["pose_enc"][1]).abs().max().item() > 1e-4
    # track head at full size (7 correlation levels, 4 iterations, 6 + 18 blocks, 64 virtual tracks,
    # 17 queries); tolerance in pixels as for the tiny golden (test_vggt_gpu.py)
    assert out["track"].shape == ref["track"].shape == (B, S, NQ, 2)
    # Tolerance in pixels, and where it comes from (profiles/r02_track_sensitivity.json, tools/track_sensitivity.py):
    # the fp32 oracle's own tracks move by up to 7e-4 px when every input pixel moves by one fp32 ulp (6e-8
    # relative), i.e. the refinement loop (flow embedded at up to 1000 rad/px, utils.py:107; 9x9 correlation
    # windows re-sampled at the moving estimate) amplifies relative error by ~1e4 px.  The parity mode's MFMA
    # products are bf16x3 (hi*hi + hi*lo + lo*hi, relative error 2^-17 = 8e-6 per product, against fp32's
    # 6e-8), so ~1e-2 .. 1e-1 px on the worst track is what this arithmetic can deliver; the bulk sits
    # near 1e-3 px.
    dtr = (out["track"].cpu() - ref["track"]).abs()
    print(f"track px err: median {dtr.median().item():.2e}, p99 {dtr.flatten().kthvalue(int(0.99 * dtr.numel())).values.item():.2e}, max {dtr.max().item():.2e}")
    # measured on MI355X: median 2.0e-3 .. 2.5e-3, p99 5e-2, max 0.16 px; vis / conf (sigmoid of the refined features):
    # median 4e-4 / 7e-4, worst element 8.7e-3 / 4.2e-2 -- the same on every run since round 3 (ordered split-K); the
    # worst elements sit on tracks whose refinement amplifies the bf16x3 product error (see above), not on an
    # arithmetic that changes from run to run
    assert dtr.median().item() < 5e-3 and dtr.max().item() < 0.5
    for k in ("vis", "conf"):
        dv = (out[k].cpu() - ref[k]).abs()
        print(f"{k} abs err: median {dv.median().item():.2e}, max {dv.max().item():.2e}")
        assert dv.median().item() < 2e-3 and dv.max().item() < 6e-2, k
    assert torch.allclose(out["track"][:, 0].cpu(), s["queries"], atol=1e-4)
    # the metric's quantity: 3D joints
    e = joints_check.mpjpe(_joints(out, s["kps"]), s["joints_ref"])
    er = _ring_mpjpe(out, s)
    print(f"parity mode (bf16x3): MPJPE of the DLT joints vs the CPU oracle = {e:.3e} (native scene), {er:.3e} (ring rig)")
    assert e < 1e-3 and er < 1e-4


def test_bench_mode_joints_mpjpe_at_bench_shape(bench_shape):
    s = bench_shape
    m = vggt.VGGT(config=s["cfg"], prec=PREC_BF16, head_prec=PREC_BF16X3)
    m.load_state_dict(s["sd"])
    out = m(s["images"].cuda(), query_points=s["queries"].cuda())
    torch.cuda.synchronize()
    ref = s["ref"]
    for k in ("pose_enc", "depth", "world_points", "track"):
        assert torch.isfinite(out[k]).all(), k
    pe = (out["pose_enc"].cpu() - ref["pose_enc"]).abs().max().item()
    e = joints_check.mpjpe(_joints(out, s["kps"]), s["joints_ref"])
    scale = float(np.abs(s["joints_ref"]).max())
    print(f"bench mode (bf16 aggregator): pose_enc max abs err {pe:.3e}, joints MPJPE {e:.3e} (scene scale {scale:.2f})")
    er = _ring_mpjpe(out, s)
    print(f"bench mode (bf16 aggregator): ring-rig MPJPE {er:.3e}")
    # bf16 operands: 5e-3 .. 8e-3 on pose_enc (measured 7.2e-3; emulated 7.7e-3, profiles/r03_precision_ablation.md);
    # NOT within the 1e-3 bar on either scene.  On the ring rig a pose_enc error of size d moves the joints by 0.3 d .. 0.6 d
    # (measured: 0.56 d for uniform errors, 0.36 d for this mode's), so the bound follows from the measured pose error;
    # the native scene (FoV > pi, clustered cameras) amplifies the same error ~20x and only gets the finiteness bound.
    assert 2e-3 < pe < 1.5e-2
    assert er < 1.0 * pe and er > 1e-3, (er, pe)
    assert e < 1.0
    # dense outputs keep bf16-level agreement at this size too (an indexing bug would show as O(1))
    rel = (out["depth"].cpu() - ref["depth"]).abs() / (ref["depth"].abs() + 1.0)
    assert rel.median().item() < 2e-2
    # a tracked point moves with the features: bf16 features shift tracks by a fraction of a pixel
    assert (out["track"].cpu() - ref["track"]).abs().median().item() < 2.0


def test_f16_mode_meets_the_joints_bar_at_bench_shape(bench_shape):
    """PREC_F16 (fp16 operands in the Linears of the 72 blocks and the patch embedding, bf16 attention products, fp32
    everything else; heads fp32-accurate): the mode bench.py's headline runs.  fp16's three extra mantissa bits put
    pose_enc within ~1e-3 of the fp32 oracle (emulated: 7.7e-4, profiles/r03_precision_ablation.md) and the joints
    inside north_star's 1e-3 on the ring rig; the matrix rate is bf16's."""
    s = bench_shape
    m = vggt.VGGT(config=s["cfg"], prec=PREC_F16, head_prec=PREC_BF16X3)
    m.load_state_dict(s["sd"])
    out = m(s["images"].cuda(), query_points=s["queries"].cuda())
    torch.cuda.synchronize()
    ref = s["ref"]
    for k in ("pose_enc", "depth", "world_points", "track"):
        assert torch.isfinite(out[k]).all(), k
    pe = (out["pose_enc"].cpu() - ref["pose_enc"]).abs().max().item()
    er = _ring_mpjpe(out, s)
    en = joints_check.mpjpe(_joints(out, s["kps"]), s["joints_ref"])
    rel = (out["depth"].cpu() - ref["depth"]).abs() / (ref["depth"].abs() + 1.0)
    relp = (out["world_points"].cpu() - ref["world_points"]).abs() / (ref["world_points"].abs() + 1.0)
    dtr = (out["track"].cpu() - ref["track"]).abs()
    print(f"f16 mode: pose_enc max abs err {pe:.3e}, joints MPJPE ring rig {er:.3e} / native scene {en:.3e}, depth rel err median "
          f"{rel.median().item():.2e} max {rel.max().item():.2e}, points median {relp.median().item():.2e}, track px median {dtr.median().item():.2e}")
    assert pe < 2e-3
    assert er < 1e-3                      # north_star's bar on the 3D joints
    assert rel.median().item() < 3e-3 and relp.median().item() < 3e-3
    assert dtr.median().item() < 0.5
    assert (out["pose_enc"][0] - out["pose_enc"][1]).abs().max().item() > 1e-4
    # the cameras (hence the joints) of the headline mode are run-to-run deterministic: no split-K atomics on their path
    # (256-row loops in the blocks at this size, ordered split-K in the fp32-accurate camera head); the dense maps too
    again = m(s["images"].cuda(), want={"camera", "depth"})
    assert torch.equal(again["pose_enc"], out["pose_enc"]) and torch.equal(again["depth"], out["depth"])
    # against the bf16 mode on the same inputs: the pose error drops by the 8x the formats differ by (loosely: > 3x)
    m16 = vggt.VGGT(config=s["cfg"], prec=PREC_BF16, head_prec=PREC_BF16X3)
    m16.load_state_dict(s["sd"])
    o16 = m16(s["images"].cuda(), want={"camera"})
    pe16 = (o16["pose_enc"].cpu() - ref["pose_enc"]).abs().max().item()
    assert pe < pe16 / 3, (pe, pe16)


def test_f16_heads_option_at_bench_shape(bench_shape):
    """head_prec = PREC_F16: the depth / point DPT heads on fp16 operands and activations (one MFMA per product instead of
    three) -- an OPTION, not what bench.py's headline runs: the reference computes these heads in fp32 (vggt.py:65), and
    fp16 convolutions leave the dense maps at ~1e-3 worst-case relative error (emulated: profiles/r03_head_precision.json;
    measured here).  The camera head stays fp32-accurate, so pose_enc and the joints are those of the default heads."""
    s = bench_shape
    m = vggt.VGGT(config=s["cfg"], prec=PREC_F16, head_prec=PREC_F16)
    m.load_state_dict(s["sd"])
    out = m(s["images"].cuda(), want={"camera", "depth", "point"})
    torch.cuda.synchronize()
    ref = s["ref"]
    pe = (out["pose_enc"].cpu() - ref["pose_enc"]).abs().max().item()
    msg = [f"f16 heads: pose_enc max abs err {pe:.3e}"]
    for k in ("depth", "depth_conf", "world_points", "world_points_conf"):
        assert out[k].shape == ref[k].shape and torch.isfinite(out[k]).all(), k
        rel = (out[k].cpu() - ref[k]).abs() / (ref[k].abs() + 1.0)
        msg.append(f"{k} rel err median {rel.median().item():.2e} max {rel.max().item():.2e}")
        assert rel.median().item() < 1e-3 and rel.max().item() < 1e-2, k
    print("; ".join(msg))
    assert pe < 2e-3 and _ring_mpjpe(out, s) < 1e-3


def test_fp8_mode_forward_at_bench_shape(bench_shape):
    """BASELINE config 5 (VGGT with e4m3 weights): SKIMI_PREC_FP8 -- MXFP8 qkv / fc1 / fc2 in the 72 blocks -- at VGGT-1B
    size against the fp32 oracle.  The reference has no fp8 path (parity unpinned: the quantiser is pinned against the
    OCP formats in test_fp8_gpu.py); this bounds what e4m3's 3 mantissa bits (16x bf16's rounding) do to the outputs:
    pose_enc ~8e-2 (measured; bf16 5e-3, fp16 7e-4), NOT within the 1e-3 joints bar -- reported separately by bench.py
    as SURVEY 8(d) asks."""
    from skiing_analysis_pytorch_amd._lib import PREC_FP8

    s = bench_shape
    m = vggt.VGGT(config=s["cfg"], prec=PREC_FP8, head_prec=PREC_BF16X3)
    m.load_state_dict(s["sd"])
    out = m(s["images"].cuda(), query_points=s["queries"].cuda())
    torch.cuda.synchronize()
    ref = s["ref"]
    for k in ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf", "track", "vis", "conf"):
        assert out[k].shape == ref[k].shape and torch.isfinite(out[k]).all(), k
    pe = (out["pose_enc"].cpu() - ref["pose_enc"]).abs().max().item()
    rel = (out["depth"].cpu() - ref["depth"]).abs() / (ref["depth"].abs() + 1.0)
    er = _ring_mpjpe(out, s)
    print(f"fp8 mode: pose_enc max abs err {pe:.3e}, ring-rig MPJPE {er:.3e}, depth rel err median {rel.median().item():.2e}")
    assert pe < 0.3 and rel.median().item() < 1e-2 and er < 0.3
    assert (out["pose_enc"][0] - out["pose_enc"][1]).abs().max().item() > 1e-4
    assert (out["track"].cpu() - ref["track"]).abs().median().item() < 2.0


@pytest.mark.parametrize("S_sv", [8, 16])
def test_single_view_clip_config2(bench_shape, S_sv):
    """BASELINE config 2 (`single_view_process` semantics: every 30th frame of ONE camera forms a single
    S = ceil(T / 30) view stack, vggt/single_view_process.py:130-163) at 518 x 518, S = 8 and 16, through
    infer.process_single_view_clip, against the oracle's cameras.  S = 16 runs global attention over
    21 984 tokens -- the longest sequence any config asks for."""
    from skiing_analysis_pytorch_amd import infer

    s = bench_shape
    if S_sv == 8:
        sel, ref_pe = s["images"][0], s["ref"]["pose_enc"][0]
    else:
        sel = torch.rand((S_sv, 3, IMG, IMG), generator=torch.Generator().manual_seed(77))
        cfgd = W.VGGTConfig(enable_depth=False, enable_point=False, enable_track=False).to_dict()
        cpu_sd = {k: v.cpu() for k, v in s["sd"].items() if not k.startswith(("depth_head", "point_head", "track_head"))}
        with torch.no_grad():
            ref_pe = vggt_oracle.vggt_forward(cpu_sd, sel[None], cfgd)["pose_enc"][0]
    T = 30 * (S_sv - 1) + 5
    frames = torch.zeros((T, 3, IMG, IMG), device="cuda")
    frames[::30] = sel.cuda()
    m = vggt.VGGT(config=s["cfg"], prec=PREC_BF16X3, head_prec=PREC_BF16X3)
    m.load_state_dict(s["sd"])
    out = infer.process_single_view_clip(m, frames, every=30)
    assert out["pose_enc"].shape == (S_sv, 9) and out["extrinsic"].shape == (S_sv, 3, 4) and out["intrinsic"].shape == (S_sv, 3, 3)
    assert (out["pose_enc"].cpu() - ref_pe).abs().max().item() < 1e-3
    E, K = vggt_oracle.pose_encoding_to_extri_intri(ref_pe[None], (IMG, IMG))
    assert (out["extrinsic"].cpu() - E[0]).abs().max().item() < 1e-3
    assert ((out["intrinsic"].cpu() - K[0]).abs() / (K[0].abs() + 1)).max().item() < 1e-3
    # the benchmark precision on the same stack: finite, bf16-close
    m16 = vggt.VGGT(config=s["cfg"], prec=PREC_BF16, head_prec=PREC_BF16X3)
    m16.load_state_dict(s["sd"])
    o16 = infer.process_single_view_clip(m16, frames, every=30)
    assert torch.isfinite(o16["pose_enc"]).all() and (o16["pose_enc"].cpu() - ref_pe).abs().max().item() < 3e-2
    # and the fp16 mode (bench.py's headline): an order of magnitude closer on the same stack (global attention over
    # up to 21 984 tokens with fp16 Linears)
    mh = vggt.VGGT(config=s["cfg"], prec=PREC_F16, head_prec=PREC_BF16X3)
    mh.load_state_dict(s["sd"])
    oh = infer.process_single_view_clip(mh, frames, every=30)
    eh = (oh["pose_enc"].cpu() - ref_pe).abs().max().item()
    print(f"config 2, S = {S_sv}: pose_enc max abs err fp16 {eh:.2e}, bf16 {(o16['pose_enc'].cpu() - ref_pe).abs().max().item():.2e}")
    assert eh < 3e-3


def test_bench_batch_of_four_equals_its_time_steps(bench_shape):
    """The bench's own call shape -- B = 4 time steps x 8 views per call (M = 43 968 token rows, 2752-workgroup attention
    launches), benchmark mode, all heads -- against the same four time steps run one per call: the time steps of a
    batch are independent (aggregator.py:184-258 never mixes them), so any difference beyond bf16 arithmetic noise
    (other tile shapes / split-K orders at M / 4) is a batch-indexing bug at the benchmarked size.  The S = 8 oracle
    comparison above pins what one time step computes."""
    s = bench_shape
    m = vggt.VGGT(config=s["cfg"], prec=PREC_BF16, head_prec=PREC_BF16X3)
    m.load_state_dict(s["sd"])
    g = torch.Generator().manual_seed(77)
    images = torch.rand((4, S, 3, IMG, IMG), generator=g).cuda()
    queries = (torch.rand((4, NQ, 2), generator=g) * (IMG - 80) + 40).cuda()
    want = {"camera", "depth", "point", "track"}
    whole = m(images, query_points=queries, want=want)
    whole = {k: v.cpu() for k, v in whole.items() if torch.is_tensor(v)}
    for b in range(4):
        one = m(images[b:b + 1], query_points=queries[b:b + 1], want=want)
        for k in ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf", "track", "vis", "conf"):
            a, r = whole[k][b:b + 1], one[k].cpu()
            assert a.shape == r.shape, k
            assert torch.isfinite(a).all(), k
            err = ((a - r).abs() / (r.abs() + 1.0))
            # M = 43 968 and M = 10 992 take different tile orders / loop variants of the bf16 GEMMs, so the tokens differ at
            # bf16 level: observed pose_enc 2.6e-3, depth 3e-4, points 1e-3, track 1.2e-3 px-relative, vis / conf 3e-2 max and
            # 2e-3 median; an indexing bug gives O(1)
            print(f"step {b} {k}: max {err.max().item():.3e} median {err.median().item():.3e}")
            # the track head runs at the aggregator's 16-bit precision in this mode (as the reference's autocast does,
            # models/vggt.py:85-91): its visibility / confidence logits carry bf16 operand noise of other tile orders
            # (observed: vis / conf 3e-2 max, 6e-3 median)
            tol = 0.5 if k == "track" else 0.15 if k in ("vis", "conf") else 5e-2
            med = 2e-2 if k in ("vis", "conf") else 5e-3
            assert err.max().item() < tol and err.median().item() < med, (k, b, err.max().item(), err.median().item())
