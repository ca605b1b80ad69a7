#!/usr/bin/env python
"""bench.py — 3D-pose frames/sec of the MI355X-native multi-view hot path.

One "step" = one pass of the hot path over one batch of synthetic input: B time steps of an
8-view 518x518 clip through the HIP VGGT forward (DINOv2 patch embed, 24 x {frame, global}
attention, camera + depth + point heads — everything `preds = self.vggt(imgs)` computes,
vggt/vggt/infer.py:84).  1 frame = one S-view time step.  Inputs are resident in HBM when the
timed region starts.  Launched by the driver as

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One rank per GPU; time steps are independent, so ranks shard them with no data-path collective
(weak scaling); the barrier + MAX-over-ranks timing is the only communication.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md chip table
S_VIEWS, IMG = 8, 518


def vggt_flops_per_step(S):
    """algorithmic FLOPs of one S-view call without the track head."""
    return S * (1017.1 + 1015.5 + 829.9 + 1.7 + 298.5 + 298.6) * 1e9 + 24 * 4 * (S * 1374) ** 2 * 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="time steps of the clip per call (B)")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent batches of --batch time steps per step, each issued from its own host thread on "
                         "its own HIP stream (infer.process_multi_view_clip(streams=N)); 1 = one batch per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vp3d", action="store_true", help="skip the VideoPose3D lifter leg (profiling runs)")
    ap.add_argument("--cpu-views", type=int, default=2, help="views of the bounded CPU-baseline sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = "WORLD_SIZE" in os.environ and "RANK" in os.environ   # launched by torch.distributed.run
    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from skiing_analysis_pytorch_amd import _lib, vggt, weights as W
    from skiing_analysis_pytorch_amd._lib import PREC_BF16, PREC_BF16X3

    cfg = W.VGGTConfig()   # VGGT-1B, the reference's VGGT()
    model = vggt.VGGT(config=cfg, prec=PREC_BF16, head_prec=PREC_BF16X3)
    sd = W.make_vggt_state_dict(cfg, seed=0, device=dev)      # random-init weights of that architecture
    model.load_state_dict(sd)
    cpu_sd = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # CPU legs: rank 0 at N = 1 only
        cpu_sd = {k: v.cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()

    B = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.rand((B, S_VIEWS, 3, IMG, IMG), generator=g, device=dev, dtype=torch.float32)
    want = {"camera", "depth", "point"}
    # 2D keypoints of the 17 joints in every view (the reference reads them from the clip's .pt file)
    kps = torch.rand((B, S_VIEWS, 17, 2), generator=g, device=dev, dtype=torch.float32) * (IMG - 40) + 20
    from skiing_analysis_pytorch_amd import geometry, parallel

    def step():
        # VGGT forward (all three heads, as `self.vggt(imgs)` computes them) -> cameras -> DLT
        # triangulation of the joints over the 8 views -> [B, 17, 3]; under torch.distributed the
        # per-rank joints are re-assembled with the path's one collective (all-gather over xGMI)
        out = model(images, want=want)
        E, K = geometry.pose_encoding_to_extri_intri(out["pose_enc"], (IMG, IMG))
        joints = geometry.triangulate_joints(K, E[..., :3, :3].contiguous(), E[..., :3, 3].contiguous(), kps)
        if use_dist:
            joints = parallel.all_gather_steps(joints, world * B)
        out["joints3d"] = joints
        return out

    # --streams N: N independent batches per step, one host thread + HIP stream each (what
    # infer.process_multi_view_clip(streams=N) does with the calls of a clip)
    NS = max(1, args.streams)
    extra = [(torch.rand((B, S_VIEWS, 3, IMG, IMG), generator=g, device=dev, dtype=torch.float32),
              torch.rand((B, S_VIEWS, 17, 2), generator=g, device=dev, dtype=torch.float32) * (IMG - 40) + 20)
             for _ in range(NS - 1)]

    def step_on(images_k, kps_k):
        out = model(images_k, want=want)
        E, K = geometry.pose_encoding_to_extri_intri(out["pose_enc"], (IMG, IMG))
        out["joints3d"] = geometry.triangulate_joints(K, E[..., :3, :3].contiguous(), E[..., :3, 3].contiguous(), kps_k)
        return out

    def step_multi(n_streams):
        """one step = n_streams independent batches, each on its own stream from its own host thread: the
        HBM-bound phases of one (GEMM store bursts, LayerNorm, upsamples) overlap the MFMA-bound phases of the
        other.  The threads join before the step's one collective."""
        import threading
        from skiing_analysis_pytorch_amd.infer import _side_streams
        main = torch.cuda.current_stream(dev)
        side = _side_streams(dev, n_streams)   # long-lived: the model keeps one workspace per stream
        outs = [None] * n_streams
        batches = [(images, kps)] + extra

        def worker(k):
            with torch.cuda.device(dev), torch.cuda.stream(side[k]):
                outs[k] = step_on(*batches[k])
        for s_ in side:
            s_.wait_stream(main)
        th = [threading.Thread(target=worker, args=(k,)) for k in range(n_streams)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        for s_ in side:
            main.wait_stream(s_)
        joints = torch.cat([o["joints3d"] for o in outs])
        if use_dist:
            joints = parallel.all_gather_steps(joints, world * n_streams * B)
        out = outs[0]
        out["joints3d"] = joints
        return out

    step()   # prepares the handle for this frame shape (and is the one-stream warm-up)
    for _ in range(args.warmup):
        step() if NS == 1 else step_multi(NS)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    lib = _lib.lib()
    # roofline leg: bracket every global-attention launch (seq = S*1374) of the timed region
    seq_global = S_VIEWS * 1374
    _lib.check(lib.skimi_profile_start(1, seq_global), "profile_start")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step() if NS == 1 else step_multi(NS)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t1 = time.perf_counter()
    ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    _lib.check(lib.skimi_profile_stop(C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "profile_stop")
    elapsed = t1 - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out["pose_enc"]).all() and out["joints3d"].shape == (world * NS * B, 17, 3)

    frames = world * args.steps * B * NS
    value = frames / elapsed
    line = {
        "metric": "3D-pose frames/sec, 8-view 518px clips",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": "VGGT-1B multi_view_process step: 8 views x 518x518, camera+depth+point heads, "
                               "pose->cameras + 8-view DLT of 17 joints (+ all-gather of the joints across ranks)",
                   "views": S_VIEWS, "image": IMG, "time_steps_per_call": B, "streams": NS, "parallelism": f"clip-dp{world}",
                   "aggregator_prec": "bf16 MFMA, fp32 accumulate/residual/LayerNorm/softmax",
                   "head_prec": "bf16x3 (fp32-accurate)"},
        "whole_path_tflops": vggt_flops_per_step(S_VIEWS) * B * NS * args.steps * world / elapsed / 1e12,
    }
    if n.value > 0:
        avg_s = ms.value * 1e-3 / n.value
        flops_per_launch = fl.value / n.value
        ach = flops_per_launch / avg_s / 1e12
        line["roofline"] = {"kernel": "attn_q64_kernel (global attention, seq 10992, 16 heads x 64)",
                            "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(B),
                            "avg_launch_us": avg_s * 1e6, "launches": int(n.value),
                            "flops_per_launch": flops_per_launch}
    if rank == 0 and world == 1 and NS > 1 and "roofline" in line:
        # outside the timed region: the same kernel with the chip to itself (one batch, one stream)
        _lib.check(lib.skimi_profile_start(1, seq_global), "profile_start")
        step()
        torch.cuda.synchronize()
        _lib.check(lib.skimi_profile_stop(C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "profile_stop")
        if n.value > 0:
            a1 = fl.value / (ms.value * 1e-3) / 1e12
            line["roofline"]["one_stream"] = {"achieved": a1, "frac": a1 / PEAK_BF16_TFLOPS,
                                              "avg_launch_us": ms.value * 1e3 / n.value, "launches": int(n.value)}
    if rank == 0 and world == 1 and not args.no_vp3d:
        line["vp3d"] = vp3d_leg(dev, cpu=not args.no_cpu_baseline)
    if rank == 0 and cpu_sd is not None:
        line["cpu_baseline"], line["parity_vs_cpu_oracle"] = cpu_baseline(cpu_sd, cfg, args.cpu_views, model, dev)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_threads():
    """Threads for the CPU legs: this process's CPU share (16 on a one-GPU box), set in torch."""
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, int(os.environ.get("SKIMI_CPU_THREADS", "16"))))
    torch.set_num_threads(threads)
    return threads


def vp3d_leg(dev, cpu=True):
    """Second leg of the path (BASELINE configs[0]): the VideoPose3D TemporalModel lifter, receptive
    field 27, 17 COCO joints, 243-frame clips of synthetic 2D keypoints, fp32-accurate mode.  Not part
    of `value`; reported beside it: HIP-event time per call, clips/s, and the achieved fraction of the
    HBM roofline on the algorithmic bytes (weights once per call + activations once per layer)."""
    from skiing_analysis_pytorch_amd import vp3d, weights as W
    from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
    fw = [3, 3, 3]
    sd = W.make_vp3d_state_dict(seed=0, filter_widths=fw)
    m = vp3d.TemporalModel(17, 2, 17, fw, prec=PREC_BF16X3)
    m.load_state_dict(sd)
    wbytes = sum(v.numel() * 4 for k, v in sd.items() if k.endswith("weight") and v.dim() == 3)
    res = {"model": "TemporalModel RF 27, 1024 channels, bf16x3 (fp32-accurate), 243-frame clips", "hbm_peak_GBps": 8000.0}
    for B in (1, 64):
        x = torch.randn(B, 243, 17, 2, device=dev)
        out = torch.empty(B, 217, 17, 3, device=dev)
        for _ in range(3):
            m(x, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50 if B == 1 else 10
        e0.record()
        for _ in range(n):
            m(x, out=out)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / n
        # activations: fp32 [B*L, 1024] written and read once per conv (10 convs, L shrinking 241 -> 217)
        act = sum(2 * B * L * 1024 * 4 for L in (241, 235, 235, 217, 217)) * 2
        res[f"clips_{B}"] = {"us_per_call": t * 1e6, "clips_per_s": B / t, "frames_per_s": B * 243 / t,
                             "algorithmic_GB": (wbytes + act) / 1e9, "achieved_GBps": (wbytes + act) / t / 1e9,
                             "frac_of_hbm_peak": (wbytes + act) / t / 8e12}
    # the receptive-field-243 lifter of BASELINE configs[4] (5 blocks, 67.8 MB of fp32 weights): 485 input
    # frames -> 243 output frames
    fw5 = [3, 3, 3, 3, 3]
    sd5 = W.make_vp3d_state_dict(seed=0, filter_widths=fw5)
    m5 = vp3d.TemporalModel(17, 2, 17, fw5, prec=PREC_BF16X3)
    m5.load_state_dict(sd5)
    wbytes5 = sum(v.numel() * 4 for k, v in sd5.items() if k.endswith("weight") and v.dim() == 3)
    res["rf243"] = {"model": "TemporalModel RF 243 (filter widths 3,3,3,3,3), 1024 channels, bf16x3, 485 -> 243 frames"}
    for B in (1, 64):
        x = torch.randn(B, 485, 17, 2, device=dev)
        out = torch.empty(B, 243, 17, 3, device=dev)
        for _ in range(3):
            m5(x, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50 if B == 1 else 5
        e0.record()
        for _ in range(n):
            m5(x, out=out)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / n
        act = sum(2 * B * L * 1024 * 4 for L in (483, 477, 477, 459, 459, 405, 405, 243, 243)) * 2
        res["rf243"][f"clips_{B}"] = {"us_per_call": t * 1e6, "clips_per_s": B / t, "frames_per_s": B * 243 / t,
                                      "algorithmic_GB": (wbytes5 + act) / 1e9, "achieved_GBps": (wbytes5 + act) / t / 1e9,
                                      "frac_of_hbm_peak": (wbytes5 + act) / t / 8e12}
    del m5, sd5
    if cpu:
        from oracle import vp3d_oracle   # test infrastructure: the timed CPU baseline only
        kp = W.make_keypoints_2d(frames=243, seed=1).numpy()
        threads = cpu_threads()
        vp3d_oracle.lift_clip(sd, kp, 1920, 1080, fw)
        t0 = time.perf_counter()
        for _ in range(3):
            ref = vp3d_oracle.lift_clip(sd, kp, 1920, 1080, fw)
        tc = (time.perf_counter() - t0) / 3
        got = vp3d.lift_clip(m, kp, 1920, 1080)
        res["cpu_oracle"] = {"s_per_clip_with_flip_tta": tc, "cores": threads,
                             "max_abs_joint_err_vs_hip": float(abs(got - ref).max())}
    return res


def pmc_traffic(time_steps):
    """HBM bytes per global-attention launch from the committed rocprofv3 PMC passes
    (profiles/r01_attn_traffic.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE), or None when the
    profile was taken at another launch shape."""
    f = Path(__file__).resolve().parent / "profiles" / "r01_attn_traffic.json"
    try:
        d = json.loads(f.read_text())
    except (OSError, ValueError):
        return None
    return d["hbm_bytes_per_launch"] if d.get("time_steps") == time_steps else None


def cpu_baseline(cpu_sd, cfg, views, model, dev):
    """The oracle (fp32 CPU restatement of the reference, oracle/vggt_oracle.py) on a bounded
    sample: ONE `views`-view 518x518 step on this host's cores.  An 8-view step costs
    flops(8)/flops(views) more; the value is scaled by that ratio and the sample says so."""
    from oracle import vggt_oracle

    # cores actually granted to this process (the GPU box gives a 1-GPU job a 16-core share;
    # os.cpu_count() reports the whole host and oversubscribes)
    threads = cpu_threads()
    img = torch.rand((1, views, 3, IMG, IMG), generator=torch.Generator().manual_seed(5))
    d = cfg.to_dict()
    d["enable_track"] = False
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = vggt_oracle.vggt_forward(cpu_sd, img, d)
        dt = time.perf_counter() - t0
    ratio = vggt_flops_per_step(S_VIEWS) / vggt_flops_per_step(views)
    base = {"value": 1.0 / (dt * ratio), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"one {views}-view 518x518 step = {dt:.1f} s measured; 8-view step = x{ratio:.2f} FLOPs (extrapolated)",
            "measured_seconds": dt}
    # the same full-size input through the benchmarked HIP model (bf16 aggregator, fp32-accurate
    # heads) against the fp32 CPU oracle: the checker, outside the timed region
    got = model(img.to(dev), want={"camera", "depth"})
    pe = (got["pose_enc"].cpu() - ref["pose_enc"]).abs().max().item()
    rel = ((got["depth"].cpu() - ref["depth"]).abs() / (ref["depth"].abs() + 1.0))
    parity = {"sample": f"VGGT-1B, {views} views x 518x518, synthetic weights; bf16 aggregator vs fp32 CPU oracle",
              "pose_enc_max_abs_err": pe, "depth_rel_err_median": rel.median().item(),
              "depth_rel_err_p99": rel.flatten().kthvalue(int(0.99 * rel.numel())).values.item()}
    return base, parity


if __name__ == "__main__":
    main()
