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


def vggt_flops_per_step(S, track=True):
    """algorithmic FLOPs of one S-view call (40.6 TFLOP at S = 8 with the track head on
    17 query points, 39.6 without)."""
    return S * (1017.1 + 1015.5 + 829.9 + 1.7 + 298.5 + 298.6 + (132.4 if track else 0.0)) * 1e9 + 24 * 4 * (S * 1374) ** 2 * 1024


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
    ap.add_argument("--no-track", action="store_true", help="leave the track head out of the step (39.6 instead of 40.6 TFLOP)")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the timing of the fp32-accurate (bf16x3) mode")
    ap.add_argument("--no-fp8", action="store_true", help="skip the MXFP8 leg (BASELINE config 5)")
    ap.add_argument("--prec", choices=["f16", "bf16"], default="f16",
                    help="operand format of the aggregator's Linears in the timed region: f16 (default: the cheapest mode whose 3D "
                         "joints stay within north_star's 1e-3 of the fp32 CPU path) or bf16 (the reference's autocast format, which "
                         "does not); the other one is timed as a leg of the same line")
    ap.add_argument("--no-other-prec", action="store_true", help="skip the leg that times the other of f16 / bf16")
    ap.add_argument("--with-upload", action="store_true",
                    help="start every step from uint8 HWC frames in pinned HOST memory: PCIe upload + the device preprocessing of "
                         "load_and_preprocess_images (vggt/load.py:38-183; vggt/vggt/infer.py:77 does both per call) inside the timed "
                         "region.  The default keeps the inputs resident in HBM (the contract's `value`); with this flag the line is "
                         "the PCIe-inclusive rate and says so in config.inputs")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes, before
        # anything in this process touches the GPU
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = "WORLD_SIZE" in os.environ and "RANK" in os.environ   # launched by torch.distributed.run
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU is the contract")
    # SKIMI_BENCH_ONE_GPU=1: rehearsal of the N-rank path on a one-GPU box -- every rank uses cuda:0 and the
    # collectives run over gloo (RCCL refuses two ranks on one device).  Never a measurement.
    rehearsal = os.environ.get("SKIMI_BENCH_ONE_GPU") == "1" and use_dist
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    dev = torch.device("cuda", local_rank)

    from skiing_analysis_pytorch_amd import _lib, vggt, weights as W
    from skiing_analysis_pytorch_amd._lib import PREC_BF16, PREC_BF16X3, PREC_F16

    cfg = W.VGGTConfig()   # VGGT-1B, the reference's VGGT()
    agg_prec = PREC_F16 if args.prec == "f16" else PREC_BF16
    model = vggt.VGGT(config=cfg, prec=agg_prec, head_prec=PREC_BF16X3)
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
    track = not args.no_track
    want = {"camera", "depth", "point"} | ({"track"} if track else set())
    # 2D keypoints of the 17 joints in every view (the reference reads them from the clip's .pt file);
    # the track head's query points are the step's view-0 keypoints in 518-space
    kps = torch.rand((B, S_VIEWS, 17, 2), generator=g, device=dev, dtype=torch.float32) * (IMG - 40) + 20
    from skiing_analysis_pytorch_amd import geometry, parallel

    # --streams N: N independent batches per step, one host thread + HIP stream each (what
    # infer.process_multi_view_clip(streams=N) does with the calls of a clip)
    NS = max(1, args.streams)
    extra = [(torch.rand((B, S_VIEWS, 3, IMG, IMG), generator=g, device=dev, dtype=torch.float32),
              torch.rand((B, S_VIEWS, 17, 2), generator=g, device=dev, dtype=torch.float32) * (IMG - 40) + 20)
             for _ in range(NS - 1)]

    active = {"model": model}    # the fp8 leg re-runs the same step on the SKIMI_PREC_FP8 model
    host_u8 = None
    if args.with_upload:
        # the same frames as uint8 HWC in pinned host memory, one tensor per frame (what a decoder hands over)
        from skiing_analysis_pytorch_amd.preprocess import load_and_preprocess_images_device

        def to_host_u8(img5):
            u8 = (img5.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 1, 3, 4, 2).reshape(-1, IMG, IMG, 3).cpu()
            return [f.contiguous().pin_memory() for f in u8]
        host_u8 = {id(images): to_host_u8(images)}
        for im_k, _ in extra:
            host_u8[id(im_k)] = to_host_u8(im_k)

    def step_on(images_k, kps_k):
        # VGGT forward (all four heads, as `self.vggt(imgs, query_points)` computes them) -> cameras -> DLT
        # triangulation of the joints over the 8 views -> [B, 17, 3]
        if host_u8 is not None:   # PCIe upload + device preprocessing (identity resize at 518 x 518, ToTensor) in the timed region
            images_k = load_and_preprocess_images_device(host_u8[id(images_k)], "crop", dev).view(B, S_VIEWS, 3, IMG, IMG)
        out = active["model"](images_k, query_points=kps_k[:, 0].contiguous() if track else None, want=want)
        E, K = geometry.pose_encoding_to_extri_intri(out["pose_enc"], (IMG, IMG))
        out["joints3d_local"] = geometry.triangulate_joints(K, E[..., :3, :3].contiguous(), E[..., :3, 3].contiguous(), kps_k)
        out["E_local"], out["K_local"] = E, K
        return out

    def gather(out, n_total):
        # under torch.distributed the per-rank joints AND cameras are re-assembled with the path's ONE collective
        # (parallel.all_gather_packed: one byte record per time step, all-gather over xGMI), as process_multi_view_clip does
        if use_dist:
            out["joints3d"], out["E"], out["K"] = parallel.all_gather_packed([out["joints3d_local"], out["E_local"], out["K_local"]], n_total)
        else:
            out["joints3d"], out["E"], out["K"] = out["joints3d_local"], out["E_local"], out["K_local"]
        return out

    def step():
        return gather(step_on(images, kps), world * B)

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
        out = outs[0]
        for key in ("joints3d_local", "E_local", "K_local"):
            out[key] = torch.cat([o[key] for o in outs])
        return gather(out, world * n_streams * B)

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
    if use_dist:
        # the gathered [N*B*streams, 17, 3] tensor must hold every rank's block: each rank checks its own
        # block bit for bit, rank 0 checks every block against that rank's checksum (outside the timed region)
        nloc = NS * B
        mine = out["joints3d"][rank * nloc:(rank + 1) * nloc]
        assert torch.equal(mine, out["joints3d_local"]), "all-gather: this rank's block differs from its local joints"
        sums = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(sums, out["joints3d_local"].double().sum().reshape(1))
        blocks = out["joints3d"].double().reshape(world, -1).sum(dim=1)
        assert torch.equal(sums, blocks), "all-gather: a rank's block is missing from the gathered joints"
        assert torch.isfinite(out["joints3d"]).all()
        assert out["E"].shape == (world * nloc, S_VIEWS, 3, 4) and torch.equal(out["E"][rank * nloc:(rank + 1) * nloc], out["E_local"])
    # BASELINE config 4's last stage on the gathered joints (outside the timed region: it runs once per clip on the host)
    from skiing_analysis_pytorch_amd import fuse
    smoothed = fuse.temporal_smooth_ema(out["joints3d"].cpu().numpy().astype("float64"))
    assert smoothed.shape == tuple(out["joints3d"].shape)

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
        "dtype": args.prec,
        "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a measurement)" if rehearsal else ""),
        "config": {"workload": "VGGT-1B multi_view_process step (BASELINE config 3): 8 views x 518x518, camera+depth+point"
                               + ("+track heads (17 query points per step), " if track else " heads, ")
                               + "pose->cameras + 8-view DLT of 17 joints (+ all-gather of the joints across ranks)",
                   "inputs": ("uint8 HWC frames in pinned host memory: PCIe upload + device preprocessing INSIDE the timed region "
                              "(--with-upload: the PCIe-inclusive rate, not the contract's resident-input value)" if args.with_upload
                              else "fp32 frames resident in HBM when the timed region starts"),
                   "views": S_VIEWS, "image": IMG, "time_steps_per_call": B, "streams": NS, "parallelism": f"clip-dp{world}",
                   "aggregator_prec": ("fp16 operands in the Linears (patch embed, qkv, proj, fc1, fc2: v_mfma_f32_32x32x16_f16), bf16 attention "
                                       "products, fp32 accumulate/residual/LayerNorm/softmax; fp16 is the reference's autocast format below "
                                       "compute capability 8 (vggt/vggt/infer.py:77-82)" if args.prec == "f16" else
                                       "bf16 MFMA, fp32 accumulate/residual/LayerNorm/softmax"),
                   "head_prec": "bf16x3 (fp32-accurate)"},
        "whole_path_tflops": vggt_flops_per_step(S_VIEWS, track) * B * NS * args.steps * world / elapsed / 1e12,
        "tflop_per_step": vggt_flops_per_step(S_VIEWS, track) / 1e12,
    }
    if n.value > 0:
        avg_s = ms.value * 1e-3 / n.value
        flops_per_launch = fl.value / n.value
        ach = flops_per_launch / avg_s / 1e12
        line["roofline"] = {"kernel": "attn_q64_kernel (global attention, seq 10992, 16 heads x 64)",
                            "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(B),
                            "traffic_source": "static: profiles/r03_attn_traffic.json (rocprofv3 --pmc passes of this kernel "
                                              "source, matched by its sha256; null when the kernel changed since)",
                            "avg_launch_us": avg_s * 1e6, "launches": int(n.value),
                            "flops_per_launch": flops_per_launch}
        pk = measured_peaks()
        if pk:   # what a register-only MFMA loop delivers on this part (tools/peaks.hip): context for `frac`, which stays on the datasheet peak
            line["roofline"]["peak_measured"] = {"unit": "TFLOP/s", "random_operands": pk["random_operands"], "zero_operands": pk["zero_operands"],
                                                 "frac_of_random_operand_peak": ach / pk["random_operands"],
                                                 "source": "static: profiles/r03_peaks.json (tools/peaks.hip, register-only v_mfma_f32_32x32x16_bf16 loop)"}
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
    other_model, other_name = None, ("bf16" if args.prec == "f16" else "f16")
    if rank == 0 and world == 1 and not args.no_other_prec:
        # the same step with the other 16-bit operand format (bf16 = the reference's own autocast precision): same
        # inputs, same step code, its own timed region; not part of `value`
        mo = vggt.VGGT(config=cfg, prec=PREC_BF16 if other_name == "bf16" else PREC_F16, head_prec=PREC_BF16X3)
        mo.load_state_dict(W.make_vggt_state_dict(cfg, seed=0, device=dev))
        torch.cuda.empty_cache()
        active["model"] = mo
        step()
        step() if NS == 1 else step_multi(NS)
        torch.cuda.synchronize()
        to = time.perf_counter()
        no = max(2, args.steps // 2)
        for _ in range(no):
            outo = step() if NS == 1 else step_multi(NS)
        torch.cuda.synchronize()
        dto = (time.perf_counter() - to) / no
        assert torch.isfinite(outo["pose_enc"]).all()
        line[other_name + "_mode"] = {"value": B * NS / dto, "unit": "frames/s", "ms_per_step": dto * 1e3, "steps": no,
                                      "mode": ("bf16 operands everywhere in the aggregator: the reference's GPU autocast precision "
                                               "(vggt/vggt/infer.py:78-84)" if other_name == "bf16" else
                                               "fp16 operands in the aggregator's Linears, bf16 attention products"),
                                      "pose_enc_max_abs_diff_vs_timed_mode": (outo["pose_enc"] - out["pose_enc"]).abs().max().item()}
        active["model"] = model
        other_model = mo
    if rank == 0 and world == 1 and not args.no_fp8:
        # BASELINE config 5: the same step with the four Linears (qkv / proj / fc1 / fc2) of every block on the MXFP8 MFMA
        # (SKIMI_PREC_FP8); not part of `value`.  Same inputs, same step code, its own timed region.
        from skiing_analysis_pytorch_amd._lib import PREC_FP8
        m8 = vggt.VGGT(config=cfg, prec=PREC_FP8, head_prec=PREC_BF16X3)
        m8.load_state_dict(W.make_vggt_state_dict(cfg, seed=0, device=dev))
        torch.cuda.empty_cache()
        active["model"] = m8
        step()
        step() if NS == 1 else step_multi(NS)
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        n8 = max(2, args.steps // 2)
        for _ in range(n8):
            out8 = step() if NS == 1 else step_multi(NS)
        torch.cuda.synchronize()
        dt8 = (time.perf_counter() - t8) / n8
        assert torch.isfinite(out8["pose_enc"]).all()
        line["fp8"] = {"value": B * NS / dt8, "unit": "frames/s", "ms_per_step": dt8 * 1e3, "steps": n8,
                       "mode": "SKIMI_PREC_FP8: MXFP8 (e4m3 + E8M0 per 32 K) qkv / proj / fc1 / fc2 of the 72 blocks on "
                               "v_mfma_scale_f32_32x32x64_f8f6f4, activations quantised inside their producers (LayerNorm -> "
                               "MXFP8, the attention kernel's output rows -> MXFP8, fc1's GELU epilogue -> MXFP8); attention "
                               "products and residual stream as the bf16 mode; heads bf16x3",
                       "pose_enc_max_abs_diff_vs_timed_mode": (out8["pose_enc"] - out["pose_enc"]).abs().max().item()}
        active["model"] = model
        fp8_model = m8
    else:
        fp8_model = None
    if rank == 0 and world == 1 and not args.no_vp3d:
        line["vp3d"] = vp3d_leg(dev, cpu=not args.no_cpu_baseline)
    if rank == 0 and cpu_sd is not None:
        line["cpu_baseline"], line["mpjpe_vs_cpu_oracle"], m3 = cpu_baseline(
            cpu_sd, cfg, {args.prec: model, other_name: other_model}, dev, track, not args.no_parity_mode, fp8_model)
        # the headline against north_star's bar on the 3D joints: `value` is the timed mode's throughput,
        # `within_bar` says whether that mode's joints are within 1e-3 of the fp32 CPU path on the ring rig,
        # `value_within_bar` is the fastest measured mode that is
        pm = line["mpjpe_vs_cpu_oracle"]
        line["within_bar"] = bool(pm[args.prec + "_mode"]["within_bar"])
        if m3 is not None:
            # the parity mode's own throughput: the very step of the timed region (same batches, streams, heads,
            # pose -> cameras -> DLT), on the model whose every operand is bf16x3
            active["model"] = m3
            step() if NS == 1 else step_multi(NS)
            torch.cuda.synchronize()
            step() if NS == 1 else step_multi(NS)    # second warm-up step (both streams' workspaces and tables exist)
            torch.cuda.synchronize()
            n3 = max(10, args.steps // 2)
            t3 = time.perf_counter()
            for _ in range(n3):
                out3 = step() if NS == 1 else step_multi(NS)
            torch.cuda.synchronize()
            dt3 = (time.perf_counter() - t3) / n3
            assert torch.isfinite(out3["joints3d"]).all()
            line["parity_mode"] = {"value": B * NS / dt3, "unit": "frames/s", "ms_per_step": dt3 * 1e3, "steps": n3,
                                   "time_steps_per_call": B, "streams": NS,
                                   "mode": "bf16x3 everywhere (fp32-accurate: the mode that meets the 1e-3 bar), the same full "
                                           "step as the timed region; attention on attention_x3.hip, Linears on the LDS-DMA "
                                           "bf16x3 kernels"}
            active["model"] = model
            del m3
        cands = [(line["value"], args.prec)] if line["within_bar"] else []
        for name, key in ((other_name, other_name + "_mode"), ("bf16x3", "parity_mode"), ("fp8", "fp8")):
            mk = {"bf16x3": "bf16x3_parity_mode", "fp8": "fp8_mode"}.get(name, name + "_mode")
            if key in line and pm.get(mk, {}).get("within_bar"):
                cands.append((line[key]["value"], name))
        if cands:
            line["value_within_bar"], line["mode_within_bar"] = max(cands)
        else:
            line["value_within_bar"], line["mode_within_bar"] = None, None
    if rank == 0:
        if "within_bar" not in line:   # no CPU-oracle leg in this run (N > 1 or --no-cpu-baseline): the bar is judged by the N = 1 run
            line["within_bar"] = None
            line["within_bar_note"] = ("not measured in this run: the joints-vs-CPU-oracle leg runs on rank 0 at N = 1 only; the same mode measured "
                                       "4.7e-4 ring-rig MPJPE there (profiles/r03_bench_line.json)")
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run N ranks (one per GPU) as children of
    `python -m torch.distributed.run` on 127.0.0.1, relay rank 0's JSON line, return non-zero if any
    rank failed.  This process never initialises the GPU (no exec after HIP init, no device held by
    the parent)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    got_line = False
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            got_line = True
        sys.stdout.write(ln)
        sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and not got_line:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return rc


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
    HBM roofline on SURVEY §8(d)'s algorithmic bytes (fp32 weights once per call + B x (input + output))."""
    from skiing_analysis_pytorch_amd import vp3d, weights as W
    from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
    fw = [3, 3, 3]
    sd = W.make_vp3d_state_dict(seed=0, filter_widths=fw)
    m = vp3d.TemporalModel(17, 2, 17, fw, prec=PREC_BF16X3)
    m.load_state_dict(sd)
    wbytes = sum(v.numel() * 4 for k, v in sd.items() if k.endswith("weight") and v.dim() == 3)
    res = {"model": "TemporalModel RF 27, 1024 channels, bf16x3 (fp32-accurate), 243-frame clips", "hbm_peak_GBps": 8000.0}
    for B in (1, 2, 64):   # 2 = the call the reference makes: the clip and its flipped copy (flip-TTA, VideoPose3D/run.py:1070-1083)
        x = torch.randn(B, 269, 17, 2, device=dev)     # 243 frames edge-padded by the receptive field (generators.py:216-239)
        out = torch.empty(B, 243, 17, 3, device=dev)
        for _ in range(3):
            m(x, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50 if B <= 2 else 10
        e0.record()
        for _ in range(n):
            m(x, out=out)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / n
        # algorithmic bytes = fp32 weights once per call + B x (input + output)
        alg = wbytes + B * (269 * 34 + 243 * 51) * 4
        res[f"clips_{B}"] = {"us_per_call": t * 1e6, "clips_per_s": B / t, "frames_per_s": B * 243 / t,
                             "algorithmic_GB": alg / 1e9, "achieved_GBps": alg / t / 1e9,
                             "frac_of_hbm_peak": alg / t / 8e12}
        if B == 1:
            res["clips_1"]["traffic"] = vp3d_traffic()
            res["clips_1"]["traffic_source"] = ("static: profiles/r03_vp3d_traffic.json (rocprofv3 --pmc passes of this kernel source, "
                                                "matched by its sha256; L2 <-> fabric bytes per call: each layer's 1 MB of activations is "
                                                "fetched once per XCD from the Infinity Cache; null when the kernel changed since)")
        if B == 2:
            res["clips_2"]["note"] = "the reference's own call shape: [clip, flipped clip] in one batch (test-time augmentation)"
        if B > 2:
            # a batch of clips re-uses every weight B x 243 times: the call is bound by the matrix pipe, not by HBM.
            # 4.31 GFLOP per clip; the fp32-accurate mode issues three bf16 MFMAs per product
            fl = 4.31e9 * B
            res[f"clips_{B}"].update({"bound": "mfma", "algorithmic_TFLOPs": fl / t / 1e12, "mfma_issue_TFLOPs": 3 * fl / t / 1e12,
                                      "frac_of_mfma_peak": 3 * fl / t / 2.5e15})
    # the receptive-field-243 lifter of BASELINE configs[4] (5 blocks, 67.8 MB of fp32 weights): 485 input
    # frames -> 243 output frames
    fw5 = [3, 3, 3, 3, 3]
    sd5 = W.make_vp3d_state_dict(seed=0, filter_widths=fw5)
    m5 = vp3d.TemporalModel(17, 2, 17, fw5, prec=PREC_BF16X3)
    m5.load_state_dict(sd5)
    wbytes5 = sum(v.numel() * 4 for k, v in sd5.items() if k.endswith("weight") and v.dim() == 3)
    res["rf243"] = {"model": "TemporalModel RF 243 (filter widths 3,3,3,3,3), 1024 channels, bf16x3, 485 -> 243 frames"}
    for B in (1, 2, 64):
        x = torch.randn(B, 485, 17, 2, device=dev)
        out = torch.empty(B, 243, 17, 3, device=dev)
        for _ in range(3):
            m5(x, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50 if B <= 2 else 5
        e0.record()
        for _ in range(n):
            m5(x, out=out)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / n
        alg = wbytes5 + B * (485 * 34 + 243 * 51) * 4
        res["rf243"][f"clips_{B}"] = {"us_per_call": t * 1e6, "clips_per_s": B / t, "frames_per_s": B * 243 / t,
                                      "algorithmic_GB": alg / 1e9, "achieved_GBps": alg / t / 1e9,
                                      "frac_of_hbm_peak": alg / t / 8e12}
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
        # the same reference-level call on the HIP path, end to end on the host clock: normalise, edge-pad, flip copy (host
        # NumPy, as the reference), upload, the B = 2 forward, TTA merge, download
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            vp3d.lift_clip(m, kp, 1920, 1080)
        tg = (time.perf_counter() - t0) / 20
        res["cpu_oracle"] = {"s_per_clip_with_flip_tta": tc, "cores": threads,
                             "max_abs_joint_err_vs_hip": float(abs(got - ref).max()),
                             "hip_lift_clip_s_per_clip_with_flip_tta": tg, "speedup_end_to_end": tc / tg}
    return res


def vp3d_traffic():
    import hashlib
    root = Path(__file__).resolve().parent
    try:
        d = json.loads((root / "profiles" / "r03_vp3d_traffic.json").read_text())
        sha = hashlib.sha256((root / "skiing_analysis_pytorch_amd" / "csrc" / "vp3d_stream.hip").read_bytes()).hexdigest()[:16]
    except (OSError, ValueError):
        return None
    return d["fabric_bytes_per_call"] if d.get("kernel_sha") == sha else None


def measured_peaks():
    f = Path(__file__).resolve().parent / "profiles" / "r03_peaks.json"
    try:
        return json.loads(f.read_text())["mfma_bf16_32x32x16_register_loop_TFLOPs"]
    except (OSError, ValueError, KeyError):
        return None


def attn_kernel_sha():
    """sha256 of the roofline kernel's source: ties a committed PMC profile to the kernel it measured"""
    import hashlib
    src = Path(__file__).resolve().parent / "skiing_analysis_pytorch_amd" / "csrc" / "attention_q64.hip"
    return hashlib.sha256(src.read_bytes()).hexdigest()[:16]


def pmc_traffic(time_steps):
    """HBM bytes per global-attention launch from the committed rocprofv3 PMC passes
    (profiles/r03_attn_traffic.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, tools/pmc_traffic.py).  A static
    figure, not measured in this run: returned only when the profile was taken at this launch shape AND
    on this kernel source (sha256 recorded in the file), otherwise None."""
    f = Path(__file__).resolve().parent / "profiles" / "r03_attn_traffic.json"
    try:
        d = json.loads(f.read_text())
    except (OSError, ValueError):
        return None
    if d.get("time_steps") != time_steps or d.get("kernel_sha") != attn_kernel_sha():
        return None
    return d["hbm_bytes_per_launch"]


def cpu_baseline(cpu_sd, cfg, models16, dev, track, parity_mode, fp8_model=None):
    """The oracle (fp32 CPU restatement of the reference, oracle/vggt_oracle.py) MEASURED on one full
    8-view 518x518 step of the benchmarked workload (all heads) on this host's cores -- the bounded sample:
    about a minute of CPU work.  The same step then goes through every HIP mode (fp16 / bf16 aggregator, MXFP8,
    bf16x3 everywhere), and the error the metric is defined on -- MPJPE (VideoPose3D/common/loss.py:11-17) of the
    8-view DLT joints against the joints from the oracle's cameras, on identical 2D keypoints -- is reported for each,
    on two scenes: the RING RIG (oracle/joints_check.py: 8 cameras on a ring of radius 3 around a 1.7-unit skier, FoV 55
    degrees; the cameras under test = ring + the mode's pose_enc error) -- the well-conditioned standard scene
    `within_bar` is judged on -- and the synthetic model's NATIVE scene (random-init camera head: FoV above pi, seven of
    eight cameras in a 0.05-unit cluster), which amplifies the same pose error 20x-200x and is reported for continuity.
    The oracle is the checker here, outside every timed GPU region."""
    from oracle import joints_check, vggt_oracle
    from skiing_analysis_pytorch_amd import geometry, vggt
    from skiing_analysis_pytorch_amd._lib import PREC_BF16X3

    # cores actually granted to this process (the GPU box gives a 1-GPU job a 16-core share;
    # os.cpu_count() reports the whole host and oversubscribes)
    threads = cpu_threads()
    gen = torch.Generator().manual_seed(5)
    img = torch.rand((1, S_VIEWS, 3, IMG, IMG), generator=gen)
    q = torch.rand((1, 17, 2), generator=gen) * (IMG - 80) + 40
    d = cfg.to_dict()
    d["enable_track"] = bool(track)
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = vggt_oracle.vggt_forward(cpu_sd, img, d, query_points=q if track else None)
        dt = time.perf_counter() - t0
    base = {"value": 1.0 / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"one full 8-view 518x518 step (all heads{', 17 query points' if track else ''}) on the fp32 CPU "
                      f"oracle = {dt:.1f} s, measured, not extrapolated",
            "measured_seconds": dt}
    kps, Xw, joints_ref = joints_check.keypoints_from_oracle_cameras(ref["pose_enc"], (IMG, IMG), joints=17, seed=5)
    ring, kps_ring, joints_ring = joints_check.ring_rig_scene(ref["pose_enc"], (IMG, IMG), joints=17, seed=5)

    def dlt(pose_enc, k2d):
        E, K = geometry.pose_encoding_to_extri_intri(pose_enc.to(dev), (IMG, IMG))
        return geometry.triangulate_joints(K, E[..., :3, :3].contiguous(), E[..., :3, 3].contiguous(), k2d.to(dev)).cpu().numpy()

    def measure(m):
        out = m(img.to(dev), query_points=q.to(dev) if track else None, want={"camera", "depth"} | ({"track"} if track else set()))
        rel = ((out["depth"].cpu() - ref["depth"]).abs() / (ref["depth"].abs() + 1.0))
        r = {"mpjpe_ring_rig": joints_check.mpjpe(dlt(joints_check.ring_rig_test_pose_enc(ring, out["pose_enc"], ref["pose_enc"]), kps_ring), joints_ring),
             "mpjpe_native_scene": joints_check.mpjpe(dlt(out["pose_enc"], kps), joints_ref),
             "pose_enc_max_abs_err": (out["pose_enc"].cpu() - ref["pose_enc"]).abs().max().item(),
             "depth_rel_err_median": rel.median().item(), "depth_rel_err_max": rel.max().item()}
        if track:
            dtr = (out["track"].cpu() - ref["track"]).abs()
            r["track_px_err_median"], r["track_px_err_max"] = dtr.median().item(), dtr.max().item()
        r["within_bar"] = r["mpjpe_ring_rig"] <= 1e-3
        r["within_bar_native_scene"] = r["mpjpe_native_scene"] <= 1e-3
        return r

    parity = {"sample": "VGGT-1B, 8 views x 518x518, synthetic weights; joints = 8-view DLT of 17 keypoints; reference = the same DLT "
                        "with the fp32 CPU oracle's cameras.  within_bar is judged on the ring rig.",
              "bar": 1e-3, "ring_rig": "8 cameras on a ring (radius 3, height 0.6, FoV 55 deg, unit quaternions) around 17 points of a "
                                       "1.7-unit skier; test cameras = ring + (pose_enc - oracle pose_enc)",
              "native_scene_scale": float(abs(joints_ref).max()),
              "native_dlt_conditioning_error": joints_check.conditioning_error(Xw, joints_ref)}
    for name, m in models16.items():
        if m is not None:
            parity[name + "_mode"] = measure(m)
    if fp8_model is not None:    # config 5's parity, reported separately
        parity["fp8_mode"] = measure(fp8_model)
    m3 = None
    if parity_mode:
        m3 = vggt.VGGT(config=cfg, prec=PREC_BF16X3, head_prec=PREC_BF16X3)
        m3.load_state_dict(cpu_sd)
        parity["bf16x3_parity_mode"] = measure(m3)
    return base, parity, m3


if __name__ == "__main__":
    main()
