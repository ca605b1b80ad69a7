"""Clip-level data parallelism: one process per GPU, time steps sharded in contiguous blocks,
ONE all-gather to re-assemble the per-step outputs (RCCL over xGMI: `backend="nccl"` is RCCL on
ROCm; `gloo` on CPU for the tests).

The reference's only multi-GPU mechanism is a thread pool that hands whole videos to free GPUs
(prepare_side_results/main.py:20-55); time steps of one clip are just as independent
(vggt/multi_view_process.py:133 loops over them), so they shard with no data-path collective.
Payload of the gather: [T/W, 17, 3] fp32 + cameras — tens of KB, latency-bound, one hop.
If W does not divide T the last rank's block is padded by repeating the last step and the pad
is dropped after the gather, which keeps the collective uniform."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(T: int, rank: int = None, world_size: int = None):
    """-> (lo, hi, T_pad): this rank owns padded steps [lo, hi); per-rank block = T_pad / W."""
    if rank is None or world_size is None:
        rank, world_size = world()
    per = (T + world_size - 1) // world_size
    return rank * per, (rank + 1) * per, per * world_size


def all_gather_steps(local: torch.Tensor, T: int) -> torch.Tensor:
    """local: this rank's [T_pad/W, ...] block -> [T, ...] on every rank (pad dropped)."""
    rank, world_size = world()
    if world_size == 1:
        return local[:T]
    local = local.contiguous()
    out = torch.empty((world_size * local.shape[0], *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out[:T]


def pack_steps(parts) -> tuple:
    """Per-step tensors [n, ...] of any dtypes -> (one uint8 buffer [n, bytes_per_step], layout).  The record of a
    time step is the concatenation of that step's slice of every part, so ONE collective moves joints and cameras
    together (north_star: "an RCCL all-gather ... only to reassemble the per-frame 3D-keypoint tensor")."""
    n = parts[0].shape[0]
    if n == 0:
        raise ValueError("pack_steps: no time steps (shard_range pads every rank to at least one)")
    cols, layout = [], []
    for p in parts:
        if p.shape[0] != n:
            raise ValueError("pack_steps: every part needs the same number of time steps")
        p = p.contiguous()
        raw = p.view(torch.uint8).reshape(n, -1)
        layout.append((p.dtype, tuple(p.shape[1:]), raw.shape[1]))
        cols.append(raw)
    return torch.cat(cols, dim=1).contiguous(), layout


def unpack_steps(buf: torch.Tensor, layout) -> list:
    """inverse of pack_steps on a [T, bytes_per_step] uint8 buffer"""
    out, off = [], 0
    T = buf.shape[0]
    for dtype, shape, nbytes in layout:
        out.append(buf[:, off:off + nbytes].reshape(-1).clone().view(dtype).reshape(T, *shape))   # clone: a fresh, aligned, flat allocation
        off += nbytes
    return out


def all_gather_packed(parts, T: int) -> list:
    """parts: this rank's per-step tensors [T_pad/W, ...] (joints, K, R, t, C ...; any dtypes, one device) ->
    the same list with [T, ...] on every rank, through ONE all_gather_into_tensor of the packed byte records
    (pad dropped).  World size 1: no collective, the parts themselves (cut to T)."""
    rank, world_size = world()
    if world_size == 1:
        return [p[:T] for p in parts]
    buf, layout = pack_steps(parts)
    out = torch.empty((world_size * buf.shape[0], buf.shape[1]), dtype=torch.uint8, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    return unpack_steps(out[:T], layout)
