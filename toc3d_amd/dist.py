"""Multi-GPU sharding of the hot path: independent frames, one process per GPU, no data-path collective.

The backbone never mixes frames (windows are per view, scorer queries per frame), and the reference
itself only ever shards *samples* (``datasets/samplers/distributed_sampler.py:41-44``: rank r owns the contiguous chunk
``[r*n/R, (r+1)*n/R)`` so temporal sequences stay on one rank).  ``frames_for_rank`` reproduces that rule;
``all_gather_features`` is the single exchange BASELINE.json config 5 asks for: the per-frame neck features gathered
where the detection head consumes them (``dense_heads/streampetr_head.py:627-631``).  Backend: ``torch.distributed``
("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


def frames_for_rank(n_frames: int, rank: int, world: int) -> range:
    """Contiguous chunk of frame ids owned by ``rank`` (the reference's DistributedSampler rule; the total is padded up
    to a multiple of ``world`` there, here the tail ranks simply get one frame less)."""
    per = -(-n_frames // world)
    lo = min(rank * per, n_frames)
    return range(lo, min(lo + per, n_frames))


def all_gather_features(feat: torch.Tensor, out: torch.Tensor = None, dtype=torch.bfloat16) -> torch.Tensor:
    """feat (Nv, C, h, w) of this rank's frame -> (world, Nv, C, h, w) on every rank, exchanged in ``dtype``
    (bf16: 3.07 MB per rank at 6x256x20x50).  One collective per frame, none inside the backbone."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    send = feat.to(dtype).contiguous()
    if out is None:
        out = torch.empty((world,) + tuple(send.shape), dtype=dtype, device=send.device)
    if world == 1:
        out[0].copy_(send)
        return out
    dist.all_gather_into_tensor(out.flatten(0, 1), send)      # (world*Nv, ...) view: the layout gloo and RCCL both accept
    return out


def max_over_ranks(seconds: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
