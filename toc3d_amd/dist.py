"""Multi-GPU sharding of the hot path: independent frames, one process per GPU, no data-path collective.

The backbone never mixes frames (windows are per view, scorer queries per frame), and the reference
itself only ever shards *samples* (``datasets/samplers/distributed_sampler.py:41-44``: rank r owns the contiguous chunk
``[r*n/R, (r+1)*n/R)`` so temporal sequences stay on one rank).  ``frames_for_rank`` reproduces that rule;
``all_gather_features`` / ``FeatureGather`` are the single exchange BASELINE.json config 5 asks for: the per-frame neck
features gathered where the detection head consumes them (``dense_heads/streampetr_head.py:627-631``).  Backend:
``torch.distributed`` ("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

``FeatureGather`` keeps the exchange off the compute stream: frame t's features are handed to a side stream, the ring
all-gather (3.07 MB per rank in bf16 at 6x256x20x50 -- latency-, not bandwidth-bound on 7 x 153 GB/s xGMI links) runs while
the backbone already works on frame t+1, and the consumer waits on the returned ticket only when it reads the gathered
tensor.  ``timed_steps`` is bench.py's timing protocol (barrier + device sync on both sides, max over ranks), kept here so
that the CPU tests drive exactly the code the GPU ranks run.
"""
from __future__ import annotations

import os
import time
from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def frames_for_rank(n_frames: int, rank: int, world: int) -> range:
    """Contiguous chunk of frame ids owned by ``rank`` (the reference's DistributedSampler rule; the total is padded up
    to a multiple of ``world`` there, here the tail ranks simply get one frame less)."""
    per = -(-n_frames // world)
    lo = min(rank * per, n_frames)
    return range(lo, min(lo + per, n_frames))


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_gather_features(feat: torch.Tensor, out: torch.Tensor = None, dtype=torch.bfloat16) -> torch.Tensor:
    """feat (Nv, C, h, w) of this rank's frame -> (world, Nv, C, h, w) on every rank, exchanged in ``dtype``
    (bf16: 3.07 MB per rank at 6x256x20x50).  One collective per frame, none inside the backbone.  Synchronous form
    (the collective is ordered on the caller's stream); ``FeatureGather`` is the overlapped one."""
    world = _world()
    send = feat.to(dtype).contiguous()
    if out is None:
        out = torch.empty((world,) + tuple(send.shape), dtype=dtype, device=send.device)
    if world == 1:
        out[0].copy_(send)
        return out
    dist.all_gather_into_tensor(out.flatten(0, 1), send)      # (world*Nv, ...) view: the layout gloo and RCCL both accept
    return out


class FeatureGather:
    """Overlapped per-frame feature exchange (``streampetr_head.py:627-631`` is where the head reads the result).

    ``submit(feat)`` converts this rank's neck features into one of ``depth`` send buffers on the caller's stream, then
    orders the all-gather behind that copy on a private side stream and returns a ticket; the caller's stream is free
    to run the next frame at once.  ``wait(ticket)`` makes the caller's stream (GPU) or the calling thread (CPU / gloo) wait
    for that exchange and returns the gathered ``(world, Nv, C, h, w)`` tensor.  A buffer pair is reused every ``depth`` frames:
    ``submit`` first waits for the exchange that last used it (a no-op unless the consumer lags ``depth`` frames behind).
    """

    def __init__(self, shape, device, dtype=torch.bfloat16, depth: int = 2, force_collective: bool = False):
        self.world = _world()
        self.force_collective = force_collective         # tests: issue the collective even in a one-rank group (exercises the RCCL / stream path)
        self.device = torch.device(device)
        self.depth = depth
        self.send = [torch.empty(tuple(shape), dtype=dtype, device=self.device) for _ in range(depth)]
        self.out = [torch.empty((self.world,) + tuple(shape), dtype=dtype, device=self.device) for _ in range(depth)]
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.done = [None] * depth                       # per slot: event (GPU) or Work (CPU) of the exchange in flight
        self.n = 0

    def submit(self, feat: torch.Tensor) -> int:
        slot = self.n % self.depth
        self._finish(slot)
        self.send[slot].copy_(feat)                      # dtype conversion + layout, on the caller's stream
        if self.world == 1 and not self.force_collective:
            self.out[slot][0].copy_(self.send[slot])
        elif self.cuda:
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                dist.all_gather_into_tensor(self.out[slot].flatten(0, 1), self.send[slot])
                ev = torch.cuda.Event()
                ev.record(self.side)
            self.done[slot] = ev
        else:
            self.done[slot] = dist.all_gather_into_tensor(self.out[slot].flatten(0, 1), self.send[slot], async_op=True)
        self.n += 1
        return self.n - 1

    def _finish(self, slot: int):
        d = self.done[slot]
        if d is None:
            return
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(d)
        else:
            d.wait()
        self.done[slot] = None

    def wait(self, ticket: int) -> torch.Tensor:
        assert self.n - self.depth <= ticket < self.n, "ticket is older than the buffer ring"
        slot = ticket % self.depth
        self._finish(slot)
        return self.out[slot]

    def drain(self):
        for s in range(self.depth):
            self._finish(s)


def exchange_census(feat: torch.Tensor, device, dtype=torch.bfloat16) -> dict:
    """One VERIFIED feature exchange, so that a multi-GPU run proves by itself that the collective saw every rank: each rank all-gathers
    (rank id, bytes sent, checksum of the bits it sent) with the same collective that moves the features, then recomputes the checksum of
    every slice of the gathered tensor and compares it with what that slice's sender announced.  Returns ``ranks_seen`` (the rank ids found
    in the gathered records, in slot order), ``gather_bytes`` (bytes every rank receives per exchange) and ``verified`` (AND over all ranks)."""
    world = _world()
    rank = dist.get_rank() if world > 1 else 0
    send = feat.to(dtype).contiguous()
    got = all_gather_features(feat, dtype=dtype)

    def checksum(t):                                     # order-independent, exact: the 16-bit patterns summed as integers
        return int(t.contiguous().view(torch.int16).to(torch.int64).sum().item())
    mine = torch.tensor([rank, send.numel() * send.element_size(), checksum(send)], dtype=torch.int64, device=send.device)
    rec = torch.empty(world, 3, dtype=torch.int64, device=send.device)
    if world > 1:
        dist.all_gather_into_tensor(rec.flatten(), mine)
    else:
        rec[0].copy_(mine)
    rec = rec.cpu()
    ok = rec[:, 0].tolist() == list(range(world)) and all(checksum(got[r]) == int(rec[r, 2]) for r in range(world)) \
        and all(int(rec[r, 1]) == send.numel() * send.element_size() for r in range(world))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=send.device)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return {"ranks_seen": [int(v) for v in rec[:, 0].tolist()], "gather_bytes": int(rec[:, 1].sum()), "verified": bool(int(flag.item()) == 1)}


def pin_rank_to_cores(local_rank: int, local_world: int) -> Optional[List[int]]:
    """Give every rank of a node its own contiguous slice of the host cores this process may use (the launch path is one
    host thread per rank; without pinning the eight ranks' threads migrate across sockets).  Returns the slice, or None
    when the platform has no affinity call or there are fewer cores than ranks."""
    if not hasattr(os, "sched_getaffinity") or local_world <= 1:
        return None
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // local_world
    if per < 1:
        return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine


def max_over_ranks(seconds: float, device) -> float:
    if _world() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device):
    """Device sync + rank barrier + device sync: what brackets bench.py's timed region on both sides."""
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    if _world() > 1:
        dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)


def timed_steps(step: Callable[[], None], steps: int, warmup: int, device, finish: Callable[[], None] = None) -> float:
    """bench.py's protocol: ``warmup`` untimed steps, then EXACTLY ``steps`` steps bracketed by ``barrier`` on both sides
    (``finish``, e.g. draining an overlapped exchange, runs inside the timed region before the closing barrier);
    returns the elapsed seconds, MAX over ranks."""
    for _ in range(warmup):
        step()
    if finish is not None:
        finish()
    barrier(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if finish is not None:
        finish()
    barrier(device)
    return max_over_ranks(time.perf_counter() - t0, device)
