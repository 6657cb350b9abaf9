"""GPU: the overlapped feature exchange on real RCCL.  Only one GPU is available to the tests, so the process group has ONE rank; the
collective is still issued (force_collective), which exercises what the CPU / gloo tests cannot: torch.distributed's "nccl" backend (RCCL),
the flattened (world*Nv, ...) output view, the side stream and the event hand-offs of toc3d_amd.dist.FeatureGather, and bench.py's
timed_steps protocol on a CUDA device."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from toc3d_amd import dist as tdist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_feature_gather_on_rccl_single_rank():
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        g = tdist.FeatureGather((6, 256, 20, 50), DEV, depth=2, force_collective=True)
        frames = [torch.randn(6, 256, 20, 50, device=DEV) for _ in range(5)]
        filler = torch.randn(2048, 2048, device=DEV)
        tickets = []

        def step():
            i = len(tickets)
            t = g.submit(frames[i % 5])                  # exchange ordered behind the conversion, on the side stream
            (filler @ filler).sum()                      # "next frame" on the compute stream meanwhile
            tickets.append(t)
            if i >= 1:
                got = g.wait(t - 1)                      # the consumer reads the previous frame's exchange
                assert got.shape == (1, 6, 256, 20, 50) and torch.equal(got[0], frames[(i - 1) % 5].to(torch.bfloat16))

        elapsed = tdist.timed_steps(step, steps=6, warmup=2, device=DEV, finish=g.drain)
        assert elapsed > 0 and g.n == 8
        assert torch.equal(g.wait(g.n - 1)[0], frames[(g.n - 1) % 5].to(torch.bfloat16))
        sync = tdist.all_gather_features(frames[0])      # the synchronous form, world 1: a conversion + copy
        assert torch.equal(sync[0], frames[0].to(torch.bfloat16))
    finally:
        dist.destroy_process_group()
